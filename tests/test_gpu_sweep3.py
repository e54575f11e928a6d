"""
GPU parity tests (-m gpu) of the blocked-inverse sweep (qoc_amd/csrc/qocx_sweep3.hip, knob
"sweep_impl" = 3: the latency mode of the single-control-set entry points) - the same gates as the
default column-chain sweep: golden vectors minted from the reference (states / cost 1e-10,
gradients 1e-8 vs AD and 1e-7 vs finite differences of the reference forward), the oracle on
random shapes, bit-identical results across memory chunks and time segments, and agreement with
the default sweep to rounding.
"""

import numpy as np
import pytest

from oracle import qoc_numpy as onp
from tests import cases as cases_mod
from tests.helpers import GRAD_CASE_NAMES, golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from qoc_amd.engine import Engine
    e = Engine(0)
    e.set_knob("sweep_impl", 3)
    # (n <= 16 would otherwise take the inverse-image sweep of qocx_sweepi.hip, whose tests are in
    # test_gpu_engine.py: this module is about the blocked sweep at both tile sizes)
    e.set_knob("sweep_inverse_small", 0)
    yield e
    e.close()


@pytest.mark.parametrize("name", GRAD_CASE_NAMES)
def test_blocked_sweep_matches_golden(engine, name):
    from tests import gpu_helpers as gh
    case = cases_mod.case_by_name(name)
    g = golden(name)
    host_specs = gh.setup_engine(engine, case)
    cost, grads, final = engine.evaluate(gh.real_controls(case, case.controls), want_grad=True)
    grads = gh.complex_grads(case, grads)
    host_costs = [getattr(onp, k)(**kw) for k, kw in host_specs]
    for b in range(len(case.controls)):
        err = cost[b]
        gb = grads[b].astype(np.complex128)
        for c in host_costs:
            err = err + c.cost(case.controls[b], None, case.N - 1)
            gb = gb + c.controls_bar(case.controls[b], None, case.N - 1)
        assert abs(err - g["error"][b]) <= 1e-10 * max(1.0, abs(g["error"][b])), (b, err)
        assert rel_err(final[b][:, :, None], g["final_states"][b]) < 1e-10
        assert rel_err(gb, g["grads_ad"][b]) < 1e-8
        scale = np.max(np.abs(g["grads_ad"][b]))
        assert np.max(np.abs(gb.flat[g["fd_index"][b]] - g["grads_fd"][b])) / scale < 1e-7


@pytest.mark.parametrize("name", ["nc10_n101", "scaled_n8", "small_complex_M2", "nonhermitian_n24",
                                  "magnus_n20_M4"])
def test_blocked_sweep_segments_chunks_and_default_sweep(engine, name):
    """Chunks / time segments resume the blocked sweep from its saved state bit for bit (squarings,
    step costs, several states, both tile sizes among the cases); the column-chain sweep agrees
    to rounding."""
    from tests import gpu_helpers as gh
    case = cases_mod.case_by_name(name)
    gh.setup_engine(engine, case)
    u = gh.real_controls(case, np.concatenate([case.controls, 0.5 * case.controls,
                                               -case.controls]))
    ref = engine.evaluate(u, True)
    try:
        for chunk, pipe in ((2, 1), (1, 3), (0, 5), (3, 2)):
            engine.set_chunk(chunk)
            engine.set_pipeline(pipe)
            out = engine.evaluate(u, True)
            for a, b in zip(ref, out):
                assert np.array_equal(a, b), (chunk, pipe)
        engine.set_chunk(0)
        engine.set_pipeline(0)
        engine.set_knob("sweep_impl", 1)
        other = engine.evaluate(u, True)
        assert np.max(np.abs(other[0] - ref[0])) < 1e-12
        assert np.max(np.abs(other[1] - ref[1])) < 1e-11 * max(1.0, np.max(np.abs(ref[1])))
        assert np.max(np.abs(other[2] - ref[2])) < 1e-12
    finally:
        engine.set_chunk(0)
        engine.set_pipeline(0)
        engine.set_knob("sweep_impl", 3)


def test_blocked_sweep_batch_independence_and_state_cotangents(engine):
    """A seed's result does not depend on its batch neighbours; host-supplied state cotangents
    (user Cost plugins) flow through the blocked adjoint sweep exactly as through the default."""
    from tests import gpu_helpers as gh
    case = cases_mod.case_by_name("nc10_n101")
    gh.setup_engine(engine, case)
    u = gh.real_controls(case, np.concatenate([case.controls] * 3))
    cost, grads, final = engine.evaluate(u, True)
    c1, g1, f1 = engine.evaluate(u[4:5], True)
    assert c1[0] == cost[4] and np.array_equal(g1[0], grads[4]) and np.array_equal(f1[0], final[4])
    rng = np.random.default_rng(5)
    steps = [7, 40, case.N - 1]
    bars = rng.standard_normal((u.shape[0], len(steps), case.S, case.n)) \
        + 1j * rng.standard_normal((u.shape[0], len(steps), case.S, case.n))
    engine.set_state_cotangents(steps, 1e-2 * bars)
    try:
        _, g3, _ = engine.evaluate(u, True)
        engine.set_knob("sweep_impl", 1)
        _, g1, _ = engine.evaluate(u, True)
    finally:
        engine.set_state_cotangents(None, None)
        engine.set_knob("sweep_impl", 3)
    assert np.max(np.abs(g3 - grads)) > 1e-6  # the cotangents did something
    assert np.max(np.abs(g3 - g1)) < 1e-11 * max(1.0, np.max(np.abs(g1)))


def test_blocked_sweep_random_shapes_fuzz(engine):
    """tests/fuzz_parity.py on the blocked sweep: random sizes, grids, Magnus policies, Hermitian
    or not, time dependent or not, 0..4 squarings, every cost kind, against the oracle."""
    from tests import fuzz_parity
    rng = np.random.default_rng(4048)
    checked = 0
    for index in range(60):
        worst, tag = fuzz_parity.one(engine, rng, index)
        if worst is None:
            continue
        checked += 1
        assert worst < 1.0, tag
    assert checked > 40
