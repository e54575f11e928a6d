"""
GPU parity tests (-m gpu) of the path for Hilbert sizes ABOVE 64 (qoc_amd/csrc/qocx_general.hip, 65 <= n <= 1024,
every Magnus policy): through the C ABI against the oracle at the tolerances of the wavefront kernels (states and
cost 1e-10, gradients 1e-8). The reference is unbounded in n (qoc/core/schroedingerdiscrete.py:356-502).
"""

import numpy as np
import pytest

from oracle import qoc_numpy as onp
from tests import cases as cases_mod
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from qoc_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


GENERAL_EDGE_CASES = [
    dict(n=65, N=6, Nc=4, K=2, S=1, dt=0.05, ces=1, sigma=0.5),    # 15 padded rows (np = 80)
    dict(n=130, N=5, Nc=5, K=3, S=3, dt=0.02, ces=2, sigma=0.3),   # np = 144: three column chunks
    dict(n=100, N=7, Nc=3, K=2, S=5, dt=0.2, ces=3, sigma=1.0),    # squarings, five states
    dict(n=200, N=4, Nc=4, K=1, S=2, dt=0.1, ces=1, sigma=0.5),    # np = 208: four column chunks
    dict(n=256, N=3, Nc=2, K=2, S=1, dt=0.03, ces=1, sigma=0.4),   # the largest size with the pivot rows in LDS
    dict(n=300, N=3, Nc=3, K=1, S=2, dt=0.05, ces=1, sigma=0.5),   # np = 304: two panel rows per thread, chain vectors in HBM
    dict(n=512, N=3, Nc=2, K=1, S=1, dt=0.02, ces=1, sigma=0.3),   # the largest size with blocks of 16 pivots
    dict(n=600, N=3, Nc=2, K=1, S=1, dt=0.02, ces=1, sigma=0.3),   # np = 608: blocks of 8 pivots, three panel rows per thread
    dict(n=1024, N=2, Nc=2, K=1, S=1, dt=0.01, ces=1, sigma=0.3),  # the largest size (the last row of the reference's report)
]


@pytest.mark.parametrize("spec", GENERAL_EDGE_CASES, ids=lambda s: "n{n}_N{N}_Nc{Nc}_K{K}_S{S}".format(**s))
def test_general_path_edge_shapes(engine, spec):
    from tests.test_gpu_engine import test_edge_shapes_against_oracle as check
    check(engine, spec)


@pytest.mark.parametrize("pade_order, count, seed", [(0, 12, 65256), (13, 5, 1365)])
def test_general_path_fuzz(engine, pade_order, count, seed):
    """tests/fuzz_parity.py at 65 <= n <= 256: random grids, state counts, Hermitian or not, time dependent
    or not, 0..4 squarings, every state-cost kind; order by norm and the reference's always-[13/13]."""
    from tests import fuzz_parity
    rng = np.random.default_rng(seed)
    engine.set_knob("pade_order", pade_order)
    checked = 0
    try:
        for index in range(count):
            worst, tag = fuzz_parity.one(engine, rng, index, nmin=65, nmax=256, smax=3)
            if worst is None:
                continue
            checked += 1
            assert worst < 1.0, tag
    finally:
        engine.set_knob("pade_order", 0)
    assert checked >= count - 2


def test_general_path_chunks_and_batch_independence(engine):
    """Memory chunks of seeds and the batch a seed sits in do not change its numbers (bit for bit)."""
    from qoc_amd.engine import COST_TARGET_COHERENT
    n, N, K, S = 72, 9, 2, 2
    rng = np.random.default_rng(7272)
    h0 = cases_mod.gue(rng, n) * 1.5
    g = [cases_mod.gue(rng, n) for _ in range(K)]
    init = rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))
    init /= np.linalg.norm(init, axis=1, keepdims=True)
    targ = rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))
    targ /= np.linalg.norm(targ, axis=1, keepdims=True)
    engine.set_schroedinger_problem(n, S, K, N, N, 0.1 * (N - 1), h0[None], np.stack(g)[None], init,
                                    costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=targ)])
    controls = 0.5 * rng.standard_normal((5, N, K))
    ref = engine.evaluate(controls, True)
    try:
        for chunk in (1, 2):
            engine.set_chunk(chunk)
            out = engine.evaluate(controls, True)
            for a, b in zip(ref, out):
                assert np.array_equal(a, b), chunk
    finally:
        engine.set_chunk(0)
    # Hermitian H: K3 runs both chains on a^H, one pass over the generator per pair of chain steps (knob general_skew)
    engine.set_knob("general_skew", 0)
    try:
        plain = engine.evaluate(controls, True)
    finally:
        engine.set_knob("general_skew", 1)
    assert np.array_equal(plain[0], ref[0]) and rel_err(plain[1], ref[1]) < 1e-12
    one = engine.evaluate(controls[3:4], True)
    assert one[0][0] == ref[0][3] and np.array_equal(one[1][0], ref[1][3]) and np.array_equal(one[2][0], ref[2][3])
    # forward only: same costs and states, no gradient work
    cost, _, final = engine.evaluate(controls, False)
    assert np.array_equal(cost, ref[0]) and np.array_equal(final, ref[2])
    # the states at every system step (save_intermediate_states of the entry points) against the oracle's
    engine.set_keep_step_states(True)
    try:
        engine.evaluate(controls[:2], False)
        steps = engine.download_step_states()
    finally:
        engine.set_keep_step_states(False)
    problem = onp.SchroedingerProblem(0.1 * (N - 1), lambda u, t: h0 + u[0] * g[0] + u[1] * g[1], init[:, :, None],
                                      N, control_eval_count=N, costs=[onp.TargetStateInfidelity(targ[:, :, None])],
                                      control_count=K)
    for b in range(2):
        inter = []
        onp.evaluate(problem, controls[b], intermediate=inter)
        assert rel_err(steps[b][:, :, :, None], np.stack(inter)) < 1e-10


def test_general_path_user_cost_and_opaque_hamiltonian():
    """The two host routes above n = 64: state cotangents supplied by the host (a user Cost plugin, with
    and without its states_bar hook) against the built-in ForbidStates, and a Hamiltonian that is not
    linear in its controls (explicit generators + their cotangents, chain rule on the host) against
    central differences of its own cost and the oracle's forward pass."""
    from qoc_amd.core import device
    from qoc_amd.standard import ForbidStates, TargetStateInfidelity
    from tests.test_host_api import _UserOccupation
    n, N, K, ces = 70, 21, 2, 5
    rng = np.random.default_rng(70)
    h0 = cases_mod.gue(rng, n)
    g = [cases_mod.gue(rng, n) for _ in range(K)]
    g2 = cases_mod.gue(rng, n)
    init = np.zeros((1, n, 1), dtype=np.complex128)
    init[0, 0, 0] = 1
    targ = np.zeros((1, n, 1), dtype=np.complex128)
    targ[0, 2, 0] = 1
    forb = np.zeros((1, 1, n, 1), dtype=np.complex128)
    forb[0, 0, 1, 0] = 1
    T = 0.15 * (N - 1)
    count = (N - 1) // ces

    def linear(u, t):
        return h0 + u[0] * g[0] + u[1] * g[1]

    args = dict(control_count=K, control_eval_count=N, cost_eval_step=ces)
    batch = 0.4 * rng.standard_normal((3, N, K))
    ev_ref = device.SchroedingerEvaluator(
        T, linear, init, N, costs=[TargetStateInfidelity(targ),
                                   ForbidStates(forb, N, cost_eval_step=ces, cost_multiplier=0.7)], **args)
    e0, g0, f0, _ = ev_ref.evaluate_batch(batch)
    ocosts = [onp.TargetStateInfidelity(targ),
              onp.ForbidStates(forb, N, cost_eval_step=ces, cost_multiplier=0.7)]
    problem = onp.SchroedingerProblem(T, linear, init, N, control_eval_count=N, costs=ocosts,
                                      cost_eval_step=ces, control_count=K)
    for b in range(batch.shape[0]):
        err, gr, fin = onp.evaluate_with_grad(problem, batch[b])
        assert abs(err - e0[b]) < 1e-10 and rel_err(f0[b], fin) < 1e-10
        assert np.max(np.abs(gr - g0[b])) < 1e-8 * np.max(np.abs(gr))
    for with_hook in (True, False):
        ev_user = device.SchroedingerEvaluator(
            T, linear, init, N, costs=[TargetStateInfidelity(targ),
                                       _UserOccupation(count, with_hook, cost_multiplier=0.7)], **args)
        e1, g1, f1, _ = ev_user.evaluate_batch(batch)
        assert np.max(np.abs(e0 - e1)) < 1e-12 and rel_err(f1, f0) < 1e-12
        assert rel_err(g1, g0) < (1e-11 if with_hook else 1e-7)

    def nonlinear(u, t):
        return h0 + u[0] * g[0] + u[1] * g[1] + (u[0] ** 2) * g2

    ev = device.SchroedingerEvaluator(T, nonlinear, init, N, costs=[TargetStateInfidelity(targ)],
                                      control_count=K, control_eval_count=N)
    assert ev.opaque_hamiltonian is not None
    errors, grads, finals, _ = ev.evaluate_batch(batch[:1], want_grad=True)
    oproblem = onp.SchroedingerProblem(T, nonlinear, init, N, control_eval_count=N,
                                       costs=[onp.TargetStateInfidelity(targ)], control_count=K)
    oerr, ofin = onp.evaluate(oproblem, batch[0])[:2]
    assert abs(errors[0] - oerr) < 1e-10 and rel_err(finals[0], ofin) < 1e-10
    for (j, k) in ((0, 0), (7, 1), (N - 1, 0)):
        h = 1e-5
        up, dn = batch[:1].copy(), batch[:1].copy()
        up[0, j, k] += h
        dn[0, j, k] -= h
        fd = (ev.evaluate_batch(up, want_grad=False)[0][0] - ev.evaluate_batch(dn, want_grad=False)[0][0]) / (2 * h)
        assert abs(fd - grads[0][j, k]) < 1e-7 * max(1.0, np.max(np.abs(grads[0])))


def test_general_path_entry_points():
    """evolve / grape through the reference's entry points at n = 80: the error falls, bounds hold."""
    import qoc_amd
    from qoc_amd.standard import Adam, TargetStateInfidelity
    n, N = 80, 11
    rng = np.random.default_rng(80)
    h0 = cases_mod.gue(rng, n)
    g0 = cases_mod.gue(rng, n)
    init = np.zeros((1, n, 1), dtype=np.complex128)
    init[0, 0, 0] = 1
    targ = np.zeros((1, n, 1), dtype=np.complex128)
    targ[0, 1, 0] = 1

    def hamiltonian(u, t):
        return h0 + u[0] * g0

    controls = 0.2 * rng.standard_normal((N, 1))
    ev = qoc_amd.evolve_schroedinger_discrete(1.0, hamiltonian, init, N, controls=controls,
                                             costs=[TargetStateInfidelity(targ)])
    problem = onp.SchroedingerProblem(1.0, hamiltonian, init, N, control_eval_count=N,
                                      costs=[onp.TargetStateInfidelity(targ)], control_count=1)
    oerr, ofin = onp.evaluate(problem, controls)[:2]
    assert abs(ev.error - oerr) < 1e-10 and rel_err(ev.final_states, ofin) < 1e-10
    result = qoc_amd.grape_schroedinger_discrete(
        1, N, [TargetStateInfidelity(targ)], 1.0, hamiltonian, init, N, initial_controls=controls.copy(),
        iteration_count=6, log_iteration_step=0, optimizer=Adam(learning_rate=5e-2),
        max_control_norms=np.array([2.0]))
    assert result.best_error < ev.error and np.all(np.abs(result.best_controls) <= 2.0 + 1e-12)


@pytest.mark.parametrize("n", [66, 300, 530])
def test_general_path_pivots_off_the_diagonal(engine, n):
    """A generator whose Pade denominator is NOT diagonally dominant (a scaled cyclic shift: the sub-diagonal of
    P = v - u carries b_1 theta > b_0): the blocked Gauss-Jordan inversion interchanges rows in every block and
    undoes the column interchanges at the end; checked that LAPACK pivots off the diagonal there too."""
    import scipy.linalg
    from qoc_amd.engine import COST_TARGET_INCOHERENT
    from tests import device_model as dm
    N, K, S = 4, 1, 2   # (n = 300: two panel rows per thread, the pivot rows of a block in global scratch; n = 530: blocks of 8 pivots)
    rng = np.random.default_rng(n)
    dt = 0.25
    shift = np.roll(np.eye(n), 1, axis=0)
    h0 = 1j * (4.0 / dt) * shift + 0.3 * cases_mod.gue(rng, n)
    g = [cases_mod.gue(rng, n)]
    a = -1j * dt * h0
    u, v = dm.pade_uv(a, 13)
    piv = scipy.linalg.lu_factor(v - u)[1]
    assert np.count_nonzero(piv != np.arange(n)) > n // 2
    init = rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))
    init /= np.linalg.norm(init, axis=1, keepdims=True)
    targ = rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))
    targ /= np.linalg.norm(targ, axis=1, keepdims=True)
    T = dt * (N - 1)
    engine.set_schroedinger_problem(n, S, K, N, N, T, h0[None], np.stack(g)[None], init,
                                    costs=[dict(kind=COST_TARGET_INCOHERENT, step_cost=0, scale=1.0, vectors=targ)])
    controls = 0.1 * rng.standard_normal((2, N, K))
    cost, grads, final = engine.evaluate(controls, want_grad=True)
    problem = onp.SchroedingerProblem(
        T, lambda uu, t: h0 + uu[0] * g[0], init[:, :, None], N, control_eval_count=N,
        costs=[onp.TargetStateInfidelity(targ[:, :, None], neglect_relative_pahse=True)], control_count=K)
    for b in range(2):
        err, gr, fin = onp.evaluate_with_grad(problem, controls[b])
        # (this propagator is far from unitary: the states grow, so every gate is relative)
        assert abs(err - cost[b]) < 1e-10 * max(1.0, abs(err))
        assert rel_err(final[b][:, :, None], fin) < 1e-10
        assert np.max(np.abs(gr - grads[b])) < 1e-8 * np.max(np.abs(gr))


def test_expm_above_64():
    """qoc_amd.standard.expm (the reference's qoc.standard.functions.expm) on the general path: n = 100, a
    non-normal matrix with a norm that needs squarings, against SciPy."""
    import scipy.linalg
    from qoc_amd.standard import functions
    rng = np.random.default_rng(100)
    a = 0.8 * (rng.standard_normal((100, 100)) + 1j * rng.standard_normal((100, 100))) / 10
    out = functions.expm(a)
    ref = scipy.linalg.expm(a)
    assert rel_err(out, ref) < 1e-12


def test_full_propagator_between_33_and_64(engine):
    """33 <= n <= 64 with more states than the wavefront sweep's LDS holds (13): a full propagator (S = n = 40)
    runs on the general path (its sweep keeps the vectors in HBM) - against the oracle, and against the wavefront
    kernels on the first 13 columns."""
    from qoc_amd.engine import COST_TARGET_COHERENT
    n, N, K = 40, 6, 2
    rng = np.random.default_rng(4040)
    h0 = cases_mod.gue(rng, n) * 1.5
    g = [cases_mod.gue(rng, n) for _ in range(K)]
    init = np.eye(n, dtype=np.complex128)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    targ = q.T.copy()
    T = 0.1 * (N - 1)
    controls = 0.5 * rng.standard_normal((2, N, K))
    engine.set_schroedinger_problem(n, n, K, N, N, T, h0[None], np.stack(g)[None], init,
                                    costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=targ)])
    cost, grads, final = engine.evaluate(controls, want_grad=True)
    problem = onp.SchroedingerProblem(T, lambda u, t: h0 + u[0] * g[0] + u[1] * g[1], init[:, :, None], N,
                                      control_eval_count=N, costs=[onp.TargetStateInfidelity(targ[:, :, None])],
                                      control_count=K)
    for b in range(2):
        err, gr, fin = onp.evaluate_with_grad(problem, controls[b])
        assert abs(err - cost[b]) < 1e-10 and rel_err(final[b][:, :, None], fin) < 1e-10
        assert np.max(np.abs(gr - grads[b])) < 1e-8 * np.max(np.abs(gr))
    engine.set_schroedinger_problem(n, 13, K, N, N, T, h0[None], np.stack(g)[None], init[:13],
                                    costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=targ[:13])])
    _, _, final13 = engine.evaluate(controls, want_grad=False)
    assert rel_err(final13, final[:, :13]) < 1e-12


@pytest.mark.parametrize("n, S, N", [(80, 80, 5), (100, 9, 6), (130, 40, 4)])
def test_general_path_many_states(engine, n, S, N):
    """Eight states or more (a full propagator above n = 64 has n of them, up to 256): the sweep as products on the
    matrix cores, the states of a seed as the rows of a matrix - against the oracle, and equal to rounding
    to the vector form (the same problem with seven states)."""
    from qoc_amd.engine import COST_TARGET_COHERENT, COST_TARGET_INCOHERENT
    K = 2
    rng = np.random.default_rng(1000 * n + S)
    h0 = cases_mod.gue(rng, n) * 1.5
    g = [cases_mod.gue(rng, n) for _ in range(K)]
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    init = np.eye(n, dtype=np.complex128)[:S]
    targ = q.T[:S].copy()
    T = 0.3 * (N - 1)   # (the norm needs a squaring)
    controls = 0.5 * rng.standard_normal((2, N, K))
    descs = [dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=targ),
             dict(kind=COST_TARGET_INCOHERENT, step_cost=1, scale=0.5 / (N - 1), vectors=targ)]
    engine.set_schroedinger_problem(n, S, K, N, N, T, h0[None], np.stack(g)[None], init, costs=descs)
    cost, grads, final = engine.evaluate(controls, want_grad=True)
    ocosts = [onp.TargetStateInfidelity(targ[:, :, None]),
              onp.TargetStateInfidelityTime(N, targ[:, :, None], neglect_relative_pahse=True, cost_multiplier=0.5)]
    problem = onp.SchroedingerProblem(T, lambda u, t: h0 + u[0] * g[0] + u[1] * g[1], init[:, :, None], N,
                                      control_eval_count=N, costs=ocosts, control_count=K)
    for b in range(2):
        err, gr, fin = onp.evaluate_with_grad(problem, controls[b])
        assert abs(err - cost[b]) < 1e-10 and rel_err(final[b][:, :, None], fin) < 1e-10
        assert np.max(np.abs(gr - grads[b])) < 1e-8 * np.max(np.abs(gr))
    # final costs only: the states of a seed in groups of rows on several workgroups (split mode) - the same
    # products on the same rows, so the same numbers as one workgroup per seed (knob general_split 0)
    engine.set_schroedinger_problem(n, S, K, N, N, T, h0[None], np.stack(g)[None], init, costs=descs[:1])
    split = engine.evaluate(controls, want_grad=True)
    engine.set_knob("general_split", 0)
    try:
        whole = engine.evaluate(controls, want_grad=True)
    finally:
        engine.set_knob("general_split", 1)
    for a, b in zip(split, whole):
        assert np.array_equal(a, b)
    problem.costs = ocosts[:1]
    problem.step_costs = []
    err, gr, fin = onp.evaluate_with_grad(problem, controls[0])
    assert abs(err - split[0][0]) < 1e-10 and rel_err(split[2][0][:, :, None], fin) < 1e-10
    assert np.max(np.abs(gr - split[1][0])) < 1e-8 * np.max(np.abs(gr))
    engine.set_schroedinger_problem(n, 7, K, N, N, T, h0[None], np.stack(g)[None], init[:7],
                                    costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=targ[:7])])
    _, _, final7 = engine.evaluate(controls, want_grad=False)
    assert rel_err(final7, final[:, :7]) < 1e-12


@pytest.mark.parametrize("policy, n, K, time_dep", [("M6", 70, 2, True), ("M4", 80, 2, True), ("M4", 66, 9, False),
                                                    ("M6", 40, 1, False)])
def test_general_path_magnus_policies(engine, policy, n, K, time_dep):
    """M4 / M6 on the general path (magnus_kernel: node generators, commutators as products on the matrix cores,
    the reverse rules of mathmethods.py:96-164): time-dependent systems, M4 with more controls than its linear form
    takes (8), and - n = 40 with 20 states - a size the wavefront kernels hand over because of the state count.
    Against the oracle at the 1e-10 / 1e-8 gates."""
    from qoc_amd.engine import COST_TARGET_COHERENT, COST_TARGET_INCOHERENT
    from tests.fuzz_parity import NODES
    N, S = 6, (20 if n == 40 else 2)
    rng = np.random.default_rng(n + K)
    h0 = cases_mod.gue(rng, n) * 1.2
    g = [cases_mod.gue(rng, n) for _ in range(K)]
    dt = 0.15
    T = dt * (N - 1)

    def base(t):
        return h0 * (1 + 0.3 * np.cos(2.0 * t)) if time_dep else h0

    init = rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))
    init /= np.linalg.norm(init, axis=1, keepdims=True)
    targ = rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))
    targ /= np.linalg.norm(targ, axis=1, keepdims=True)
    if time_dep:
        times = [j * dt + c * dt for j in range(N - 1) for c in NODES[policy]]
        h0s, gs = np.stack([base(t) for t in times]), np.stack([np.stack(g) for _ in times])
    else:
        h0s, gs = h0[None], np.stack(g)[None]
    descs = [dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=targ),
             dict(kind=COST_TARGET_INCOHERENT, step_cost=1, scale=0.5 / (N - 1), vectors=targ)]
    engine.set_schroedinger_problem(n, S, K, N, N, T, h0s, gs, init, costs=descs, magnus_policy=policy)
    controls = 0.6 * rng.standard_normal((2, N, K))
    cost, grads, final = engine.evaluate(controls, want_grad=True)
    ocosts = [onp.TargetStateInfidelity(targ[:, :, None]),
              onp.TargetStateInfidelityTime(N, targ[:, :, None], neglect_relative_pahse=True, cost_multiplier=0.5)]
    problem = onp.SchroedingerProblem(T, lambda u, t: base(t) + sum(u[k] * g[k] for k in range(K)), init[:, :, None],
                                      N, control_eval_count=N, costs=ocosts, magnus_policy=policy, control_count=K)
    for b in range(2):
        err, gr, fin = onp.evaluate_with_grad(problem, controls[b])
        assert abs(err - cost[b]) < 1e-10 and rel_err(final[b][:, :, None], fin) < 1e-10
        assert np.max(np.abs(gr - grads[b])) < 1e-8 * np.max(np.abs(gr))


def test_example_full_propagator_of_two_transmons():
    """examples/two_transmon_cz_full_propagator.py (n = 100, 100 states): GRAPE lowers the error."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                        "two_transmon_cz_full_propagator.py")
    spec = importlib.util.spec_from_file_location("example_cz", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    result = module.main(iteration_count=4)
    assert result.best_iteration > 0 and np.isfinite(result.best_error) and result.best_error < 1.0
