"""
GPU parity tests (-m gpu): the HIP engine, through the C ABI, against the oracle's golden
vectors (minted from the reference) and against the NumPy model of the device algorithm.
Tolerances: states and cost 1e-10 relative, gradients 1e-8 relative.
"""

import numpy as np
import pytest

from oracle import qoc_numpy as onp
from tests import cases as cases_mod
from tests import device_model as dm
from tests.helpers import golden, oracle_costs, rel_err

pytestmark = pytest.mark.gpu

GRAD_CASES = [c.name for c in cases_mod.all_cases() if c.controls is not None]


@pytest.fixture(scope="module")
def engine():
    from qoc_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_wave_primitives(engine):
    failures, report = engine.selftest()
    assert failures == 0, report


@pytest.mark.parametrize("n", [2, 4, 8, 16, 17, 32, 33, 48, 64])
def test_pade_factor_kernel(engine, n):
    rng = np.random.default_rng(100 + n)
    mats = []
    theta = 5.371920351148152
    # (norms on either side of the squaring thresholds: the two-wave kernel decides the squaring
    # count from square-root-free bounds and forms the exact norm only when they disagree)
    for scale, skew in [(0.3, True), (2.0, True), (5.2, True), (5.5, True), (30.0, True),
                        (1.0, False), (11.0, False), (300.0, False),
                        (theta * (1 - 1e-12), True), (theta * (1 + 1e-12), True),
                        (4 * theta * (1 - 1e-12), False), (4 * theta * (1 + 1e-12), False),
                        (theta / 1.3, True), (theta * 1.3, False)]:
        g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        a = -1j * (g + g.conj().T) / 2 if skew else g
        mats.append(a * (scale / onp.one_norm(a)))
    # norms around the thresholds of the lower Pade orders (qocx_wave.h: order by norm)
    for order in (3, 5, 7, 9):
        for frac in (0.3, 0.7, 0.98, 1.02):
            g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
            a = -1j * (g + g.conj().T) / 2 if frac != 0.7 else g
            mats.append(a * (dm.PADE_THETA[order] * frac / onp.one_norm(a)))
    mats = np.stack(mats)
    default_mfma = 1
    for policy in (0, 13):
        # 17 <= n <= 32: the fused factorisation with its Schur updates on the matrix cores
        # (qocx_lu4.h: diagonal pivots, checked; the large-norm cases here leave the diagonal and
        # take the general elimination) and the one-wave elimination alone
        for lu_mfma in ((1, 0) if n > 16 else (default_mfma,)):  # (n > 32: qocx_lu4m.hip in front of lu4_kernel)
            engine.set_knob("pade_order", policy)
            engine.set_knob("lu_mfma", lu_mfma)
            try:
                out = engine.debug_pade_factor(mats)
            finally:
                engine.set_knob("pade_order", 0)
                engine.set_knob("lu_mfma", default_mfma)
            check_pade_factor(out, mats, policy)


@pytest.mark.parametrize("n", [3, 16, 20, 32])
def test_pade_inverse_kernel(engine, n):
    """K1b's sibling for the dense-state sweep (qocx_lu.h inv_body; knob "lu_inverse"): P^-1 by
    in-place Gauss-Jordan elimination with the factorisation's pivot rule, instead of the factors."""
    rng = np.random.default_rng(300 + n)
    mats = []
    for scale, skew in [(0.1, True), (0.8, True), (3.0, False), (9.0, True), (40.0, False)]:
        g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        a = -1j * (g + g.conj().T) / 2 if skew else g
        mats.append(a * (scale / onp.one_norm(a)))
    mats = np.stack(mats)
    engine.set_knob("lu_inverse", 1)
    try:
        out = engine.debug_pade_factor(mats)
    finally:
        engine.set_knob("lu_inverse", 0)
    for m, a in enumerate(mats):
        f = dm.pade_factor(a, order=int(out["order"][m]))
        p_mat = dm.pade_uv(f["a"], f["order"])
        p_mat = p_mat[1] - p_mat[0]
        assert rel_err(out["q"][m], f["q"]) < 1e-12
        assert rel_err(out["lu"][m] @ p_mat, np.eye(n)) < 1e-12 * np.linalg.cond(p_mat)


@pytest.mark.parametrize("n, count", [(16, 9), (8, 4), (1, 3), (13, 6)])
def test_pade_inverse_four_to_a_wave(engine, n, count):
    """n <= 16, every matrix of the launch diagonally dominant (eps(theta) <= 0.40, qocx_lu5.h): P^-1
    of four matrices per wave by DPP multiply-adds, no pivot search (inv16_dpp_kernel), against the
    one-matrix-per-wave Gauss-Jordan kernel with LAPACK's pivot rule (knob "lu_dpp" 0) and against the
    model; `count` not a multiple of four leaves rows of lanes without a matrix."""
    rng = np.random.default_rng(900 + n)
    mats = []
    for c in range(count):
        g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        a = -1j * (g + g.conj().T) / 2 if c % 2 == 0 else g
        mats.append(a * (float(rng.uniform(0.01, 0.65)) / max(onp.one_norm(a), 1e-300)))
    mats = np.stack(mats)
    engine.set_knob("lu_inverse", 1)
    try:
        out = engine.debug_pade_factor(mats)
        engine.set_knob("lu_dpp", 0)
        ref = engine.debug_pade_factor(mats)
    finally:
        engine.set_knob("lu_dpp", 1)
        engine.set_knob("lu_inverse", 0)
    assert np.array_equal(out["q"], ref["q"]) and np.array_equal(out["order"], ref["order"])
    for m, a in enumerate(mats):
        f = dm.pade_factor(a, order=int(out["order"][m]))
        p_mat = dm.pade_uv(f["a"], f["order"])
        p_mat = p_mat[1] - p_mat[0]
        assert rel_err(out["lu"][m], ref["lu"][m]) < 1e-13
        assert rel_err(out["lu"][m] @ p_mat, np.eye(n)) < 1e-13 * np.linalg.cond(p_mat)


def check_pade_factor(out, mats, policy, expect_lower=True):
    lower = 0
    for m, a in enumerate(mats):
        # the kernel decides from an upper bound of the norm: never a lower order than the norm
        # allows, and 13 whenever the policy says so
        order = int(out["order"][m])
        assert order in (3, 5, 7, 9, 13)
        assert order >= dm.pade_order(onp.one_norm(a), policy)
        bound = np.max(np.sum(np.abs(a.real) + np.abs(a.imag), axis=0))
        if policy == 13:
            assert order == 13
        elif mats.shape[1] > 16 and mats.shape[1] <= 32:  # two-wave K1a: from the square-root-free bound
            assert order == dm.pade_order(bound)
        else:                                             # one- and four-wave K1a: from the exact norm
            assert order == dm.pade_order(onp.one_norm(a))
        lower += order < 13
        f = dm.pade_factor(a, order=order)
        assert out["s"][m] == f["s"]
        assert rel_err(out["q"][m], f["q"]) < 1e-12
        assert np.array_equal(out["perm"][m], f["perm"])
        assert rel_err(out["lu"][m], f["lu"]) < 1e-11
        assert rel_err(out["dinv"][m], 1.0 / np.diag(f["lu"])) < 1e-11
        # and the propagator itself against expm_pade (reference restatement)
        u = dm.solve_lu(out["lu"][m], out["perm"][m], out["q"][m])
        for _ in range(int(out["s"][m])):
            u = u @ u
        assert rel_err(u, onp.expm_pade(a)) < (1e-10 if order == 13 else 1e-13)
    if policy == 0 and expect_lower:
        assert lower >= 12


@pytest.mark.parametrize("pade_order", [0, 13])
@pytest.mark.parametrize("name", GRAD_CASES)
def test_engine_matches_golden(engine, name, pade_order):
    """Every reference-minted fixture at the parity gates, under both Pade policies: the order by
    norm (default) and always [13/13], which is what the reference executes (knob "pade_order")."""
    from tests import gpu_helpers as gh
    case = cases_mod.case_by_name(name)
    g = golden(name)
    host_specs = gh.setup_engine(engine, case)
    engine.set_knob("pade_order", pade_order)
    try:
        cost, grads, final = engine.evaluate(gh.real_controls(case, case.controls), want_grad=True)
        orders = engine.pade_orders()
    finally:
        engine.set_knob("pade_order", 0)
    assert pade_order == 0 or orders[13] == sum(orders.values())
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


@pytest.mark.parametrize("magnus", ["M2", "M4", "M6"])
def test_forward_only_and_iswap(engine, magnus):
    case = cases_mod.case_iswap(magnus)
    from tests import gpu_helpers as gh
    gh.setup_engine(engine, case)
    engine.set_keep_step_states(True)
    engine.upload_controls(1)
    engine.eval_resident(False)
    cost, _, final = engine.download_results(want_grad=False)
    steps = engine.download_step_states()
    engine.set_keep_step_states(False)
    g = golden(case.name)
    assert rel_err(final[0][:, :, None], g["final_states"][0]) < 1e-10
    target = np.array(((1, 0, 0, 0), (0, 0, -1j, 0), (0, -1j, 0, 0), (0, 0, 0, 1)))
    assert np.allclose(final[0].T, target)
    assert np.allclose(steps[0, 0].T, np.eye(4))
    assert rel_err(steps[0, -1], final[0]) == 0


def test_chunked_equals_unchunked(engine):
    from tests import gpu_helpers as gh
    case = cases_mod.case_by_name("nc10_n101")
    gh.setup_engine(engine, case)
    u = np.concatenate([case.controls, case.controls * 0.5, -case.controls])
    ref = engine.evaluate(u, True)
    engine.set_chunk(2)
    out = engine.evaluate(u, True)
    engine.set_chunk(0)
    for a, b in zip(ref, out):
        assert np.array_equal(a, b)
    # pipelined sub-chunks (sweep overlapped on side streams) vs one stream
    for pipe in (1, 3, 6):
        engine.set_pipeline(pipe)
        out = engine.evaluate(u, True)
        for a, b in zip(ref, out):
            assert np.array_equal(a, b)
    engine.set_pipeline(0)


@pytest.mark.parametrize("n, S, dt, scale, non_herm", [(8, 1, 0.05, 1.0, False), (16, 2, 0.4, 4.0, False),
                                                       (32, 1, 0.05, 1.0, False), (27, 3, 0.3, 3.0, True),
                                                       (5, 1, 1.0, 6.0, True)])
def test_propagator_image_sweep_of_one_control_set(engine, n, S, dt, scale, non_herm):
    """
    Round 5 (knob "sweep_umode", on in latency mode): with one control set at a time the inverse-image sweep
    (qocx_sweepi.hip) applies the propagator U = P^-1 Q itself - umul_kernel leaves it in the Q image - ONE
    matrix-vector product per sub-step instead of two; the adjoint sweep hands lambda' to K3, which forms
    x = P^-H lambda' from the P^-1 image. Against the two-product form: cost, gradient and final states
    equal to rounding - unit adjoint and the forward -> cost -> adjoint order, step costs, several states,
    squarings (dt * scale large), row exchanges (non-Hermitian generator), one and two MFMA tiles, one
    launch and time segments. Reference: qoc/core/schroedingerdiscrete.py:393-436, expm.py:246-250.
    """
    from tests import gpu_helpers as gh
    case = cases_mod.case_random("umode_n%d" % n, n=n, N=31, seeds=1, h_seed=7900 + n, S=S, K=2, Nc=11, dt=dt,
                                 sigma=0.6)
    case.h0 = case.h0 * scale
    if non_herm:
        rng = np.random.default_rng(7950 + n)
        case.h0 = case.h0 - 0.3j * np.diag(rng.uniform(0, 1, n)) + 1.0 * np.roll(np.eye(n), 1, axis=0)
    rng = np.random.default_rng(7960 + n)
    forb = np.stack([cases_mod.column_states(cases_mod.random_unitary(rng, n)[:, :2]) for _ in range(S)])
    variants = [(case.cost_specs, 1),
                (case.cost_specs + [("ForbidStates", dict(forbidden_states=forb, system_eval_count=case.N,
                                                          cost_eval_step=3, cost_multiplier=0.3))], 3)]
    try:
        engine.set_knob("latency", 1)
        for specs, ces in variants:
            case.cost_specs, case.cost_eval_step = specs, ces
            gh.setup_engine(engine, case)
            u = np.asarray(case.controls[:1])
            for pipe in (0, 1, 3):
                engine.set_pipeline(pipe)
                engine.set_knob("sweep_umode", 0)
                ref = engine.evaluate(u, True)
                engine.set_knob("sweep_umode", 1)
                out = engine.evaluate(u, True)
                for a, b in zip(ref, out):
                    assert np.max(np.abs(a - b)) <= 2e-11 * max(1.0, np.max(np.abs(a))), (ces, pipe)
    finally:
        engine.set_knob("latency", 0)
        engine.set_knob("sweep_umode", 1)
        engine.set_pipeline(0)


@pytest.mark.parametrize("n, N, non_herm", [(8, 41, False), (8, 40, False), (5, 24, False), (2, 9, False),
                                            (7, 33, True)])
def test_two_steps_to_a_tile_at_n_up_to_8(engine, n, N, non_herm):
    """
    Round 5 (knob "pack8", on): at n <= 8 K1a and K1b take two consecutive steps of a seed as the diagonal
    blocks of ONE 16 x 16 tile (pade_pq8_kernel, inv16_dpp_kernel<1, true>). A
    block-diagonal generator stays block diagonal through the Pade evaluation and through the inverse; the
    pair shares the order and squaring count of its larger member. Against one step per tile: equal to
    rounding (a step may be evaluated by a higher-order approximant than it needs), and bit for bit where
    both steps of every pair choose the same order anyway; odd and even step counts, time segments whose
    lengths are odd, control magnitudes that put neighbouring steps on different orders, a non-Hermitian
    generator. Reference: qoc/standard/functions/expm.py:210-252.
    """
    from tests import gpu_helpers as gh
    case = cases_mod.case_random("pack8_n%d" % n, n=n, N=N, seeds=3, h_seed=7700 + n, S=1, K=1, Nc=N,
                                 dt=0.05, sigma=0.6)
    case.h0 = case.h0 * 0.1  # (a small drift: the steps of the last control array alternate between orders 3 and 5)
    if non_herm:
        rng = np.random.default_rng(7800 + n)
        case.h0 = case.h0 - 0.2j * np.diag(rng.uniform(0, 1, n))
    gh.setup_engine(engine, case)
    # (the last control array swings between tiny and large: neighbouring steps on different orders)
    swing = case.controls[0] * np.where((np.arange(case.controls[0].shape[0]) // 2) % 2 == 0, 0.01, 2.5)[:, None]
    u = np.concatenate([case.controls, swing[None]])
    try:
        for pipe in (1, 3, 0):
            engine.set_pipeline(pipe)
            engine.set_knob("pack8", 0)
            ref = engine.evaluate(u, True)
            engine.set_knob("pack8", 1)
            out = engine.evaluate(u, True)
            for a, b in zip(ref, out):
                assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(a))), (pipe,)
    finally:
        engine.set_knob("pack8", 1)
        engine.set_pipeline(0)


@pytest.mark.parametrize("n, dt, scale, non_herm", [(32, 0.05, 1.0, False), (20, 0.4, 6.0, False),
                                                    (27, 0.3, 3.0, True), (8, 0.05, 1.0, False),
                                                    (13, 1.0, 6.0, True)])
def test_one_state_sweep_equals_general_sweep(engine, n, dt, scale, non_herm):
    """
    qocx_sweep1.hip (one state per seed, n <= 32; knob "sweep_one", on) against the general
    column-chain sweep it replaces on that path: the same arithmetic in the same order, so cost,
    gradient and final state are equal BIT FOR BIT - with the forward -> cost -> adjoint order and
    with the unit adjoint, in one launch and in time segments (resumed sweeps), with squarings
    (dt * scale large), with step costs every third step, with row exchanges in the factorisation
    (non-Hermitian generators with large off-diagonal entries) and at one and two MFMA tiles.
    Reference: qoc/core/schroedingerdiscrete.py:393-436.
    """
    from tests import gpu_helpers as gh
    case = cases_mod.case_random("sweep1_n%d" % n, n=n, N=41, seeds=3, h_seed=7300 + n, S=1, K=2,
                                 Nc=17, dt=dt, sigma=0.6)
    case.h0 = case.h0 * scale
    if non_herm:
        rng = np.random.default_rng(7400 + n)
        case.h0 = case.h0 - 0.3j * np.diag(rng.uniform(0, 1, n)) + 2.0 * np.roll(np.eye(n), 1, axis=0)
    rng = np.random.default_rng(7500 + n)
    forb = np.stack([cases_mod.column_states(cases_mod.random_unitary(rng, n)[:, 2:4])])
    variants = [
        (case.cost_specs, 1),
        (case.cost_specs + [("ForbidStates", dict(forbidden_states=forb, system_eval_count=case.N,
                                                  cost_eval_step=3, cost_multiplier=0.3))], 3),
    ]
    try:
        engine.set_knob("sweep_inverse_small", 0)  # (n <= 16: keep the column-chain sweep)
        for specs, ces in variants:
            case.cost_specs, case.cost_eval_step = specs, ces
            gh.setup_engine(engine, case)
            u = np.concatenate([case.controls, -0.7 * case.controls])
            for unit in (1, 0):
                engine.set_knob("unit_adjoint", unit)
                for pipe in (1, 4, 0):
                    engine.set_pipeline(pipe)
                    engine.set_knob("sweep_one", 0)
                    general = engine.evaluate(u, True)
                    engine.set_knob("sweep_one", 1)
                    one = engine.evaluate(u, True)
                    for a, b in zip(general, one):
                        assert np.array_equal(a, b), (specs[-1][0], ces, unit, pipe)
    finally:
        engine.set_pipeline(0)
        engine.set_knob("unit_adjoint", 1)
        engine.set_knob("sweep_inverse_small", 1)
        engine.set_knob("sweep_one", 1)


@pytest.mark.parametrize("n, dt, sigma", [(32, 0.05, 0.6), (17, 0.04, 0.3), (25, 0.004, 0.5)])
def test_three_wave_pade_kernel_equals_two_wave_kernel(engine, n, dt, sigma):
    """
    qocx_pade3.hip (Hermitian generators, every step at Pade order 3 or 5: one tile per wave on three
    waves, P factored by the vector-unit factorisation of qocx_lu5.h; knob "k1a_three", on) against the
    two-wave kernel with the checked MFMA factorisation (knobs "k1a_three" 0, "lu_dpp" 0): the same
    rational function, sums in another order - cost, gradient and final state equal to 1e-12 - and
    both within the parity gates of the oracle. dt = 0.004 puts the steps on order 3; the first problem
    has steps at order 7 among those at order 5 (they stay on the two-wave kernel).
    Reference: qoc/standard/functions/expm.py:119-135, :246.
    """
    from tests import gpu_helpers as gh
    from oracle import qoc_numpy as onp
    case = cases_mod.case_random("pade3_n%d" % n, n=n, N=33, seeds=3, h_seed=7600 + n, S=1, K=2, Nc=9,
                                 dt=dt, sigma=sigma)
    gh.setup_engine(engine, case)
    u = np.concatenate([case.controls, -0.5 * case.controls])
    try:
        engine.set_knob("k1a_three", 0)
        engine.set_knob("lu_dpp", 0)
        two = engine.evaluate(u, True)
        orders_two = engine.pade_orders()
        engine.set_knob("lu_dpp", 1)
        two_dpp = engine.evaluate(u, True)
        engine.set_knob("k1a_three", 1)
        three = engine.evaluate(u, True)
        # (the first problem mixes orders 5 and 7: both kernels run over the same grid and every
        # workgroup takes or leaves its step by the order in the step table)
        assert engine.pade_orders() == orders_two and orders_two[3] + orders_two[5] > 0
    finally:
        engine.set_knob("k1a_three", 1)
        engine.set_knob("lu_dpp", 1)
    for other in (two_dpp, three):
        assert abs(other[0] - two[0]).max() <= 1e-12
        assert np.abs(other[1] - two[1]).max() <= 1e-12 * max(1.0, np.abs(two[1]).max())
        assert np.abs(other[2] - two[2]).max() <= 1e-12
    h0, g = np.asarray(case.h0), [np.asarray(m) for m in case.g_re]
    problem = onp.SchroedingerProblem(
        case.T, lambda uu, t: h0 + uu[0] * g[0] + uu[1] * g[1], case.initial_states, case.N,
        control_eval_count=case.Nc, control_count=2,
        costs=[onp.TargetStateInfidelity(case.cost_specs[0][1]["target_states"])])
    for b in (0, 4):
        err, gr, fin = onp.evaluate_with_grad(problem, u[b])
        assert abs(err - three[0][b]) <= 1e-10
        assert rel_err(three[2][b], fin[:, :, 0]) <= 1e-10
        assert np.abs(gr - three[1][b]).max() <= 1e-8 * max(1.0, np.abs(gr).max())


@pytest.mark.parametrize("name", ["nc10_n101", "scaled_n8", "c3_fullU_short"])
def test_unit_adjoint_and_two_sided_pipeline(engine, name):
    """
    A single final TargetStateInfidelity makes the cotangent of the final states a scalar times the
    targets: the adjoint sweep then back-propagates the TARGETS (no dependence on the forward
    sweep), K3 emits complex numbers and the scatter kernel applies the scalar
    (qocx_sweep_common.h; knob "unit_adjoint"). Same derivative as the classic order - forward,
    cost, adjoint - to rounding; and the two-sided schedule built on it (knob "bidir": both sweeps
    at once, factorisation from both ends) is pure scheduling: bit-identical.
    """
    from tests import gpu_helpers as gh
    case = cases_mod.case_by_name(name)
    gh.setup_engine(engine, case)
    u = gh.real_controls(case, case.controls)
    u = np.concatenate([u, 0.5 * u, -u])
    try:
        engine.set_knob("unit_adjoint", 0)
        classic = engine.evaluate(u, True)
        engine.set_knob("unit_adjoint", 1)
        unit = engine.evaluate(u, True)
        assert np.array_equal(classic[0], unit[0]) and np.array_equal(classic[2], unit[2])
        assert np.max(np.abs(classic[1] - unit[1])) <= 1e-12 * np.max(np.abs(classic[1]))
        for pipe in (4, 8):
            engine.set_pipeline(pipe)
            engine.set_knob("bidir", 1)
            two_sided = engine.evaluate(u, True)
            engine.set_knob("bidir", 0)
            one_sided = engine.evaluate(u, True)
            for a, b, c in zip(unit, two_sided, one_sided):
                assert np.array_equal(a, b) and np.array_equal(a, c)
        if case.n > 16:
            # K1b fused into the two-wave K1a (P stays in LDS; knob "fuse_lu", the default since
            # the Pade order follows the norm) against the two kernels: the same
            # factors, bit for bit
            engine.set_pipeline(0)
            engine.set_knob("fuse_lu", 0)
            apart = engine.evaluate(u, True)
            for a, b in zip(unit, apart):
                assert np.array_equal(a, b)
    finally:
        engine.set_knob("fuse_lu", 1)
        engine.set_knob("unit_adjoint", 1)
        engine.set_knob("bidir", 1)
        engine.set_pipeline(0)


@pytest.mark.parametrize("n,K,herm", [(20, 3, True), (12, 2, True), (24, 2, False), (40, 3, True)])
def test_m4_commutator_free_form(engine, n, K, herm):
    """
    MagnusPolicy.M4 with time-independent H0, G_k: the commutators [G_k, H0], [G_k, G_l] are
    constant matrices, so the M4 generator is linear in 2 K + K (K - 1) / 2 effective controls
    and runs on the M2 kernels (qocx_device.h M4LinArgs; knob "m4_linear"). Same numbers as the
    Magnus kernels of qocx_magnus.hip (reference mathmethods.py:96-122) to rounding, same numbers
    as the oracle at the parity tolerances, and - being a property of the problem - bit-identical
    across batch chunks and time segments.
    """
    from tests import gpu_helpers as gh
    case = cases_mod.case_random("m4lin_n%d" % n, n=n, N=13, seeds=3, h_seed=7300 + n, S=2, K=K,
                                 Nc=7, dt=0.4, magnus="M4", sigma=1.0, full_unitary=True)
    case.h0 = case.h0 * 3.0
    if not herm:
        rng = np.random.default_rng(77)
        case.h0 = case.h0 - 0.1j * np.diag(rng.uniform(0, 1, n))
    gh.setup_engine(engine, case)
    u = gh.real_controls(case, case.controls)
    try:
        engine.set_knob("m4_linear", 0)
        kernels = engine.evaluate(u, True)
        engine.set_knob("m4_linear", 1)
        linear = engine.evaluate(u, True)
        assert np.max(np.abs(kernels[0] - linear[0])) <= 1e-12
        assert rel_err(linear[2], kernels[2]) <= 1e-12
        assert np.max(np.abs(kernels[1] - linear[1])) <= 1e-11 * np.max(np.abs(kernels[1]))
        engine.set_knob("unit_adjoint", 0)
        classic = engine.evaluate(u, True)
        assert np.array_equal(classic[0], linear[0])
        assert np.max(np.abs(classic[1] - linear[1])) <= 1e-12 * np.max(np.abs(classic[1]))
        engine.set_knob("unit_adjoint", 1)
        for chunk, pipe in ((1, 3), (2, 4), (0, 6)):
            engine.set_chunk(chunk)
            engine.set_pipeline(pipe)
            out = engine.evaluate(u, True)
            for a, b in zip(linear, out):
                assert np.array_equal(a, b), (chunk, pipe)
        h0, g = np.asarray(case.h0), [np.asarray(m) for m in case.g_re]
        problem = onp.SchroedingerProblem(
            case.T, lambda uu, t: h0 + sum(uu[k] * g[k] for k in range(K)), case.initial_states,
            case.N, control_eval_count=case.Nc, magnus_policy="M4", control_count=K,
            costs=[onp.TargetStateInfidelity(case.cost_specs[0][1]["target_states"])])
        for b in range(len(u)):
            err, gr, fin = onp.evaluate_with_grad(problem, u[b])
            assert abs(err - linear[0][b]) <= 1e-10
            assert rel_err(linear[2][b], fin[:, :, 0]) <= 1e-10
            assert np.max(np.abs(gr - linear[1][b])) <= 1e-8 * np.max(np.abs(gr))
    finally:
        engine.set_knob("m4_linear", 1)
        engine.set_knob("unit_adjoint", 1)
        engine.set_chunk(0)
        engine.set_pipeline(0)


def test_error_paths(engine):
    """The C ABI reports misuse and numerical trouble loudly (include/qocx.h error codes)."""
    from qoc_amd.engine import Engine, QocxError
    from tests import gpu_helpers as gh
    fresh = Engine(0)
    try:
        with pytest.raises(QocxError) as err:  # no problem set
            fresh.eval_resident(True)
        assert err.value.code == -3
        with pytest.raises(QocxError) as err:  # unsupported size
            fresh.set_schroedinger_problem(1025, 1, 0, 0, 5, 1.0, np.eye(1025), None, np.eye(1025)[:1])
        assert err.value.code == -1 and "hilbert_size" in err.value.message
        with pytest.raises(QocxError) as err:  # more than 64 states below hilbert_size 65
            fresh.set_schroedinger_problem(40, 65, 0, 0, 5, 1.0, np.eye(40), None, np.ones((65, 40)))
        assert "state_count" in err.value.message
        with pytest.raises(QocxError):  # nt neither 1 nor (N-1) * nodes
            fresh.set_schroedinger_problem(4, 1, 0, 0, 5, 1.0, np.stack([np.eye(4)] * 3), None,
                                           np.eye(4)[:1])
    finally:
        fresh.close()
    case = cases_mod.case_by_name("scaled_n8")
    gh.setup_engine(engine, case)
    bad = gh.real_controls(case, case.controls).copy()
    bad[0, 3, 0] = np.nan
    with pytest.raises(QocxError) as err:
        engine.evaluate(bad, want_grad=True)
    assert "non-finite" in err.value.message
    huge = gh.real_controls(case, case.controls) * 1e6  # ||dt H|| needs > 2^10 sub-steps
    with pytest.raises(QocxError) as err:
        engine.evaluate(huge, want_grad=True)
    assert err.value.code == -5
    with pytest.raises(QocxError):  # cotangents for another batch size
        engine.upload_controls(gh.real_controls(case, case.controls))
        engine.set_state_cotangents([1], np.zeros((5, 1, case.S, case.n), dtype=np.complex128))
        engine.eval_resident(True)
    engine.set_state_cotangents(None, None)
    # the engine is still usable afterwards
    cost, grads, final = engine.evaluate(gh.real_controls(case, case.controls), want_grad=True)
    assert np.all(np.isfinite(cost)) and np.all(np.isfinite(grads))


EDGE_CASES = [
    # n, N, Nc, K, S, dt, cost_eval_step, sigma
    dict(n=1, N=5, Nc=5, K=1, S=1, dt=0.3, ces=1, sigma=1.0),     # scalar "matrix"
    dict(n=2, N=2, Nc=2, K=2, S=2, dt=0.7, ces=1, sigma=1.0),     # a single propagator step
    dict(n=17, N=6, Nc=3, K=3, S=3, dt=0.2, ces=2, sigma=0.8),    # two tiles, 15 padded rows
    dict(n=16, N=9, Nc=2, K=1, S=16, dt=0.1, ces=3, sigma=2.0),   # full tile, Nc = 2 (one line)
    dict(n=32, N=4, Nc=7, K=2, S=64, dt=0.4, ces=1, sigma=0.5),   # maximum state count
    dict(n=5, N=12, Nc=12, K=4, S=1, dt=0.9, ces=20, sigma=3.0),  # cost_eval_step > N, squarings
    dict(n=31, N=5, Nc=9, K=8, S=2, dt=0.05, ces=1, sigma=0.1),   # Nc > N, many controls
    dict(n=20, N=14, Nc=5, K=2, S=6, dt=0.3, ces=2, sigma=1.5),   # 6 states on 4 sweep waves (2,2,1,1)
    dict(n=9, N=10, Nc=10, K=2, S=5, dt=0.9, ces=3, sigma=3.0),   # multi-wave sweep with squarings
    # 33 <= n <= 64: sixteen tiles (four-wave K1a, NB = 4 forms of K1b / K2 / K3)
    dict(n=33, N=6, Nc=4, K=2, S=1, dt=0.05, ces=1, sigma=0.5),   # 31 padded rows
    dict(n=64, N=5, Nc=5, K=3, S=3, dt=0.02, ces=2, sigma=0.3),   # full size, two sweep waves
    dict(n=40, N=7, Nc=3, K=2, S=5, dt=0.2, ces=3, sigma=1.0),    # four sweep waves, squarings
    dict(n=64, N=4, Nc=4, K=1, S=13, dt=0.1, ces=1, sigma=0.5),   # the most states that fit the LDS
]


@pytest.mark.parametrize("spec", EDGE_CASES, ids=lambda s: "n{n}_N{N}_Nc{Nc}_K{K}_S{S}".format(**s))
def test_edge_shapes_against_oracle(engine, spec):
    """Engine vs the oracle (itself pinned by the reference fixtures) on the corner shapes."""
    from qoc_amd.engine import COST_FORBID, COST_TARGET_COHERENT, COST_TARGET_INCOHERENT
    n, N, Nc, K, S = spec["n"], spec["N"], spec["Nc"], spec["K"], spec["S"]
    rng = np.random.default_rng(1000 * n + N)
    gue = cases_mod.gue
    h0 = gue(rng, n) * 2.0
    g = [gue(rng, n) for _ in range(K)]
    S_eff = min(S, 64)
    init = rng.standard_normal((S_eff, n)) + 1j * rng.standard_normal((S_eff, n))
    init /= np.linalg.norm(init, axis=1, keepdims=True)
    targ = rng.standard_normal((S_eff, n)) + 1j * rng.standard_normal((S_eff, n))
    targ /= np.linalg.norm(targ, axis=1, keepdims=True)
    forb = rng.standard_normal((S_eff, 2, n)) + 1j * rng.standard_normal((S_eff, 2, n))
    forb /= np.linalg.norm(forb, axis=2, keepdims=True)
    T = spec["dt"] * (N - 1)
    ces = spec["ces"]
    count = max((N - 1) // ces, 1)
    controls = spec["sigma"] * rng.standard_normal((3, Nc, K))
    descs = [dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=0.7, vectors=targ),
             dict(kind=COST_TARGET_INCOHERENT, step_cost=1, scale=1.3 / count, vectors=targ),
             dict(kind=COST_FORBID, step_cost=1, scale=0.9 / (count * S_eff),
                  vectors=forb.reshape(-1, n), counts=[2] * S_eff)]
    engine.set_schroedinger_problem(n, S_eff, K, Nc, N, T, h0[None], np.stack(g)[None], init,
                                    costs=descs, cost_eval_step=ces)
    cost, grads, final = engine.evaluate(controls, want_grad=True)
    ocosts = [onp.TargetStateInfidelity(targ[:, :, None], cost_multiplier=0.7),
              onp.TargetStateInfidelityTime(N, targ[:, :, None], neglect_relative_pahse=True,
                                            cost_eval_step=ces, cost_multiplier=1.3)
              if (N - 1) // ces > 0 else None,
              onp.ForbidStates(forb[:, :, :, None], N, cost_eval_step=ces, cost_multiplier=0.9)
              if (N - 1) // ces > 0 else None]
    ocosts = [c for c in ocosts if c is not None]
    problem = onp.SchroedingerProblem(
        T, lambda u, t: h0 + sum(u[k] * g[k] for k in range(K)), init[:, :, None], N,
        control_eval_count=Nc, costs=ocosts, cost_eval_step=ces, control_count=K)
    for b in range(controls.shape[0]):
        err, gr, fin = onp.evaluate_with_grad(problem, controls[b])
        if (N - 1) // ces > 0:
            assert abs(err - cost[b]) < 1e-10 * max(1.0, abs(err)), (err, cost[b])
            # (n = 1: every cost is phase invariant, the gradient is zero up to rounding)
            assert np.max(np.abs(grads[b] - gr)) < 1e-8 * max(np.max(np.abs(gr)), 1e-3)
        assert rel_err(final[b][:, :, None], fin) < 1e-10


def test_multi_state_sweep_segments(engine):
    """
    S > 1 runs the sweep on several waves per seed (states dealt to the waves, costs evaluated by
    wave 0 between workgroup barriers): the time-segmented pipeline, which resumes the sweep from
    its saved states / lambda, must reproduce the single-launch result bit for bit.
    """
    from qoc_amd.engine import COST_FORBID, COST_TARGET_COHERENT, COST_TARGET_INCOHERENT
    n, N, Nc, K, S, ces = 24, 41, 11, 2, 7, 4
    rng = np.random.default_rng(99)
    h0 = cases_mod.gue(rng, n) * 1.5
    g = [cases_mod.gue(rng, n) for _ in range(K)]
    init = rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))
    init /= np.linalg.norm(init, axis=1, keepdims=True)
    targ = rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))
    targ /= np.linalg.norm(targ, axis=1, keepdims=True)
    forb = rng.standard_normal((S, 1, n)) + 1j * rng.standard_normal((S, 1, n))
    forb /= np.linalg.norm(forb, axis=2, keepdims=True)
    count = (N - 1) // ces
    descs = [dict(kind=COST_TARGET_COHERENT, step_cost=1, scale=0.5 / count, vectors=targ),
             dict(kind=COST_TARGET_INCOHERENT, step_cost=0, scale=1.0, vectors=targ),
             dict(kind=COST_FORBID, step_cost=1, scale=0.4 / (count * S), vectors=forb.reshape(-1, n),
                  counts=[1] * S)]
    engine.set_schroedinger_problem(n, S, K, Nc, N, 0.2 * (N - 1), h0[None], np.stack(g)[None], init,
                                    costs=descs, cost_eval_step=ces)
    controls = rng.standard_normal((5, Nc, K))
    engine.set_pipeline(1)
    ref = engine.evaluate(controls, True)
    for pipe in (2, 5):
        engine.set_pipeline(pipe)
        out = engine.evaluate(controls, True)
        for a, b in zip(ref, out):
            assert np.array_equal(a, b)
    engine.set_pipeline(0)
    assert np.all(np.isfinite(ref[0])) and np.all(np.isfinite(ref[1]))


@pytest.mark.parametrize("name", ["magnus_n20_M6", "small_complex_M4", "nonhermitian_n24",
                                  "big_nonherm_n40", "big_n64_fullU"])
def test_segments_and_chunks_with_other_kernel_variants(engine, name):
    """Time segments / memory chunks through the Magnus kernels, the explicit-generator K3 and
    the general (non-Hermitian) kernels give bit-identical results."""
    from tests import gpu_helpers as gh
    case = cases_mod.case_by_name(name)
    gh.setup_engine(engine, case)
    u = gh.real_controls(case, np.concatenate([case.controls, 0.5 * case.controls]))
    ref = engine.evaluate(u, True)
    try:
        for chunk, pipe in ((1, 3), (3, 2), (0, 5)):
            engine.set_chunk(chunk)
            engine.set_pipeline(pipe)
            out = engine.evaluate(u, True)
            for a, b in zip(ref, out):
                assert np.array_equal(a, b), (chunk, pipe)
    finally:
        engine.set_chunk(0)
        engine.set_pipeline(0)


@pytest.mark.parametrize("name", ["c2_transmon", "c3_subset", "small_complex_M2", "small_complex_M6",
                                  "nc10_n101", "scaled_n8", "nonhermitian_n24", "magnus_n20_M4"])
def test_latency_mode_inverse_image_sweep(engine, name):
    """Latency mode (one control set at a time; knob "latency") runs the inverse-image sweep of
    qocx_sweepi.hip - P^-1 from inv_kernel, a sub-step = two matrix-vector products, fetch waves beside
    the compute wave, two-sided where the unit adjoint applies (knob "sweep_inverse"): against the
    golden fixtures at the parity gates, against the column-chain sweep, and the same numbers for
    every time segmentation."""
    from tests import gpu_helpers as gh
    case = cases_mod.case_by_name(name)
    g = golden(name)
    host_specs = gh.setup_engine(engine, case)
    host_costs = [getattr(onp, k)(**kw) for k, kw in host_specs]
    u = gh.real_controls(case, case.controls)
    try:
        engine.set_knob("latency", 1)
        engine.set_knob("sweep_inverse", 0)
        chain = engine.evaluate(u, True)
        engine.set_knob("sweep_inverse", 1)
        ref = engine.evaluate(u, True)
        for a, b in zip(ref, chain):
            assert np.max(np.abs(a - b)) <= 1e-11 * max(1.0, np.max(np.abs(b)))
        for pipe in (1, 2, 4, 7):
            engine.set_pipeline(pipe)
            out = engine.evaluate(u, True)
            for a, b in zip(ref, out):
                assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(a))), pipe
        engine.set_pipeline(0)
        fwd = engine.evaluate(u, False)
        assert np.array_equal(fwd[0], ref[0]) and np.array_equal(fwd[2], ref[2])
    finally:
        engine.set_knob("latency", 0)
        engine.set_knob("sweep_inverse", 1)
        engine.set_pipeline(0)
    cost, grads, final = ref
    grads = gh.complex_grads(case, grads)
    for b in range(len(case.controls)):
        err = cost[b]
        gb = grads[b].astype(np.complex128)
        for c in host_costs:
            err = err + c.cost(case.controls[b], None, case.N - 1)
            gb = gb + c.controls_bar(case.controls[b], None, case.N - 1)
        assert abs(err - g["error"][b]) <= 1e-10 * max(1.0, abs(g["error"][b]))
        assert rel_err(final[b][:, :, None], g["final_states"][b]) < 1e-10
        assert rel_err(gb, g["grads_ad"][b]) < 1e-8


def test_inverse_image_sweep_several_states_small_n(engine):
    """n <= 16 with up to eight states on the inverse-image sweep (qocx_sweepi.hip) in every batch:
    random problems against the oracle and against the column-chain sweep."""
    from tests import fuzz_parity
    checked = 0
    for index in range(16):
        out = []
        for inverse in (1, 0):
            engine.set_knob("sweep_inverse_small", inverse)
            rng = np.random.default_rng(6000 + index)
            worst, tag = fuzz_parity.one(engine, rng, index, nmin=2, nmax=16, smin=3, smax=8,
                                         results=out)
            if worst is None:
                break
            assert worst < 1.0, (inverse, tag)
        engine.set_knob("sweep_inverse_small", 1)
        if len(out) == 2:
            checked += 1
            for a, b in zip(*out):
                assert np.max(np.abs(a - b)) <= 1e-11 * max(1.0, np.max(np.abs(b))), tag
    assert checked >= 12


def test_dense_state_sweep(engine):
    """8 <= S <= 32 states at 17 <= n <= 32 run on the dense-state sweep (qocx_sweepd.hip: P^-1 from
    K1b's Gauss-Jordan sibling, two MFMA GEMMs per sub-step over all states; knob "sweep_dense"):
    random problems - step costs of every kind, cost_eval_step, squarings, Hermitian or not, time
    dependent or not, all Magnus policies - against the oracle at the parity gates, against the
    column-chain sweep, and bit-identical across time segments and memory chunks."""
    from tests import fuzz_parity
    checked = 0
    for index in range(24):
        out = []
        # dense sweep + K3 on the matrix cores | dense sweep + vector-unit K3 (the default) | neither
        for dense, k3 in ((1, 1), (1, 0), (0, 0)):
            engine.set_knob("sweep_dense", dense)
            engine.set_knob("krylov_dense", k3)
            rng = np.random.default_rng(5000 + index)
            worst, tag = fuzz_parity.one(engine, rng, index, nmin=17, nmax=32, smin=8, smax=32,
                                         results=out)
            if worst is None:
                break
            assert worst < 1.0, (dense, k3, tag)
        engine.set_knob("sweep_dense", 1)
        engine.set_knob("krylov_dense", 0)
        if len(out) == 3:
            checked += 1
            for other in out[:2]:
                for a, b in zip(other, out[2]):
                    assert np.max(np.abs(a - b)) <= 1e-11 * max(1.0, np.max(np.abs(b))), tag
    assert checked >= 18
    case = cases_mod.case_by_name("c3_fullU_short")
    from tests import gpu_helpers as gh
    gh.setup_engine(engine, case)
    u = gh.real_controls(case, np.concatenate([case.controls, 0.5 * case.controls, -case.controls]))
    ref = engine.evaluate(u, True)
    try:
        for chunk, pipe in ((1, 3), (2, 4), (0, 8)):
            engine.set_chunk(chunk)
            engine.set_pipeline(pipe)
            out = engine.evaluate(u, True)
            for a, b in zip(ref, out):
                assert np.array_equal(a, b), (chunk, pipe)
    finally:
        engine.set_chunk(0)
        engine.set_pipeline(0)


def test_random_shapes_fuzz(engine):
    """tests/fuzz_parity.py: 80 random problems (sizes, grids, Magnus policies, Hermitian or not,
    time dependent or not, 0..4 squarings) against the oracle at the parity tolerances."""
    from tests import fuzz_parity
    rng = np.random.default_rng(2024)
    checked = 0
    for index in range(80):
        worst, tag = fuzz_parity.one(engine, rng, index)
        if worst is None:
            continue
        checked += 1
        assert worst < 1.0, tag
    assert checked > 60


def test_random_shapes_fuzz_big(engine):
    """The same for 33 <= n <= 64 (M2): 24 random problems against the oracle."""
    from tests import fuzz_parity
    rng = np.random.default_rng(6464)
    checked = 0
    for index in range(24):
        worst, tag = fuzz_parity.one(engine, rng, index, nmin=33, nmax=64)
        if worst is None:
            continue
        checked += 1
        assert worst < 1.0, tag
    assert checked > 15


@pytest.mark.parametrize("n", [36, 48, 56, 64])
def test_hermitian_tiles_of_the_four_wave_pade_kernel(engine, n):
    """Round 4: for Hermitian generators whose norm bound is below theta_9 the four-wave K1a forms two
    thirds of the tiles of every product and mirrors the rest (knob "k1a_herm4"). Every order the path
    contains - 3, 5, 7, 9, chosen through the time step - against the kernel that forms all tiles (knob
    0), at rounding level, and against the oracle at the parity gate."""
    import bench
    from qoc_amd.engine import COST_TARGET_COHERENT
    rng = np.random.default_rng(3600 + n)
    h0 = bench.gue(rng, n)
    g = [bench.gue(rng, n) for _ in range(2)]
    psi0 = np.eye(n, dtype=np.complex128)[:1]
    target = np.eye(n, dtype=np.complex128)[1:2]
    N = 9
    u = 0.2 * rng.standard_normal((3, N, 2))
    # the host's bound of ||H||_1 (it picks the kernel: below theta_9 / dt the Hermitian-tile build runs)
    h_bound = onp.one_norm(h0) + sum(np.max(np.abs(u[..., k])) * onp.one_norm(g[k]) for k in range(2))
    seen = set()
    for theta in (0.012, 0.2, 0.8, 2.0):  # bounds below theta_3, theta_5, theta_7, theta_9
        dt = theta / h_bound
        engine.set_schroedinger_problem(
            n, 1, 2, N, N, dt * (N - 1), h0[None], np.stack(g)[None], psi0,
            costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
        try:
            engine.set_knob("k1a_herm4", 0)
            c0, g0, f0 = engine.evaluate(u, True)
            engine.set_knob("k1a_herm4", 1)
            c1, g1, f1 = engine.evaluate(u, True)
            orders = engine.pade_orders()
        finally:
            engine.set_knob("k1a_herm4", 1)
        seen |= {o for o, cnt in orders.items() if cnt}
        assert orders[13] == 0
        assert np.max(np.abs(c1 - c0)) < 1e-13
        assert np.max(np.abs(f1 - f0)) < 1e-13
        assert np.max(np.abs(g1 - g0)) < 1e-12 * max(1.0, np.max(np.abs(g0)))
        # the oracle's forward pass (expm of the same generators, the reference's order)
        for b in range(u.shape[0]):
            psi = psi0[0].copy()
            for j in range(N - 1):
                um = 0.5 * (u[b, j] + u[b, j + 1])  # (control knots at the system times: the mid point)
                psi = onp.expm_pade(-1j * dt * (h0 + sum(c * gk for c, gk in zip(um, g)))) @ psi
            assert np.max(np.abs(f1[b, 0] - psi)) < 1e-11
    assert seen == {3, 5, 7, 9}, seen


def test_magnus_above_n32_on_the_lds_resident_kernels(engine):
    """Round 4: MagnusPolicy.M4 / M6 above n = 32 run the LDS-resident multi-wave kernels
    (qocx_magnus4w.hip: three waves and four matrices in LDS for n <= 48, four waves and two matrices
    at a time for n <= 64) instead of the one-wave kernels, whose sixteen-tile matrices live in
    scratch memory. Random problems of both size classes under M4 / M6 - Hermitian and not, with and
    without a time-dependent H0 - against the oracle, and against the one-wave kernels (knob
    "magnus_4w" 0) at rounding level."""
    from tests import fuzz_parity
    for nmin, nmax, seed in ((33, 48, 4801), (49, 64, 6401)):
        rng = np.random.default_rng(seed)
        seen = set()
        for index in range(40):
            results = []
            state = rng.bit_generator.state
            worst, tag = fuzz_parity.one(engine, rng, index, nmin=nmin, nmax=nmax, smin=1, smax=2, results=results)
            policy = [tok for tok in tag.split() if tok in ("M2", "M4", "M6")][0]
            if worst is None or policy == "M2":
                continue
            assert worst < 1.0, tag
            # the same problem on the one-wave kernels
            engine.set_knob("magnus_4w", 0)
            try:
                rng2 = np.random.default_rng(0)
                rng2.bit_generator.state = state
                results2 = []
                worst2, tag2 = fuzz_parity.one(engine, rng2, index, nmin=nmin, nmax=nmax, smin=1, smax=2,
                                               results=results2)
            finally:
                engine.set_knob("magnus_4w", 1)
            assert tag2 == tag and worst2 < 1.0
            for new, old in zip(results[0], results2[0]):
                scale = max(1.0, float(np.max(np.abs(old))))
                assert np.max(np.abs(np.asarray(new) - np.asarray(old))) < 1e-11 * scale, tag
            seen.add(policy)
            if seen == {"M4", "M6"} and index >= 8:
                break
        assert seen == {"M4", "M6"}, (nmin, seen)


def test_product_library_rejects_diagnostic_knobs(engine):
    """The timing experiments whose results are garbage are not reachable through libqocx.so."""
    from qoc_amd.engine import QocxError
    for name in ("dbg_skip", "sweep3_dbg", "k1a_dbg", "sweep3_stamps"):
        with pytest.raises(QocxError):
            engine.set_knob(name, 1)
    engine.set_knob("pade_order", 0)  # a variant knob is accepted


@pytest.mark.parametrize("n", [20, 32, 40, 48, 64])
def test_mfma_factorisation_is_the_path_taken(engine, n):
    """The LU with its Schur updates on the matrix cores (qocx_lu4.h inside the two-wave K1a, lu9 /
    lu4m kernels for n > 32) takes its pivots on the diagonal speculatively and hands a matrix whose
    pivots leave it to the general elimination - a silent net under a broken fast path. So: on the Pade
    denominators of well-scaled generators NO matrix may fall back (qocx_lu_fallbacks), on generators
    scaled to the edge of the squaring threshold some must, and both give the model's factors."""
    rng = np.random.default_rng(500 + n)

    def batch(scale, count=24, skew=True):
        out = []
        for _ in range(count):
            g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
            a = -1j * (g + g.conj().T) / 2 if skew else g
            out.append(a * (scale / onp.one_norm(a)))
        return np.stack(out)

    small = batch(0.2)
    out = engine.debug_pade_factor(small)
    assert engine.lu_fallbacks() == 0
    check_pade_factor(out, small, 0, expect_lower=False)
    # generators with a few dominant sub-diagonal entries: P = b0 (I - a / 2 + ...) then has columns
    # whose largest entry sits below the diagonal, and LAPACK's rule interchanges rows
    large = 0.01 * batch(1.0, skew=False)
    for m in range(len(large)):
        for i in (0, n // 2, n - 2):
            large[m, i + 1, i] += 4.8 * np.exp(1j * rng.uniform(0, 2 * np.pi))
    out = engine.debug_pade_factor(large)
    assert engine.lu_fallbacks() > 0
    check_pade_factor(out, large, 0, expect_lower=False)


def test_headline_evaluation_takes_no_fallback(engine):
    """bench.py's workload at a reduced seed count: every one of the factorisations stays on the
    diagonal-pivot MFMA path."""
    import bench
    from qoc_amd.engine import COST_TARGET_COHERENT
    h0, g, psi0, target = bench.make_problem()
    engine.set_schroedinger_problem(
        bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
        h0[None], np.stack(g)[None], psi0,
        costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
    engine.evaluate(bench.make_controls(0, 8), want_grad=True)
    assert engine.lu_fallbacks() == 0
    orders = engine.pade_orders()
    assert orders[5] == 8 * (bench.N_EVAL - 1)
