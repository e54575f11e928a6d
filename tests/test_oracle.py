"""
CPU tests: the oracle (oracle/qoc_numpy.py) against the golden vectors minted from the
reference (tools/gen_golden.py) and against the reference's analytic known answers.
Tolerances: forward 1e-10 relative (SURVEY.md 8d parity gates), gradients 1e-8 relative to
the AD fixture and 1e-7 to the finite-difference fixture of the reference forward.
"""

import numpy as np
import pytest

from oracle import qoc_numpy as onp
from tests import cases as cases_mod
from tests import device_model as dm
from tests.helpers import (CASE_NAMES, GRAD_CASE_NAMES, golden, oracle_problem, rel_err)


def test_expm_golden():
    g = golden("units")
    for i in range(int(g["expm_count"])):
        out = onp.expm_pade(g["expm_in_%d" % i])
        assert rel_err(out, g["expm_out_%d" % i]) < 1e-13


def test_expm_scale_counts():
    # expm.py:238-241: s = max(0, ceil(log2(norm / theta13))), none below theta13
    t = onp.THETA_13
    assert onp.pade_scale_count(0.0) == 0
    assert onp.pade_scale_count(t * 0.999) == 0
    assert onp.pade_scale_count(t) == 0
    assert onp.pade_scale_count(t * 1.001) == 1
    assert onp.pade_scale_count(t * 2.0) == 1
    assert onp.pade_scale_count(t * 2.001) == 2
    assert onp.pade_scale_count(t * 17) == 5


def test_expm_vs_scipy_and_unitarity():
    import scipy.linalg as la
    rng = np.random.default_rng(5)
    for n, scale in [(8, 0.5), (32, 4.0), (32, 60.0)]:
        h = cases_mod.gue(rng, n) * scale
        u = onp.expm_pade(-1j * h)
        assert rel_err(u, la.expm(-1j * h)) < 1e-12
        assert np.max(np.abs(u.conj().T @ u - np.eye(n))) < 1e-12


def test_interpolation_golden():
    g = golden("units")
    out = np.stack([onp.interpolate_linear_set(x, g["interp_xs"], g["interp_ys"])
                    for x in g["interp_xq"]])
    assert np.array_equal(out, g["interp_out"])
    # weights reproduce the formula
    for x in g["interp_xq"]:
        i1, w1, i2, w2 = onp.interpolation_weights(x, g["interp_xs"])
        y = w1 * g["interp_ys"][i1] + w2 * g["interp_ys"][i2]
        assert np.allclose(y, onp.interpolate_linear_set(x, g["interp_xs"], g["interp_ys"]),
                           rtol=0, atol=1e-13)


@pytest.mark.parametrize("policy", ["M2", "M4", "M6"])
def test_magnus_golden(policy):
    g = golden("units")
    m0, m1, m2 = g["magnus_m0"], g["magnus_m1"], g["magnus_m2"]
    gen = lambda t: m0 + t * m1 + np.sin(3 * t) * m2
    dt, t = float(g["magnus_dt"]), float(g["magnus_t"])
    gens = [gen(t + dt * c) for c in onp.MAGNUS_NODES[policy]]
    m, _ = onp.magnus_combine(policy, dt, gens)
    assert rel_err(m, g["magnus_out_" + policy]) < 1e-14


def test_magnus_identity():
    # reference tests/test_core.py:337-349
    for policy in ("M2", "M4", "M6"):
        gens = [np.eye(3) for _ in onp.MAGNUS_NODES[policy]]
        m, _ = onp.magnus_combine(policy, 2.5, gens)
        assert np.allclose(m, 2.5 * np.eye(3))


def test_clip_strip_slap_golden():
    g = golden("units")
    cr, cc = g["clip_in_real"].copy(), g["clip_in_complex"].copy()
    onp.clip_control_norms(cr, g["clip_norms"])
    onp.clip_control_norms(cc, g["clip_norms"])
    assert np.array_equal(cr, g["clip_out_real"])
    assert np.array_equal(cc, g["clip_out_complex"])
    flat = onp.strip_controls(True, g["clip_in_complex"])
    assert np.array_equal(flat, g["strip_complex"])
    assert np.array_equal(onp.slap_controls(True, flat, (6, 2)), g["slap_complex"])


def test_cost_known_answers():
    # reference tests/test_standard.py:70-90 (ForbidStates = 5/80)
    system_eval_count = 11
    state0 = np.array([[1], [0]])
    forbid0_0 = np.array([[1], [0]])
    state1 = np.array([[0], [1]])
    forbid1_0 = np.divide(np.array([[1], [1]]), np.sqrt(2))
    forbid1_1 = np.divide(np.array([[1j], [1j]]), np.sqrt(2))
    states = np.stack((state0, state1,))
    forbidden = np.array([[forbid0_0, forbid0_0], [forbid1_0, forbid1_1]])
    fs = onp.ForbidStates(forbidden, system_eval_count)
    assert np.allclose(fs.cost(None, states, 0), 0.75 / 10)
    # tests/test_standard.py:166-191 (TargetStateInfidelity); the third expectation of the
    # reference test (0.25) is the neglect_relative_pahse=True value -- SURVEY.md section 4.
    s0 = np.array([[0], [1]])
    t0 = np.array([[1], [0]])
    assert np.allclose(onp.TargetStateInfidelity(np.stack((t0,))).cost(None, np.stack((s0,)), None), 1)
    assert np.allclose(onp.TargetStateInfidelity(np.stack((s0,))).cost(None, np.stack((s0,)), None), 0)
    s0 = np.array([[1], [0]])
    s1 = (np.array([[1j], [1]]) / np.sqrt(2))
    t0 = np.array([[1j], [0]])
    t1 = np.array([[1], [1]]) / np.sqrt(2)
    both = np.stack((s0, s1,))
    tb = np.stack((t0, t1,))
    assert np.allclose(onp.TargetStateInfidelity(tb, neglect_relative_pahse=True)
                       .cost(None, both, None), 0.25)
    assert np.allclose(onp.TargetStateInfidelityTime(11, tb, neglect_relative_pahse=True)
                       .cost(None, both, None), 0.025)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_forward_matches_reference(name):
    case = cases_mod.case_by_name(name)
    g = golden(name)
    problem = oracle_problem(case)
    controls = [None] if case.controls is None else list(case.controls)
    for b, u in enumerate(controls):
        err, final = onp.evaluate(problem, u)
        assert abs(err - g["error"][b]) <= 1e-12 * max(1.0, abs(g["error"][b]))
        assert rel_err(final, g["final_states"][b]) < 1e-12


def test_iswap_known_answer():
    target = np.array(((1, 0, 0, 0), (0, 0, -1j, 0), (0, -1j, 0, 0), (0, 0, 0, 1)))
    for policy in ("M2", "M4", "M6"):
        case = cases_mod.case_iswap(policy)
        _, final = onp.evaluate(oracle_problem(case), None)
        assert np.allclose(final, cases_mod.column_states(target))


@pytest.mark.parametrize("name", GRAD_CASE_NAMES)
def test_gradient_matches_fixtures(name):
    case = cases_mod.case_by_name(name)
    g = golden(name)
    problem = oracle_problem(case)
    count = 1 if name == "c3_subset" else len(case.controls)  # keep the CPU suite short
    for b in range(count):
        err, grads, final = onp.evaluate_with_grad(problem, case.controls[b])
        assert abs(err - g["error"][b]) <= 1e-12 * max(1.0, abs(g["error"][b]))
        assert grads.dtype == case.controls.dtype
        assert rel_err(grads, g["grads_ad"][b]) < 1e-8
        scale = np.max(np.abs(g["grads_ad"][b]))
        fd_dev = np.max(np.abs(grads.flat[g["fd_index"][b]] - g["grads_fd"][b])) / scale
        assert fd_dev < 1e-7


def _pade_denominator(a, order):
    """P = v - u of the [order/order] approximant (expm.py:119-159, :246), from the power series."""
    b = dm.PADE_COEFFS[order]
    n = a.shape[0]
    p, power = np.zeros((n, n), dtype=np.complex128), np.eye(n, dtype=np.complex128)
    for j, bj in enumerate(b):
        p += ((-1) ** j) * bj * power
        power = power @ a
    return p


def _eps(order, theta):
    b = dm.PADE_COEFFS[order]
    return sum(b[j] / b[0] * theta ** j for j in range(1, order + 1))


@pytest.mark.parametrize("order", [3, 5, 7, 9, 13])
def test_dominant_pade_denominators_pivot_on_the_diagonal(order):
    """
    qoc_amd/csrc/qocx_lu5.h factors the Pade denominator WITHOUT looking for pivots when the bound
    eps(theta) = sum_{j>=1} (b_j / b0) theta^j of ||P / b0 - I||_1 is at most 0.40 (theta: an upper
    bound of the 1-norm of the scaled generator): below 1 / (1 + sqrt 2) LAPACK's zgetrf - what
    numpy.linalg.solve runs at qoc/standard/functions/expm.py:246 - provably takes the diagonal entry at
    every step. Checked here against LAPACK itself (scipy.linalg.lu_factor): generators whose norm sits
    right at the threshold - Hermitian, general complex, and adversarial ones that concentrate a whole
    column of P on one off-diagonal entry with |re| = |im| - never see a row exchange; and the
    condition is not vacuous: far above the threshold LAPACK does exchange rows.
    """
    from scipy.linalg import lu_factor
    rng = np.random.default_rng(500 + order)
    # the largest theta the kernel accepts for this order
    lo, hi = 0.0, 8.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if _eps(order, mid) <= 0.40 else (lo, mid)
    theta = min(lo, dm.PADE_THETA[order])
    assert _eps(order, theta) <= 0.40
    if order == 5:  # every order-5 step qualifies: eps(theta_5) = 0.134
        assert theta == dm.PADE_THETA[5] and _eps(5, theta) < 0.135
    for n in (2, 7, 32):
        for trial in range(40):
            g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
            kind = trial % 4
            if kind == 0:
                a = -1j * (g + g.conj().T)                      # skew-Hermitian (the headline's kind)
            elif kind == 1:
                a = g                                           # general complex
            elif kind == 2:                                     # one entry carries a whole column
                a = np.zeros((n, n), dtype=np.complex128)
                a[(np.arange(n) + 1) % n, np.arange(n)] = (1 + 1j) * rng.choice([-1, 1], n)
            else:                                               # nilpotent: the series does not cancel
                a = np.triu(np.abs(g.real) + 1j * np.abs(g.imag), 1)
            a = a * (theta / onp.one_norm(a))
            p = _pade_denominator(a, order)
            assert onp.one_norm(p / dm.PADE_COEFFS[order][0] - np.eye(n)) <= 0.40 * (1 + 1e-12)
            _, piv = lu_factor(p)
            assert np.array_equal(piv, np.arange(n)), (n, trial, kind)
    # ... and above the threshold the rule matters (order 13, the only one whose theta reaches there):
    # a rotation generator a = t [[0, 1], [-1, 0]] has P = v(t) I - w(t) a with v ~ cos-like and
    # w ~ sin-like polynomials, and where |w t| > |v| LAPACK takes the second row first
    if order == 13:
        exchanged = 0
        for t in np.linspace(0.5, 5.3, 49):
            a = t * np.array([[0, 1], [-1, 0]], dtype=np.complex128)
            _, piv = lu_factor(_pade_denominator(a, 13))
            if not np.array_equal(piv, np.arange(2)):
                exchanged += 1
                assert not _eps(13, onp.one_norm(a)) <= 0.40
        assert exchanged > 0
