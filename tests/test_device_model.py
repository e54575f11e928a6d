"""
CPU tests: the NumPy model of the device algorithm (tests/device_model.py: LU-factored Pade
step, serial solve sweeps, Krylov-chain adjoint) against the oracle's dense adjoint and the
golden fixtures.  Locks the math of the HIP kernels independently of the device code.
"""

import numpy as np
import pytest

from oracle import qoc_numpy as onp
from tests import cases as cases_mod
from tests import device_model as dm
from tests.helpers import golden, oracle_problem, rel_err

M2_GRAD_CASES = [c.name for c in cases_mod.all_cases()
                 if c.controls is not None and c.magnus == "M2" and c.name != "c3_subset"]


def test_lu_and_solves():
    rng = np.random.default_rng(0)
    for n in (3, 8, 32):
        p = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        y = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
        lu, perm = dm.lu_partial_pivot(p)
        assert rel_err(dm.solve_lu(lu, perm, y), np.linalg.solve(p, y)) < 1e-11
        assert rel_err(dm.solve_lu_adjoint(lu, perm, y), np.linalg.solve(p.conj().T, y)) < 1e-11


def test_krylov_adjoint_equals_dense_vjp():
    """abar from Krylov chains == expm_pade_vjp(rbar = lam psi^H) for s = 0 and s > 0, and for
    every Pade order the device selects by norm (device_model.PADE_THETA): the [m/m] approximant
    below theta_m and ITS Krylov adjoint against the reference's [13/13] formulas and their dense
    reverse rule - the same matrix and the same derivative to rounding."""
    rng = np.random.default_rng(1)
    for n, scale in ((8, 2.0), (8, 30.0), (16, 9.0), (8, 0.01), (12, 0.2), (12, 0.9), (16, 5.0),
                     (32, 0.15)):
        a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        a = a * (scale / onp.one_norm(a))
        psi = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
        lam = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
        r, cache = onp.expm_pade_cached(a)
        dense = onp.expm_pade_vjp(cache, lam @ psi.conj().T)
        f = dm.pade_factor(a)
        assert f["order"] == dm.pade_order(scale)
        subs = [psi]
        for _ in range(2 ** f["s"]):
            subs.append(dm.solve_lu(f["lu"], f["perm"], f["q"] @ subs[-1]))
        assert rel_err(subs[-1], r @ psi) < (1e-11 if f["order"] == 13 else 1e-14)
        triples, lam_m = [], lam
        for m in range(2 ** f["s"] - 1, -1, -1):
            x = dm.solve_lu_adjoint(f["lu"], f["perm"], lam_m)
            lam_m = f["q"].conj().T @ x
            triples.append((x, subs[m], subs[m + 1]))
        assert rel_err(lam_m, r.conj().T @ lam) < 1e-11
        abar = dm.krylov_abar(f["a"], triples, f["order"]) * (2 ** -f["s"])
        assert rel_err(abar, dense) < (1e-10 if f["order"] == 13 else 1e-13)
        # the order the device kernel works in (Horner recurrence for the rho vectors)
        abar_h = dm.krylov_abar_horner(f["a"], triples, f["order"]) * (2 ** -f["s"])
        assert rel_err(abar_h, dense) < 1e-10
        assert rel_err(abar_h, abar) < 1e-12


@pytest.mark.parametrize("name", M2_GRAD_CASES)
def test_model_matches_oracle_and_fixtures(name):
    case = cases_mod.case_by_name(name)
    g = golden(name)
    problem = oracle_problem(case)
    for b in range(len(case.controls)):
        err, grads, final = dm.evaluate_with_grad(problem, case.controls[b])
        assert abs(err - g["error"][b]) <= 1e-11 * max(1.0, abs(g["error"][b]))
        assert rel_err(final, g["final_states"][b]) < 1e-10
        assert rel_err(grads, g["grads_ad"][b]) < 1e-8
