"""Shared helpers for the tests: golden loading and case -> oracle/product objects."""

import os

import numpy as np

from oracle import qoc_numpy as onp
from tests import cases as cases_mod

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def oracle_costs(case):
    return [getattr(onp, kind)(**kw) for kind, kw in case.cost_specs]


def oracle_problem(case):
    return onp.SchroedingerProblem(
        case.T, case.hamiltonian(), case.initial_states, case.N,
        control_eval_count=case.Nc, costs=oracle_costs(case),
        cost_eval_step=case.cost_eval_step, magnus_policy=case.magnus,
        complex_controls=case.complex_controls, control_count=case.K)


def rel_err(x, ref):
    ref = np.asarray(ref)
    scale = np.max(np.abs(ref))
    return np.max(np.abs(np.asarray(x) - ref)) / (scale if scale > 0 else 1.0)


CASE_NAMES = [c.name for c in cases_mod.all_cases()]
GRAD_CASE_NAMES = [c.name for c in cases_mod.all_cases() if c.controls is not None]


def lindblad_grad_close(grads, ref, case=None):
    """
    Gradient gate of the Lindblad fixtures.

    * Fixtures with a well-conditioned gradient (`case.grad_rtol`, max |g| >= 1e-2:
      lindblad_wc_*): **1e-8 of the largest entry**, no floor - north_star's bar.
    * The round-1 fixtures aim at random densities (cost ~0.99, max |g| = 4e-6 .. 8e-3): 1e-6 of
      the largest entry, with an absolute floor of 5e-10. The floor is the reference's own
      noise: its adaptive integrator reproduces its forward result only to ~4e-9 over the 500
      steps of the full-length fixture (tools/gen_golden_lindblad.py prints it), which shows up
      at ~1e-10 in a gradient of size 1e-5.
    """
    import numpy as np
    dev = np.max(np.abs(grads - ref))
    rtol = getattr(case, "grad_rtol", None)
    if rtol is not None:
        return dev < rtol * np.max(np.abs(ref))
    return dev < max(1e-6 * np.max(np.abs(ref)), 5e-10)


_REAL_MAKE_BACKEND = None


def set_backend_factory(factory):
    """
    Tests only: run the product's HOST logic (entry points, structure probing, optimizers, save
    files) on a machine without a GPU by swapping qoc_amd.core.device.make_backend for a NumPy
    model of the device (tests/oracle_backend.py). The product itself has no hook and no CPU
    fallback; this patches the module attribute from outside. None restores the real engine.
    """
    global _REAL_MAKE_BACKEND
    from qoc_amd.core import device
    if _REAL_MAKE_BACKEND is None:
        _REAL_MAKE_BACKEND = device.make_backend
    device.make_backend = (lambda device=-1: factory()) if factory is not None else _REAL_MAKE_BACKEND
