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
