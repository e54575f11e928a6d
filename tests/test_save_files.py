"""
CPU tests of the save-file side effects of the four entry points (dataset names and shapes of
qoc/models/schroedingermodels.py:66-95, :240-308 and lindbladmodels.py:60-90, :254-309) with a
dictionary-backed stand-in for h5py (tests/fake_h5py.py; h5py is not installed here). The
engine is the oracle backend.
"""

import sys

import numpy as np
import pytest

import qoc_amd
import qoc_amd.standard.costs as product_costs
from qoc_amd.core import device
from qoc_amd.standard import Adam
from tests import cases as cases_mod
from tests import helpers
from tests import fake_h5py
from tests.oracle_backend import OracleBackend


@pytest.fixture(autouse=True)
def environment(monkeypatch):
    monkeypatch.setitem(sys.modules, "h5py", fake_h5py)
    fake_h5py.STORE.clear()
    helpers.set_backend_factory(OracleBackend)
    yield
    helpers.set_backend_factory(None)


def costs_of(case):
    return [getattr(product_costs, kind)(**kw) for kind, kw in case.cost_specs]


def test_evolve_schroedinger_save(tmp_path, capsys):
    case = cases_mod.case_by_name("ctrlcosts_r")
    path = str(tmp_path / "evolve.h5")
    result = qoc_amd.evolve_schroedinger_discrete(
        case.T, case.hamiltonian(), case.initial_states, case.N, controls=case.controls[0],
        costs=costs_of(case), save_file_path=path, save_intermediate_states=True)
    assert "QOC is saving this evolution to {}.".format(path) in capsys.readouterr().out
    f = fake_h5py.STORE[path]
    assert str(f["method"].array) == "evolve_schroedinger_discrete"
    assert f["program_type"].array == 1 and f["system_eval_count"].array == case.N
    assert np.array_equal(f["controls"].array, case.controls[0])
    inter = f["intermediate_states"].array
    assert inter.shape == (case.N,) + case.initial_states.shape
    assert np.array_equal(inter[0], case.initial_states)
    assert np.allclose(inter[-1], result.final_states, atol=1e-14)
    assert str(f["magnus_policy"].array) == "magnus_m2"


def test_grape_schroedinger_save(tmp_path, capsys):
    case = cases_mod.case_by_name("ctrlcosts_r")
    path = str(tmp_path / "grape.h5")
    result = qoc_amd.grape_schroedinger_discrete(
        case.K, case.Nc, costs_of(case), case.T, case.hamiltonian(), case.initial_states, case.N,
        initial_controls=case.controls[0], iteration_count=5, log_iteration_step=0,
        optimizer=Adam(learning_rate=1e-2), max_control_norms=np.array([5.0, 5.0]),
        save_file_path=path, save_intermediate_states=True, save_iteration_step=2)
    assert "QOC is saving this optimization run to {}.".format(path) in capsys.readouterr().out
    f = fake_h5py.STORE[path]
    # iterations 0, 2, 4 are saved: ceil(5 / 2) = 3 rows
    assert f["controls"].array.shape == (3, case.Nc, case.K)
    assert f["error"].array.shape == (3,) and np.all(f["error"].array < 10)
    assert f["final_states"].array.shape == (3,) + case.initial_states.shape
    assert f["grads"].array.shape == (3, case.Nc, case.K)
    assert f["intermediate_states"].array.shape == (3, case.N) + case.initial_states.shape
    assert np.array_equal(f["controls"].array[0], case.controls[0])
    assert np.array_equal(f["initial_controls"].array, case.controls[0])
    assert abs(f["error"].array[0] - result.best_error) >= 0  # recorded, finite
    assert [bytes(x).decode() for x in f["cost_names"].array][0] == "target_state_infidelity"
    assert str(f["optimizer"].array).startswith("adam, beta_1: 0.9") and f["iteration_count"].array == 5
    # the saved error of a saved iteration is the error of its saved controls
    check = qoc_amd.evolve_schroedinger_discrete(
        case.T, case.hamiltonian(), case.initial_states, case.N,
        controls=f["controls"].array[2], costs=costs_of(case))
    assert abs(check.error - f["error"].array[2]) < 1e-12


def test_lindblad_saves(tmp_path, capsys):
    case = cases_mod.lindblad_case_by_name("lindblad_n4")
    path = str(tmp_path / "evolve_l.h5")
    result = qoc_amd.evolve_lindblad_discrete(
        case.T, case.initial_densities, case.N, controls=case.controls[0],
        cost_eval_step=case.cost_eval_step, costs=costs_of(case),
        hamiltonian=case.hamiltonian(), lindblad_data=case.lindblad_data(),
        save_file_path=path, save_intermediate_densities=True)
    f = fake_h5py.STORE[path]
    assert str(f["method"].array) == "evolve_lindblad_discrete"
    inter = f["intermediate_densities"].array
    assert inter.shape == (case.N,) + case.initial_densities.shape
    assert np.array_equal(inter[0], case.initial_densities)
    assert np.allclose(inter[-1], result.final_densities, atol=1e-13)
    path = str(tmp_path / "grape_l.h5")
    qoc_amd.grape_lindblad_discrete(
        case.K, case.Nc, costs_of(case), case.T, case.initial_densities, case.N,
        cost_eval_step=case.cost_eval_step, hamiltonian=case.hamiltonian(),
        lindblad_data=case.lindblad_data(), initial_controls=case.controls[0],
        iteration_count=3, log_iteration_step=0, max_control_norms=np.array([5.0, 5.0]),
        save_file_path=path, save_intermediate_densities=True, save_iteration_step=1)
    capsys.readouterr()
    f = fake_h5py.STORE[path]
    assert f["final_densities"].array.shape == (3,) + case.initial_densities.shape
    assert f["intermediate_densities"].array.shape == (3, case.N) + case.initial_densities.shape
    assert str(f["method"].array) == "grape_lindblad_discrete"
    assert np.all(np.abs(np.trace(f["final_densities"].array, axis1=-2, axis2=-1) - 1) < 1e-10)
    assert np.all(np.diff(f["error"].array) < 0)  # Adam on a smooth problem: monotone here
