"""
CPU tests of the host-side mirror of the reference interface: entry points, Cost / Optimizer
plugins, control plumbing, Hamiltonian structure extraction. The GPU engine is replaced by the
oracle through the test hook tests.helpers.set_backend_factory (tests/oracle_backend.py);
the same entry points run on the real engine in tests/test_gpu_api.py.
"""

import numpy as np
import pytest

import qoc_amd
from qoc_amd.core import common, device, structure
from qoc_amd.models import (Cost, InterpolationPolicy, MagnusPolicy, OperationPolicy,
                            PerformancePolicy, ProgramType)
from qoc_amd.standard import (SIGMA_X, SIGMA_Y, Adam, LBFGSB, SGD, ControlNorm, ForbidStates,
                              TargetStateInfidelity, TargetStateInfidelityTime,
                              get_annihilation_operator, get_creation_operator)
import qoc_amd.standard.costs as product_costs
from oracle import qoc_numpy as onp
from tests import cases as cases_mod
from tests import helpers
from tests.helpers import golden, rel_err
from tests.oracle_backend import OracleBackend


@pytest.fixture(autouse=True)
def oracle_engine():
    helpers.set_backend_factory(OracleBackend)
    yield
    helpers.set_backend_factory(None)


def product_cost_list(case):
    return [getattr(product_costs, kind)(**kw) for kind, kw in case.cost_specs]


# ---- models / plugins ---------------------------------------------------------------------------

def test_policy_strings():
    assert str(MagnusPolicy.M2) == "magnus_m2" and repr(MagnusPolicy.M6) == "magnus_m6"
    assert str(InterpolationPolicy.LINEAR) == "interpolation_linear"
    assert str(ProgramType.EVOLVE) == "evolve" and str(ProgramType.GRAPE) == "grape"
    assert str(OperationPolicy.GPU) == "operation_policy_gpu"
    assert str(PerformancePolicy.MEMORY) == "performance_policy_memory"
    assert MagnusPolicy.M4.value == 2


def test_cost_base_class():
    c = Cost(cost_multiplier=2.5)
    assert c.cost_multiplier == 2.5 and str(c) == "parent_cost" and not c.requires_step_evaluation
    with pytest.raises(NotImplementedError):
        c.cost(None, None, 0)


def test_state_cost_known_answers():
    # reference tests/test_standard.py:70-90, :166-223 (stale third cases keyed by the flag)
    state0, state1 = np.array([[1], [0]]), np.array([[0], [1]])
    forbid1_0 = np.array([[1], [1]]) / np.sqrt(2)
    forbid1_1 = np.array([[1j], [1j]]) / np.sqrt(2)
    forbidden = np.array([[state0, state0], [forbid1_0, forbid1_1]])
    fs = ForbidStates(forbidden, 11)
    assert np.allclose(fs.cost(None, np.stack((state0, state1)), 0), 0.75 / 10)
    t0 = np.array([[1], [0]])
    assert np.allclose(TargetStateInfidelity(np.stack((t0,))).cost(None, np.stack((state1,)), None), 1)
    assert np.allclose(TargetStateInfidelity(np.stack((state1,))).cost(None, np.stack((state1,)), None), 0)
    s1 = np.array([[1j], [1]]) / np.sqrt(2)
    tb = np.stack((np.array([[1j], [0]]), np.array([[1], [1]]) / np.sqrt(2)))
    both = np.stack((state0, s1))
    assert np.allclose(TargetStateInfidelity(tb, neglect_relative_pahse=True).cost(None, both, None), 0.25)
    assert np.allclose(TargetStateInfidelityTime(11, tb, neglect_relative_pahse=True).cost(None, both, None), 0.025)
    coherent = TargetStateInfidelity(tb).cost(None, both, None)
    assert np.allclose(coherent, onp.TargetStateInfidelity(tb).cost(None, both, None))
    assert np.allclose(TargetStateInfidelityTime(11, tb).cost(None, both, None), coherent / 10)


@pytest.mark.parametrize("complex_controls", [False, True])
def test_control_costs_value_and_gradient(complex_controls):
    case = cases_mod.case_control_costs(complex_controls)
    rng = np.random.default_rng(3)
    for kind, kw in case.cost_specs[1:]:
        mine = getattr(product_costs, kind)(**kw)
        ref = getattr(onp, kind)(**kw)
        u = case.controls[0]
        assert np.allclose(mine.cost(u, None, 0), ref.cost(u, None, 0), rtol=1e-14, atol=0)
        bar = mine.controls_bar(u, None, 0)
        assert bar.dtype == u.dtype
        assert rel_err(bar, ref.controls_bar(u, None, 0)) < 1e-13
        # directional finite difference of the product's own cost()
        d = rng.standard_normal(u.shape)
        if complex_controls:
            d = d + 1j * rng.standard_normal(u.shape)
        h = 1e-6
        fd = (mine.cost(u + h * d, None, 0) - mine.cost(u - h * d, None, 0)) / (2 * h)
        assert abs(fd - np.sum(np.real(np.conj(bar) * d))) < 1e-8
    with pytest.raises(NameError):
        product_costs.ControlArea(2, 5).cost(np.ones((5, 2)), None, 0)


def test_adam_known_answers():
    # reference tests/test_standard.py:252-276
    adam = Adam()
    grads = np.array([[0, 1], [2, 3]])
    params = np.array([[0, 1], [2, 3]], dtype=np.float64)
    adam.run(None, 0, params, None, None)
    p1 = adam.update(params, grads)   # (sic) argument order of the reference test
    p2 = adam.update(np.array([[0, 0.999], [1.999, 2.999]]), grads)
    assert np.allclose(p1, np.array([[0, 0.999], [1.999, 2.999]]))
    assert np.allclose(p2, np.array([[0, 0.99900003], [1.99900001, 2.99900001]]))


def test_adam_golden_trajectory_and_sgd():
    g = golden("units")
    adam = Adam(learning_rate=0.05, learning_rate_decay=7.0, clip_grads=0.6, scale_grads=1.5)
    params = g["adam_traj"][0]
    adam.run(None, 0, params, None, None)
    for grads, expected in zip(g["adam_grads"], g["adam_traj"][1:]):
        params = adam.update(grads, params)
        assert np.array_equal(params, expected)
    assert np.allclose(SGD(learning_rate=1).update(np.ones(5), np.ones(5)), np.zeros(5))
    assert "adam, beta_1: 0.9" in str(Adam())


def test_clip_strip_slap_and_defaults():
    g = golden("units")
    cr, cc = g["clip_in_real"].copy(), g["clip_in_complex"].copy()
    common.clip_control_norms(cr, g["clip_norms"])
    common.clip_control_norms(cc, g["clip_norms"])
    assert np.array_equal(cr, g["clip_out_real"]) and np.array_equal(cc, g["clip_out_complex"])
    flat = common.strip_controls(True, g["clip_in_complex"])
    assert np.array_equal(flat, g["strip_complex"])
    assert np.array_equal(common.slap_controls(True, flat, (6, 2)), g["slap_complex"])
    # real controls: slap returns a view, so clipping aliases the optimizer's vector
    params = np.array([3.0, -4.0, 0.1, 0.2])
    view = common.slap_controls(False, params, (2, 2))
    common.clip_control_norms(view, np.array([1.0, 1.0]))
    assert np.array_equal(params, np.array([1.0, -1.0, 0.1, 0.2]))
    controls, norms = common.initialize_controls(True, 2, 5, 1.0, None, None)
    assert np.array_equal(norms, np.ones(2))
    assert np.allclose(controls, (0.1 - 0.1j) / np.sqrt(2))
    with pytest.raises(ValueError):
        common.initialize_controls(True, 1, 3, 1.0, np.ones((3, 1)), None)
    with pytest.raises(ValueError):
        common.initialize_controls(False, 1, 3, 1.0, np.ones((3, 1)) * 1j, None)
    with pytest.raises(ValueError):
        common.initialize_controls(False, 1, 3, 1.0, np.ones((3, 1)) * 2, None)
    cos = common.gen_controls_cos(False, 1, 40, 1.0, np.array([2.0]))
    assert np.isclose(cos[0, 0], 1.0) and cos.shape == (40, 1)


def test_constants():
    a, ad = get_annihilation_operator(4), get_creation_operator(4)
    assert np.allclose(a, ad.T) and np.isclose(a[2, 3], np.sqrt(3))
    assert np.allclose(qoc_amd.standard.get_eij(1, 2, 3)[1, 2], 1)


# ---- Hamiltonian structure ------------------------------------------------------------------------

def test_probe_hamiltonian():
    rng = np.random.default_rng(0)
    h0, a = cases_mod.gue(rng, 4), cases_mod.annihilation(4)
    ham = lambda u, t: h0 + u[0] * a + np.conjugate(u[0]) * a.conj().T
    times = [0.1, 0.2, 0.3]
    p0, g = structure.probe_hamiltonian(ham, 4, 1, True, times)
    assert p0.shape == (1, 4, 4) and g.shape == (1, 2, 4, 4)
    assert np.allclose(g[0, 0], a + a.conj().T) and np.allclose(g[0, 1], 1j * (a - a.conj().T))
    ham_t = lambda u, t: h0 * np.cos(t) + u[0] * (a + a.conj().T)
    p0, g = structure.probe_hamiltonian(ham_t, 4, 1, False, times)
    assert p0.shape == (3, 4, 4) and np.allclose(p0[1], h0 * np.cos(0.2))
    with pytest.raises(structure.NonLinearHamiltonianError):
        structure.probe_hamiltonian(lambda u, t: h0 + u[0] ** 2 * a, 4, 1, False, times)
    u = np.array([[1 + 2j, 3 - 1j]])
    r = structure.to_real_controls(u, True)
    assert np.array_equal(r, np.array([[1.0, 2.0, 3.0, -1.0]]))
    assert np.array_equal(structure.from_real_gradients(r, True), u)


# ---- entry points -----------------------------------------------------------------------------------

@pytest.mark.parametrize("name", [c.name for c in cases_mod.all_cases()])
def test_evolve_matches_reference_fixtures(name):
    case = cases_mod.case_by_name(name)
    g = golden(name)
    controls = [None] if case.controls is None else list(case.controls[:2])
    if name == "c3_subset":
        controls = controls[:1]
    for b, u in enumerate(controls):
        result = qoc_amd.evolve_schroedinger_discrete(
            case.T, case.hamiltonian(), case.initial_states, case.N, controls=u,
            cost_eval_step=case.cost_eval_step, costs=product_cost_list(case),
            magnus_policy=getattr(MagnusPolicy, case.magnus))
        assert abs(result.error - g["error"][b]) < 1e-11 * max(1, abs(g["error"][b]))
        assert result.final_states.shape == case.initial_states.shape
        assert rel_err(result.final_states, g["final_states"][b]) < 1e-11


def test_evolve_with_user_cost_and_errors():
    case = cases_mod.case_by_name("nc10_n101")

    class Population(Cost):
        name = "population"
        requires_step_evaluation = True

        def cost(self, controls, states, step):
            return float(np.abs(states[0, 0, 0]) ** 2) * 1e-3

    r = qoc_amd.evolve_schroedinger_discrete(case.T, case.hamiltonian(), case.initial_states,
                                             case.N, controls=case.controls[0],
                                             costs=[Population()], cost_eval_step=10)
    inter = []
    onp.evaluate(onp.SchroedingerProblem(case.T, case.hamiltonian(), case.initial_states, case.N,
                                         control_eval_count=case.Nc, control_count=case.K),
                 case.controls[0], intermediate=inter)
    expected = sum(abs(inter[s][0, 0, 0]) ** 2 * 1e-3 for s in range(10, case.N, 10))
    assert abs(r.error - expected) < 1e-12
    with pytest.raises(NotImplementedError):
        qoc_amd.evolve_schroedinger_discrete(case.T, case.hamiltonian(), case.initial_states,
                                             case.N, controls=case.controls[0],
                                             interpolation_policy="cubic")
    with pytest.raises(ValueError):
        qoc_amd.evolve_schroedinger_discrete(case.T, case.hamiltonian(), case.initial_states,
                                             case.N, controls=case.controls[0], magnus_policy=7)


def test_grape_clips_controls():
    # reference tests/test_core.py:563-602
    hm = 0.5 * (np.kron(SIGMA_X, SIGMA_X) + np.kron(SIGMA_Y, SIGMA_Y))
    hamiltonian = lambda controls, t: controls[0] * hm
    initial_states = np.array([[[0], [1], [0], [0]]])
    forbidden_states = np.array([[[[0], [1], [0], [0]]]])
    max_control_norms = np.repeat(1e-10, 1)
    result = qoc_amd.grape_schroedinger_discrete(
        1, 11, [ForbidStates(forbidden_states, 11)], 10, hamiltonian, initial_states, 11,
        iteration_count=15, log_iteration_step=0, max_control_norms=max_control_norms)
    assert np.less_equal(np.abs(result.best_controls[:, 0]), max_control_norms[0]).all()
    assert result.best_final_states.shape == (1, 4, 1)


def test_grape_optimizes_and_logs(capsys):
    case = cases_mod.case_by_name("ctrlcosts_r")
    costs = product_cost_list(case)
    result = qoc_amd.grape_schroedinger_discrete(
        case.K, case.Nc, costs, case.T, case.hamiltonian(), case.initial_states, case.N,
        initial_controls=case.controls[0], iteration_count=12, log_iteration_step=5,
        optimizer=Adam(learning_rate=5e-2), max_control_norms=np.array([5.0, 5.0]))
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "iter   |   total error  |    grads_l2   "
    assert out[1] == "=" * 41
    rows = [line.split("|")[0].strip() for line in out[2:]]
    assert rows == ["0", "5", "10", "11"]
    first_error = float(out[2].split("|")[1])
    assert result.best_error < first_error
    assert result.best_iteration > 0 and result.best_controls.shape == (case.Nc, case.K)
    # the first iteration's value/gradient are the golden ones
    g = golden(case.name)
    assert abs(first_error - g["error"][0]) < 1e-8


def test_grape_complex_controls_min_error_and_lbfgsb():
    case = cases_mod.case_small_complex("M2")
    costs = product_cost_list(case)
    seen = []

    def conditions(controls):
        seen.append(controls.copy())
        controls[0, :] = 0
        return controls

    result = qoc_amd.grape_schroedinger_discrete(
        case.K, case.Nc, costs, case.T, case.hamiltonian(), case.initial_states, case.N,
        complex_controls=True, cost_eval_step=case.cost_eval_step,
        initial_controls=case.controls[0], impose_control_conditions=conditions,
        iteration_count=4, log_iteration_step=0, optimizer=SGD(learning_rate=0.1),
        max_control_norms=np.array([3.0]))
    assert len(seen) == 4 and np.iscomplexobj(result.best_controls)
    assert np.all(result.best_controls[0] == 0)
    # min_error above the first error: terminate after one evaluation
    result = qoc_amd.grape_schroedinger_discrete(
        case.K, case.Nc, costs, case.T, case.hamiltonian(), case.initial_states, case.N,
        complex_controls=True, cost_eval_step=case.cost_eval_step,
        initial_controls=case.controls[0], iteration_count=50, log_iteration_step=0,
        min_error=10.0, max_control_norms=np.array([3.0]))
    assert result.best_iteration == 0
    real_case = cases_mod.case_by_name("ctrlcosts_r")
    result = qoc_amd.grape_schroedinger_discrete(
        real_case.K, real_case.Nc, product_cost_list(real_case), real_case.T,
        real_case.hamiltonian(), real_case.initial_states, real_case.N,
        initial_controls=real_case.controls[0], iteration_count=3, log_iteration_step=0,
        optimizer=LBFGSB(), max_control_norms=np.array([5.0, 5.0]))
    assert result.best_error < golden(real_case.name)["error"][0]


def test_batch_evaluator_matches_single():
    case = cases_mod.case_by_name("scaled_n8")
    ev = device.SchroedingerEvaluator(case.T, case.hamiltonian(), case.initial_states, case.N,
                                      control_count=case.K, control_eval_count=case.Nc,
                                      costs=product_cost_list(case))
    errors, grads, final, _ = ev.evaluate_batch(case.controls)
    g = golden(case.name)
    assert rel_err(errors, g["error"]) < 1e-11 and rel_err(grads, g["grads_ad"]) < 1e-9
    assert rel_err(final, g["final_states"]) < 1e-11


# ---- user Cost plugins in GRAPE: host-supplied state cotangents ----------------------------------

class _UserOccupation(Cost):
    """What ForbidStates computes for one forbidden basis state, written as a user plugin."""
    name = "user_occupation"
    requires_step_evaluation = True
    uses_controls = False

    def __init__(self, count, with_hook, cost_multiplier=1.):
        super().__init__(cost_multiplier)
        self.count = count
        self.with_hook = with_hook
        self.calls = 0

    def cost(self, controls, states, step):
        self.calls += 1
        return self.cost_multiplier / self.count * float(np.abs(states[0, 1, 0]) ** 2)

    def states_bar(self, controls, states, step):
        if not self.with_hook:
            return None
        out = np.zeros_like(states)
        out[0, 1, 0] = 2 * self.cost_multiplier / self.count * states[0, 1, 0]
        return out


class _UserFinalPlusControls(Cost):
    """A final-time cost that also depends on the controls explicitly (no hooks at all)."""
    name = "user_final"
    requires_step_evaluation = False

    def cost(self, controls, states, step):
        return float(np.real(states[0, 0, 0] * np.conj(states[0, 2, 0]))) + 0.01 * float(
            np.sum(np.abs(controls) ** 2))


@pytest.mark.parametrize("with_hook", [True, False])
def test_grape_with_user_state_cost_matches_builtin(with_hook):
    case = cases_mod.case_by_name("nc10_n101")
    ces = 10
    count = (case.N - 1) // ces
    forb = np.zeros((1, 1, case.n, 1), dtype=np.complex128)
    forb[0, 0, 1, 0] = 1
    target = product_cost_list(case)
    builtin = target + [ForbidStates(forb, case.N, cost_eval_step=ces, cost_multiplier=0.7)]
    user = _UserOccupation(count, with_hook, cost_multiplier=0.7)
    ev_ref = device.SchroedingerEvaluator(
        case.T, case.hamiltonian(), case.initial_states, case.N, control_count=case.K,
        control_eval_count=case.Nc, costs=builtin, cost_eval_step=ces)
    ev_user = device.SchroedingerEvaluator(
        case.T, case.hamiltonian(), case.initial_states, case.N, control_count=case.K,
        control_eval_count=case.Nc, costs=target + [user], cost_eval_step=ces)
    e0, g0, f0, _ = ev_ref.evaluate(case.controls[0])
    e1, g1, f1, _ = ev_user.evaluate(case.controls[0])
    assert abs(e0 - e1) < 1e-13 and rel_err(f1, f0) < 1e-13
    assert rel_err(g1, g0) < (1e-12 if with_hook else 1e-7)
    if not with_hook:
        assert user.calls > 4 * case.n * count  # finite differences of cost()
    # and the whole entry point runs with it
    result = qoc_amd.grape_schroedinger_discrete(
        case.K, case.Nc, target + [user], case.T, case.hamiltonian(), case.initial_states,
        case.N, cost_eval_step=ces, initial_controls=case.controls[0], iteration_count=3,
        log_iteration_step=0, max_control_norms=np.array([5.0, 5.0]))
    assert result.best_error <= e1 + 1e-12


def test_user_cost_with_explicit_control_dependence():
    case = cases_mod.case_by_name("ctrlcosts_r")
    cost = _UserFinalPlusControls()
    ev = device.SchroedingerEvaluator(
        case.T, case.hamiltonian(), case.initial_states, case.N, control_count=case.K,
        control_eval_count=case.Nc, costs=[cost])
    u = case.controls[0]
    err, grads, _, _ = ev.evaluate(u)
    rng = np.random.default_rng(2)
    d = rng.standard_normal(u.shape)
    h = 1e-5
    fd = (ev.evaluate(u + h * d, want_grad=False)[0] - ev.evaluate(u - h * d, want_grad=False)[0]) / (2 * h)
    assert abs(fd - np.sum(grads * d)) < 1e-7 * max(1.0, abs(fd))


def test_file_and_json_helpers(tmp_path):
    import json
    from qoc_amd.standard import CustomJSONEncoder, generate_save_file_path
    first = generate_save_file_path("run", str(tmp_path / "out"))
    assert first.endswith("out/00000_run.h5")
    open(first, "w").close()
    open(str(tmp_path / "out" / "00007_run.h5"), "w").close()
    open(str(tmp_path / "out" / "00003_other.h5"), "w").close()
    assert generate_save_file_path("run", str(tmp_path / "out")).endswith("00008_run.h5")
    assert generate_save_file_path("other", str(tmp_path / "out")).endswith("00004_other.h5")
    text = json.dumps({"a": np.arange(3), "b": np.float64(0.5), "c": np.int32(7)},
                      cls=CustomJSONEncoder)
    assert json.loads(text) == {"a": [0, 1, 2], "b": 0.5, "c": 7}


def test_expm_through_the_engine():
    # reference tests/test_standard.py:228-247 intends this comparison with scipy
    import scipy.linalg
    from qoc_amd.standard import expm
    rng = np.random.default_rng(9)
    for n, scale in ((2, 0.5), (5, 3.0), (17, 9.0)):
        a = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) * scale / n
        assert rel_err(expm(a), onp.expm_pade(a)) < 1e-12
        assert rel_err(expm(a), scipy.linalg.expm(a)) < 1e-10
    with pytest.raises(ValueError):
        expm(np.zeros((2, 3)))


# ---- multi-start GRAPE (grape_schroedinger_discrete_batch) ---------------------------------------

def _batch_problem(case, seeds, sigma=0.3):
    rng = np.random.default_rng(321)
    shape = (seeds, case.Nc, case.K)
    u = sigma * rng.standard_normal(shape)
    if case.complex_controls:
        u = u + 1j * sigma * rng.standard_normal(shape)
    return u


@pytest.mark.parametrize("name", ["ctrlcosts_r", "small_complex_M2"])
def test_batch_grape_equals_independent_single_runs(name, capsys):
    """Every seed of the batched driver walks the trajectory of grape_schroedinger_discrete
    started from the same controls: same best error / iteration / controls (the oracle backend
    evaluates a batch seed by seed, so the agreement is exact)."""
    case = cases_mod.case_by_name(name)
    u0 = _batch_problem(case, 3)
    norms = np.full(case.K, 2.0)
    kw = dict(complex_controls=case.complex_controls, cost_eval_step=case.cost_eval_step,
              iteration_count=5, max_control_norms=norms,
              magnus_policy=getattr(MagnusPolicy, case.magnus))
    batch = qoc_amd.grape_schroedinger_discrete_batch(
        case.K, case.Nc, product_cost_list(case), case.T, case.hamiltonian(),
        case.initial_states, case.N, u0.copy(), optimizer=Adam(learning_rate=3e-2),
        log_iteration_step=2, **kw)
    out = capsys.readouterr().out.splitlines()
    assert out[0].startswith("iter   |  summed error")
    assert [line.split("|")[0].strip() for line in out[2:]] == ["0", "2", "4"]
    for b in range(3):
        single = qoc_amd.grape_schroedinger_discrete(
            case.K, case.Nc, product_cost_list(case), case.T, case.hamiltonian(),
            case.initial_states, case.N, initial_controls=u0[b].copy(),
            optimizer=Adam(learning_rate=3e-2), log_iteration_step=0, **kw)
        assert batch.best_error[b] == single.best_error
        assert batch.best_iteration[b] == single.best_iteration
        assert np.array_equal(batch.best_controls[b], single.best_controls)
        assert np.array_equal(batch.best_final_states[b], single.best_final_states)
    assert batch.best.best_error == np.min(batch.best_error)
    assert batch.global_best_error == batch.best.best_error
    assert np.all(batch.iterations_run == 5)


def test_batch_grape_per_seed_termination_conditions_and_errors():
    case = cases_mod.case_by_name("ctrlcosts_r")
    u0 = _batch_problem(case, 4, sigma=0.2)

    def conditions(controls):
        controls[0, :] = 0
        return controls

    base = dict(cost_eval_step=case.cost_eval_step, max_control_norms=np.full(case.K, 1.0),
                log_iteration_step=0, impose_control_conditions=conditions)
    args = (case.K, case.Nc, product_cost_list(case), case.T, case.hamiltonian(),
            case.initial_states, case.N)
    ref = qoc_amd.grape_schroedinger_discrete_batch(*args, u0.copy(), iteration_count=3,
                                                    optimizer=SGD(learning_rate=0.5), **base)
    # a seed whose first error is below min_error stops at once, the others carry on untouched
    threshold = float(np.sort(ref.best_error)[1]) + 1.0  # generous: at least two seeds stop early
    early = qoc_amd.grape_schroedinger_discrete_batch(*args, u0.copy(), iteration_count=3,
                                                      optimizer=SGD(learning_rate=0.5),
                                                      min_error=threshold, **base)
    assert np.all(early.iterations_run >= 1) and np.any(early.iterations_run == 1)
    for b in range(4):
        assert np.all(early.best_controls[b][0] == 0)
        assert np.max(np.abs(early.best_controls[b])) <= 1.0 + 1e-12
    with pytest.raises(ValueError):
        qoc_amd.grape_schroedinger_discrete_batch(*args, u0[0], **base)
    with pytest.raises(NotImplementedError):
        qoc_amd.grape_schroedinger_discrete_batch(*args, u0.copy(), optimizer=LBFGSB(), **base)
    with pytest.raises(ValueError):  # initial controls beyond max_control_norms
        qoc_amd.grape_schroedinger_discrete_batch(*args, 10 * u0, **base)


def test_batched_optimizer_states_equal_per_seed_plugins():
    """the built-in Adam / SGD run on [B, P] arrays (clip, update: host threads
    of libqocx in the reference's operation order); any other plugin keeps one deep copy per seed.
    Both routes give the same trajectories bit for bit - here with per-seed termination, gradient
    clipping and a learning-rate decay in play."""
    case = cases_mod.case_by_name("small_complex_M2")
    case_r = cases_mod.case_by_name("ctrlcosts_r")

    class PluginAdam(Adam):  # not type(...) is Adam: takes the per-seed route
        pass

    for c, kwargs in ((case_r, dict(learning_rate=0.3, clip_grads=0.05)),
                      (case, dict(learning_rate=0.2, learning_rate_decay=3.0))):
        u0 = _batch_problem(c, 5, sigma=0.3)
        args = (c.K, c.Nc, product_cost_list(c), c.T, c.hamiltonian(), c.initial_states, c.N)
        base = dict(complex_controls=c.complex_controls, cost_eval_step=c.cost_eval_step,
                    max_control_norms=np.full(c.K, 0.6), log_iteration_step=0, iteration_count=4,
                    magnus_policy=getattr(MagnusPolicy, c.magnus))
        first = qoc_amd.grape_schroedinger_discrete_batch(*args, np.clip(u0.real, -0.4, 0.4)
                                                          + 1j * np.clip(u0.imag, -0.4, 0.4)
                                                          if c.complex_controls else np.clip(u0, -0.6, 0.6),
                                                          optimizer=Adam(**kwargs),
                                                          **dict(base, iteration_count=1))
        threshold = float(np.sort(first.best_error)[2])  # three seeds stop after one evaluation
        runs = []
        for opt in (Adam(**kwargs), PluginAdam(**kwargs)):
            start = (np.clip(u0.real, -0.4, 0.4) + 1j * np.clip(u0.imag, -0.4, 0.4)
                     if c.complex_controls else np.clip(u0, -0.6, 0.6))
            runs.append(qoc_amd.grape_schroedinger_discrete_batch(
                *args, start, optimizer=opt, min_error=threshold, **base))
        a, b = runs
        assert np.array_equal(a.best_error, b.best_error)
        assert np.array_equal(a.iterations_run, b.iterations_run)
        assert len(set(a.iterations_run.tolist())) >= 2  # seeds stop at different iterations
        for s in range(5):
            assert np.array_equal(a.best_controls[s], b.best_controls[s])
            assert np.array_equal(a.best_final_states[s], b.best_final_states[s])


def test_batch_driver_host_time_per_iteration(monkeypatch):
    """Host work of one multi-start iteration at the headline batch (256 seeds, 1001 x 2 real
    controls, Adam): clip, bookkeeping and 256 optimizer updates. The per-seed Python loop took
    13 ms; the [B, P] form is bound by the host's memory bandwidth (seven
    4 MB arrays per update): ~5 ms on the 8-core build container. The evaluator is a stub."""
    import time
    from qoc_amd.core import schroedingerdiscrete as sd
    B, Nc, K, n = 256, 1001, 2, 4
    rng = np.random.default_rng(5)
    grads = rng.standard_normal((B, Nc, K))
    finals = np.zeros((B, 1, n, 1), dtype=np.complex128)
    spent = []

    class StubEvaluator(object):
        def __init__(self, *a, **k):
            self.calls = 0

        def evaluate_batch(self, controls, want_grad=True):
            if self.calls:
                spent.append(time.perf_counter() - self.stamp)
            self.calls += 1
            errors = 1.0 + 0.001 * rng.standard_normal(B) - 0.01 * self.calls
            self.stamp = time.perf_counter()
            return errors, grads, finals, None
    monkeypatch.setattr(sd, "SchroedingerEvaluator", StubEvaluator)
    u0 = 0.1 * rng.standard_normal((B, Nc, K))
    result = qoc_amd.grape_schroedinger_discrete_batch(
        K, Nc, [], 1.0, lambda u, t: np.eye(n), np.zeros((1, n, 1)), 11, u0, iteration_count=8,
        log_iteration_step=0, optimizer=Adam(learning_rate=1e-2))
    assert np.all(result.iterations_run == 8)
    per_iteration = float(np.median(spent))
    print("host time per multi-start iteration: {:.2f} ms".format(per_iteration * 1e3))
    assert per_iteration < 9e-3


# ---- opaque (non-linear) Hamiltonians: host-side sampling and gradient assembly -----------------

def test_opaque_interpolation_matches_reference_rule():
    g = golden("units")  # minted from the reference's interpolate_linear_set
    xs, ys, xq = g["interp_xs"], g["interp_ys"], g["interp_xq"]
    rows = structure.interpolation_rows(float(xs[-1]), len(xs), xq)
    assert np.array_equal(structure.controls_at(ys, rows, xq), g["interp_out"])


def test_opaque_generator_gradients_against_finite_differences():
    """generator_gradients = the chain rule through M_j(u) = -i dt H(u(t_j), t_j): checked on a
    cost that is an arbitrary smooth function of the generators, whose cotangents are known."""
    case = cases_mod.case_by_name("opaque_stark_complex")
    h = case.hamiltonian()
    dt = case.T / (case.N - 1)
    times = [j * dt + 0.5 * dt for j in range(case.N - 1)]
    rows = structure.interpolation_rows(case.T, case.Nc, times)
    rng = np.random.default_rng(8)
    weights = rng.standard_normal((case.N - 1, case.n, case.n)) \
        + 1j * rng.standard_normal((case.N - 1, case.n, case.n))

    def cost(u):  # sum_j |<W_j, M_j>|^2 : cotangent 2 <W_j, M_j> W_j
        gens, _ = structure.sample_generators(h, u, rows, times, dt, case.n)
        ip = np.sum(np.conj(weights) * gens, axis=(1, 2))
        return float(np.sum(np.abs(ip) ** 2)), 2 * ip[:, None, None] * weights
    u0 = case.controls[0]
    _, bars = cost(u0)
    grads = structure.generator_gradients(h, u0, rows, times, dt, bars, True)
    for index in [(0, 0), (5, 0), (case.Nc - 1, 0)]:
        for direction in (1.0, 1.0j):
            step = 1e-5
            up, dn = u0.copy(), u0.copy()
            up[index] += step * direction
            dn[index] -= step * direction
            fd = (cost(up)[0] - cost(dn)[0]) / (2 * step)
            ref = grads[index].real if direction == 1.0 else grads[index].imag
            assert abs(fd - ref) < 1e-7 * max(1.0, abs(ref))


def check_linearized_opaque_case(name):
    """A hamiltonian that is not linear in the controls under MagnusPolicy.M4 / M6 (the reference
    takes any callable under any policy, schroedingerdiscrete.py:483-497): the host hands the
    engine the tangent of the callable at the current controls. Against fixtures minted from the
    reference: forward 1e-10, gradient 1e-8 vs AD of the same op sequence and 1e-7 vs finite
    differences of the reference forward; and GRAPE runs on it. (CPU: oracle backend; GPU: engine.)"""
    case = cases_mod.case_by_name(name)
    g = golden(name)
    policy = getattr(MagnusPolicy, case.magnus)
    args = dict(cost_eval_step=case.cost_eval_step, costs=product_cost_list(case),
                magnus_policy=policy)
    for b, u in enumerate(case.controls):
        result = qoc_amd.evolve_schroedinger_discrete(
            case.T, case.hamiltonian(), case.initial_states, case.N, controls=u, **args)
        assert abs(result.error - g["error"][b]) < 1e-10
        assert rel_err(result.final_states, g["final_states"][b]) < 1e-10
    ev = device.SchroedingerEvaluator(
        case.T, case.hamiltonian(), case.initial_states, case.N, control_count=case.K,
        control_eval_count=case.Nc, complex_controls=case.complex_controls,
        costs=product_cost_list(case), cost_eval_step=case.cost_eval_step, magnus_policy=policy)
    assert ev.linearized_hamiltonian is not None and ev.opaque_hamiltonian is None
    errors, grads, finals, _ = ev.evaluate_batch(np.stack(case.controls), want_grad=True)
    for b in range(len(case.controls)):
        assert abs(errors[b] - g["error"][b]) < 1e-10
        assert rel_err(grads[b], g["grads_ad"][b]) < 1e-8
        scale = np.max(np.abs(g["grads_ad"][b]))
        assert np.max(np.abs(np.asarray(grads[b]).flat[g["fd_index"][b]] - g["grads_fd"][b])) / scale < 1e-7
    result = qoc_amd.grape_schroedinger_discrete(
        case.K, case.Nc, product_cost_list(case), case.T, case.hamiltonian(), case.initial_states,
        case.N, complex_controls=case.complex_controls, initial_controls=case.controls[0].copy(),
        iteration_count=6, log_iteration_step=0, optimizer=Adam(learning_rate=3e-2),
        max_control_norms=np.full(case.K, 3.0), magnus_policy=policy)
    assert result.best_error < g["error"][0] and result.best_iteration > 0


@pytest.mark.parametrize("name", ["opaque_eps2_M4", "opaque_stark_M6"])
def test_opaque_hamiltonian_under_higher_magnus_policies(name):
    check_linearized_opaque_case(name)


def test_product_has_no_backend_hook_and_no_cpu_fallback():
    """The product carries no test hook (the tests patch qoc_amd.core.device.make_backend from
    outside, tests/helpers.py) and no CPU fallback: without a GPU the engine refuses to start."""
    from qoc_amd import engine
    assert not hasattr(device, "set_backend_factory")
    helpers.set_backend_factory(None)
    try:
        lib = engine.load_library()
        import ctypes
        count = ctypes.c_int(0)
        has_gpu = lib.qocx_device_count(ctypes.byref(count)) == 0 and count.value > 0
        if has_gpu:
            pytest.skip("a GPU is visible: the engine starts")
        with pytest.raises(Exception):
            device.make_backend()
    finally:
        helpers.set_backend_factory(OracleBackend)
