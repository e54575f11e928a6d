"""
schroedingerdiscrete.py - evolve_schroedinger_discrete and grape_schroedinger_discrete.

Same positional/keyword signatures, result objects, side effects (stdout table, save file) and
optimizer-callback protocol as qoc/core/schroedingerdiscrete.py:28-353. The time loop, the
matrix exponential and the gradient (the reference's :356-502 traced by autograd) run on the
MI355X through qoc_amd.core.device.SchroedingerEvaluator.
"""

import numpy as np

from qoc_amd.core.common import (clip_control_norms, initialize_controls, slap_controls,
                                 strip_controls)
from qoc_amd.core.device import SchroedingerEvaluator
from qoc_amd.models import (Dummy, EvolveSchroedingerDiscreteState, EvolveSchroedingerResult,
                            GrapeSchroedingerDiscreteState, GrapeSchroedingerResult,
                            InterpolationPolicy, MagnusPolicy)
from qoc_amd.standard.optimizers import Adam


def evolve_schroedinger_discrete(evolution_time, hamiltonian, initial_states, system_eval_count,
                                 controls=None, cost_eval_step=1, costs=list(),
                                 interpolation_policy=InterpolationPolicy.LINEAR,
                                 magnus_policy=MagnusPolicy.M2, save_file_path=None,
                                 save_intermediate_states=False):
    """
    Evolve state vectors under the Schroedinger equation and compute the optimization error.

    Arguments as in the reference (schroedingerdiscrete.py:28-84):
    evolution_time :: float; hamiltonian :: (controls (control_count), time) -> (n x n);
    initial_states :: (state_count x n x 1); system_eval_count :: int >= 2;
    controls :: (control_eval_count x control_count) or None; costs :: iterable(Cost).
    Returns EvolveSchroedingerResult{error, final_states}.
    """
    if controls is not None:
        controls = np.asarray(controls)
        control_eval_count, control_count = controls.shape[0], controls.shape[1]
    else:
        control_eval_count, control_count = 0, 0
    pstate = EvolveSchroedingerDiscreteState(control_eval_count, cost_eval_step, costs,
                                             evolution_time, hamiltonian, initial_states,
                                             interpolation_policy, magnus_policy, save_file_path,
                                             save_intermediate_states, system_eval_count)
    pstate.save_initial(controls)
    evaluator = SchroedingerEvaluator(
        evolution_time, hamiltonian, initial_states, system_eval_count,
        control_count=control_count, control_eval_count=control_eval_count,
        complex_controls=controls is not None and np.iscomplexobj(controls), costs=costs,
        cost_eval_step=cost_eval_step, interpolation_policy=interpolation_policy,
        magnus_policy=magnus_policy, need_gradients=False, latency_mode=True)
    error, _, final_states, step_states = evaluator.evaluate(
        controls, want_grad=False, want_step_states=pstate.save_intermediate_states_)
    if pstate.save_intermediate_states_:
        pstate.save_all_intermediate_states(0, step_states)
    return EvolveSchroedingerResult(error=error, final_states=final_states)


def grape_schroedinger_discrete(control_count, control_eval_count, costs, evolution_time,
                                hamiltonian, initial_states, system_eval_count,
                                complex_controls=False, cost_eval_step=1,
                                impose_control_conditions=None, initial_controls=None,
                                interpolation_policy=InterpolationPolicy.LINEAR,
                                iteration_count=1000, log_iteration_step=10,
                                magnus_policy=MagnusPolicy.M2, max_control_norms=None,
                                min_error=0, optimizer=Adam(), save_file_path=None,
                                save_intermediate_states=False, save_iteration_step=0):
    """
    Optimize time-discrete controls for the evolution of a set of states (GRAPE).
    Arguments as in the reference (schroedingerdiscrete.py:106-212).
    Returns GrapeSchroedingerResult{best_controls, best_error, best_final_states, best_iteration}.
    """
    initial_controls, max_control_norms = initialize_controls(
        complex_controls, control_count, control_eval_count, evolution_time, initial_controls,
        max_control_norms)
    pstate = GrapeSchroedingerDiscreteState(
        complex_controls, control_count, control_eval_count, cost_eval_step, costs,
        evolution_time, hamiltonian, impose_control_conditions, initial_controls, initial_states,
        interpolation_policy, iteration_count, log_iteration_step, max_control_norms,
        magnus_policy, min_error, optimizer, save_file_path, save_intermediate_states,
        save_iteration_step, system_eval_count)
    pstate.evaluator = SchroedingerEvaluator(
        evolution_time, hamiltonian, initial_states, system_eval_count,
        control_count=control_count, control_eval_count=control_eval_count,
        complex_controls=complex_controls, costs=costs, cost_eval_step=cost_eval_step,
        interpolation_policy=interpolation_policy, magnus_policy=magnus_policy,
        need_gradients=True, latency_mode=True)
    pstate.log_and_save_initial()
    reporter = Dummy()
    reporter.iteration = 0
    result = GrapeSchroedingerResult()
    flat_controls = strip_controls(pstate.complex_controls, pstate.initial_controls)
    pstate.optimizer.run(_esd_wrap, pstate.iteration_count, flat_controls, _esdj_wrap,
                         args=(pstate, reporter, result))
    return result


def _cost_format(flat_controls, pstate):
    """optimizer format -> clipped, conditioned cost-function format (:308-315)."""
    controls = slap_controls(pstate.complex_controls, flat_controls, pstate.controls_shape)
    clip_control_norms(controls, pstate.max_control_norms)  # in place, aliases real params
    if pstate.impose_control_conditions is not None:
        controls = pstate.impose_control_conditions(controls)
    return controls


def _esd_wrap(controls, pstate, reporter, result):
    """function(params, *args) -> (error, terminate); used by optimizers that ask for values."""
    controls = _cost_format(controls, pstate)
    error, _, final_states, _ = pstate.evaluator.evaluate(controls, want_grad=False)
    reporter.error = error
    reporter.final_states = final_states
    return error, bool(error <= pstate.min_error)


def _esdj_wrap(controls, pstate, reporter, result):
    """jacobian(params, *args) -> (grads, terminate); one device evaluation per call."""
    controls = _cost_format(controls, pstate)
    save_states = pstate.save_intermediate_states_
    error, grads, final_states, step_states = pstate.evaluator.evaluate(
        controls, want_grad=True, want_step_states=save_states)
    reporter.error = error
    reporter.final_states = final_states
    if save_states:
        pstate.save_all_intermediate_states(reporter.iteration, step_states)
    if error < result.best_error:  # strict, as schroedingerdiscrete.py:333
        result.best_controls = controls
        result.best_error = error
        result.best_final_states = final_states
        result.best_iteration = reporter.iteration
    pstate.log_and_save(controls, error, final_states, grads, reporter.iteration)
    reporter.iteration += 1
    return strip_controls(pstate.complex_controls, grads), bool(error <= pstate.min_error)


# ---- multi-start GRAPE: B independent optimisations in lock step ---------------

class GrapeSchroedingerBatchResult(object):
    """Per-seed bests of grape_schroedinger_discrete_batch; `best` is the overall winner as a
    GrapeSchroedingerResult. With a communicator the arrays hold this rank's seeds and
    `global_best_error` the minimum over all ranks."""

    def __init__(self, seed_count):
        self.best_controls = [None] * seed_count
        self.best_error = np.repeat(np.finfo(np.float64).max, seed_count)
        self.best_final_states = [None] * seed_count
        self.best_iteration = np.full(seed_count, -1, dtype=np.int64)
        self.iterations_run = np.zeros(seed_count, dtype=np.int64)
        self.global_best_error = None

    @property
    def best(self):
        b = int(np.argmin(self.best_error))
        return GrapeSchroedingerResult(
            best_controls=self.best_controls[b], best_error=float(self.best_error[b]),
            best_final_states=self.best_final_states[b], best_iteration=int(self.best_iteration[b]))


def _optimizer_clone(optimizer, flat_controls):
    """A private copy of a step-wise optimizer plugin, initialised as its run() would."""
    import copy
    if not hasattr(optimizer, "update"):
        raise NotImplementedError(
            "grape_schroedinger_discrete_batch drives the optimizer step by step and needs its "
            "update(grads, params) (Adam, SGD); {} only offers run().".format(optimizer))
    clone = copy.deepcopy(optimizer)
    if hasattr(clone, "gradient_moment"):  # Adam.run(), adam.py:83-88 of the reference
        clone.iteration_count = 0
        clone.gradient_moment = np.zeros_like(flat_controls)
        clone.gradient_square_moment = np.zeros_like(flat_controls)
    return clone


def grape_schroedinger_discrete_batch(control_count, control_eval_count, costs, evolution_time,
                                      hamiltonian, initial_states, system_eval_count,
                                      initial_controls, complex_controls=False, cost_eval_step=1,
                                      impose_control_conditions=None,
                                      interpolation_policy=InterpolationPolicy.LINEAR,
                                      iteration_count=1000, log_iteration_step=10,
                                      magnus_policy=MagnusPolicy.M2, max_control_norms=None,
                                      min_error=0, optimizer=Adam(), comm=None):
    """
    Multi-start GRAPE: B = len(initial_controls) independent optimisations of the same problem,
    one batched device evaluation per iteration (the engine's batch axis; the reference runs one
    control set per process). Seed b follows EXACTLY the iteration of
    grape_schroedinger_discrete (reference :293-353 per seed): clip -> conditions -> evaluate ->
    best-so-far (strict <) -> optimizer update, its own optimizer state (a deep copy of
    `optimizer`), its own termination at error <= min_error (a finished seed is frozen and no
    longer updated; the batch ends when every seed has finished or after iteration_count
    iterations).

    initial_controls :: (B x control_eval_count x control_count), each conforming to
    max_control_norms. comm (qoc_amd.parallel communicator, optional): the seed axis is sharded
    over its ranks, every rank optimises its own block, and the logged error is the all-reduced
    sum (the path's single collective); result arrays are rank local.
    Returns GrapeSchroedingerBatchResult.
    """
    from qoc_amd import parallel
    initial_controls = np.asarray(initial_controls)
    if initial_controls.ndim != 3:
        raise ValueError("initial_controls must be (seed_count x control_eval_count x "
                         "control_count), got shape {}".format(initial_controls.shape))
    comm = comm if comm is not None else parallel.SingleComm()
    lo, hi = parallel.shard_bounds(initial_controls.shape[0], comm.rank, comm.world)
    seeds = []
    for b in range(lo, hi):
        controls_b, max_control_norms = initialize_controls(
            complex_controls, control_count, control_eval_count, evolution_time,
            initial_controls[b], max_control_norms)
        seeds.append(np.array(controls_b))
    B = len(seeds)
    shape = (control_eval_count, control_count)
    pstate = Dummy()
    pstate.complex_controls = complex_controls
    pstate.controls_shape = shape
    pstate.max_control_norms = max_control_norms
    pstate.impose_control_conditions = impose_control_conditions
    evaluator = SchroedingerEvaluator(
        evolution_time, hamiltonian, initial_states, system_eval_count,
        control_count=control_count, control_eval_count=control_eval_count,
        complex_controls=complex_controls, costs=costs, cost_eval_step=cost_eval_step,
        interpolation_policy=interpolation_policy, magnus_policy=magnus_policy,
        need_gradients=True, latency_mode=B <= 128)
    params = [strip_controls(complex_controls, c) for c in seeds]
    optimizers = [_optimizer_clone(optimizer, p) for p in params]
    active = np.ones(B, dtype=bool)
    result = GrapeSchroedingerBatchResult(B)
    should_log = log_iteration_step != 0
    if should_log and comm.rank == 0:
        print("iter   |  summed error  |   min error    |  active seeds \n"
              "===========================================================")
    for iteration in range(iteration_count):
        # cost-function format of every seed (clipping acts in place on the optimizer's params
        # for real controls, exactly as in the single-seed driver)
        controls = [_cost_format(params[b], pstate) for b in range(B)]
        if B > 0:
            errors, grads, finals, _ = evaluator.evaluate_batch(np.stack(controls), want_grad=True)
        else:
            errors, grads, finals = np.zeros(0), np.zeros((0,) + shape), np.zeros(0)
        for b in range(B):
            if not active[b]:
                continue
            result.iterations_run[b] = iteration + 1
            if errors[b] < result.best_error[b]:
                result.best_controls[b] = controls[b]
                result.best_error[b] = errors[b]
                result.best_final_states[b] = finals[b]
                result.best_iteration[b] = iteration
        if should_log and (iteration % log_iteration_step == 0 or iteration == iteration_count - 1):
            local = np.array([float(np.sum(errors[active])) if B else 0.0])
            total = comm.allreduce_sum(local)[0]
            low = -comm.allreduce_max(np.array([-float(np.min(errors)) if B else -np.inf]))[0]
            count = comm.allreduce_sum(np.array([float(np.sum(active))]))[0]
            if comm.rank == 0:
                print("{:^6d} | {:^1.8e} | {:^1.8e} | {:^6d}".format(iteration, total, low,
                                                                     int(count)))
        for b in range(B):
            if not active[b]:
                continue
            if errors[b] <= min_error:  # the optimizer loop of this seed ends (terminate = True)
                active[b] = False
                continue
            params[b] = optimizers[b].update(strip_controls(complex_controls, grads[b]), params[b])
        still = comm.allreduce_sum(np.array([float(np.sum(active))]))[0]
        if still == 0:
            break
    best_local = float(np.min(result.best_error)) if B else np.inf
    result.global_best_error = float(-comm.allreduce_max(np.array([-best_local]))[0])
    return result
