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
