"""
lindbladdiscrete.py - evolve_lindblad_discrete and grape_lindblad_discrete.

Same positional/keyword signatures, result objects, side effects (stdout table, save file) and
optimizer-callback protocol as qoc/core/lindbladdiscrete.py:31-257. The integration of the
master equation and the gradient (the reference's :357-441 and the adaptive RKDP5 of
mathmethods.py:352-480, traced by autograd) run on the MI355X through
qoc_amd.core.device.LindbladEvaluator (fixed-step DOP853 + discrete adjoint; DESIGN.md 9 for
the parity tolerances that follow from replacing an adaptive integrator).
"""

import numpy as np

from qoc_amd.core.common import (clip_control_norms, initialize_controls, slap_controls,
                                 strip_controls)
from qoc_amd.core.device import LindbladEvaluator
from qoc_amd.core.structure import NonLinearHamiltonianError
from qoc_amd.models import (Dummy, EvolveLindbladDiscreteState, EvolveLindbladResult,
                            GrapeLindbladDiscreteState, GrapeLindbladResult,
                            InterpolationPolicy)
from qoc_amd.standard.optimizers import Adam


def evolve_lindblad_discrete(evolution_time, initial_densities, system_eval_count,
                             controls=None, cost_eval_step=1, costs=list(), hamiltonian=None,
                             interpolation_policy=InterpolationPolicy.LINEAR, lindblad_data=None,
                             save_file_path=None, save_intermediate_densities=False):
    """
    Evolve density matrices under the Lindblad master equation and compute the optimization
    error. Arguments as in the reference (lindbladdiscrete.py:31-91):
    initial_densities :: (density_count x n x n); hamiltonian :: (controls, time) -> (n x n);
    lindblad_data :: (time) -> (dissipators (L), operators (L x n x n)).
    Returns EvolveLindbladResult{error, final_densities}.
    """
    if controls is not None:
        controls = np.asarray(controls)
        control_eval_count, control_count = controls.shape[0], controls.shape[1]
    else:
        control_eval_count, control_count = 0, 0
    pstate = EvolveLindbladDiscreteState(control_eval_count, cost_eval_step, costs,
                                         evolution_time, hamiltonian, initial_densities,
                                         interpolation_policy, lindblad_data, save_file_path,
                                         save_intermediate_densities, system_eval_count)
    pstate.save_initial(controls)
    common = dict(hamiltonian=hamiltonian, lindblad_data=lindblad_data, costs=costs,
                  cost_eval_step=cost_eval_step, interpolation_policy=interpolation_policy,
                  need_gradients=False)
    try:
        evaluator = LindbladEvaluator(
            evolution_time, initial_densities, system_eval_count, control_count=control_count,
            control_eval_count=control_eval_count,
            complex_controls=controls is not None and np.iscomplexobj(controls), **common)
        device_controls = controls
    except NonLinearHamiltonianError:
        # not linear in the controls: fold this control array into a time-dependent Hamiltonian
        evaluator = LindbladEvaluator(evolution_time, initial_densities, system_eval_count,
                                      frozen_controls=controls, **common)
        device_controls = None
    error, _, final_densities, step_densities = evaluator.evaluate(
        device_controls, want_grad=False,
        want_step_densities=pstate.save_intermediate_densities_)
    if pstate.save_intermediate_densities_:
        pstate.save_all_intermediate_densities(0, step_densities)
    return EvolveLindbladResult(error=error, final_densities=final_densities)


def grape_lindblad_discrete(control_count, control_eval_count, costs, evolution_time,
                            initial_densities, system_eval_count, complex_controls=False,
                            cost_eval_step=1, hamiltonian=None, impose_control_conditions=None,
                            initial_controls=None,
                            interpolation_policy=InterpolationPolicy.LINEAR,
                            iteration_count=1000, lindblad_data=None, log_iteration_step=10,
                            max_control_norms=None, min_error=0, optimizer=Adam(),
                            save_file_path=None, save_intermediate_densities=False,
                            save_iteration_step=0):
    """
    Optimize time-discrete controls for the evolution of a set of densities under the Lindblad
    equation (GRAPE). Arguments as in the reference (lindbladdiscrete.py:106-212).
    Returns GrapeLindbladResult{best_controls, best_error, best_final_densities,
    best_iteration}.
    """
    initial_controls, max_control_norms = initialize_controls(
        complex_controls, control_count, control_eval_count, evolution_time, initial_controls,
        max_control_norms)
    pstate = GrapeLindbladDiscreteState(
        complex_controls, control_count, control_eval_count, cost_eval_step, costs,
        evolution_time, hamiltonian, impose_control_conditions, initial_controls,
        initial_densities, interpolation_policy, iteration_count, lindblad_data,
        log_iteration_step, max_control_norms, min_error, optimizer, save_file_path,
        save_intermediate_densities, save_iteration_step, system_eval_count)
    pstate.evaluator = LindbladEvaluator(
        evolution_time, initial_densities, system_eval_count, hamiltonian=hamiltonian,
        lindblad_data=lindblad_data, control_count=control_count,
        control_eval_count=control_eval_count, complex_controls=complex_controls, costs=costs,
        cost_eval_step=cost_eval_step, interpolation_policy=interpolation_policy,
        need_gradients=True, control_bounds=max_control_norms)
    pstate.log_and_save_initial()
    reporter = Dummy()
    reporter.iteration = 0
    result = GrapeLindbladResult()
    flat_controls = strip_controls(pstate.complex_controls, pstate.initial_controls)
    pstate.optimizer.run(_eld_wrap, pstate.iteration_count, flat_controls, _eldj_wrap,
                         args=(pstate, reporter, result))
    return result


def _cost_format(flat_controls, pstate):
    """optimizer format -> clipped, conditioned cost-function format (:272-280)."""
    controls = slap_controls(pstate.complex_controls, flat_controls, pstate.controls_shape)
    clip_control_norms(controls, pstate.max_control_norms)
    if pstate.impose_control_conditions is not None:
        controls = pstate.impose_control_conditions(controls)
    return controls


def _eld_wrap(controls, pstate, reporter, result):
    controls = _cost_format(controls, pstate)
    error, _, final_densities, _ = pstate.evaluator.evaluate(controls, want_grad=False)
    reporter.error = error
    reporter.final_densities = final_densities
    return error, bool(error <= pstate.min_error)


def _eldj_wrap(controls, pstate, reporter, result):
    controls = _cost_format(controls, pstate)
    save_densities = pstate.save_intermediate_densities_
    error, grads, final_densities, step_densities = pstate.evaluator.evaluate(
        controls, want_grad=True, want_step_densities=save_densities)
    reporter.error = error
    reporter.final_densities = final_densities
    if save_densities:
        pstate.save_all_intermediate_densities(reporter.iteration, step_densities)
    if error < result.best_error:  # strict, as lindbladdiscrete.py:334
        result.best_controls = controls
        result.best_error = error
        result.best_final_densities = final_densities
        result.best_iteration = reporter.iteration
    pstate.log_and_save(controls, error, final_densities, grads, reporter.iteration)
    reporter.iteration += 1
    return strip_controls(pstate.complex_controls, grads), bool(error <= pstate.min_error)
