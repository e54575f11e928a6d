"""
schroedingermodels.py - program states and results of the Schroedinger entry points.

Behaviour follows qoc/models/schroedingermodels.py: result field names (:113-131, :347-370),
the stdout table (:232-238, :315-317) and the HDF5 layout (:66-95, :274-308, :240-250).
h5py is imported lazily: it is only needed when a save_file_path is given.
"""

import numpy as np

from qoc_amd.models.policies import ProgramType
from qoc_amd.models.programstate import GrapeState, ProgramState


def _h5():
    try:
        import h5py
    except ImportError as exc:  # pragma: no cover - depends on the environment
        raise ImportError("saving to {} needs h5py, which is not installed".format("HDF5")) from exc
    from filelock import FileLock, Timeout
    return h5py, FileLock, Timeout


class EvolveSchroedingerDiscreteState(ProgramState):
    method = "evolve_schroedinger_discrete"

    def __init__(self, control_eval_count, cost_eval_step, costs, evolution_time, hamiltonian,
                 initial_states, interpolation_policy, magnus_policy, save_file_path,
                 save_intermediate_states_, system_eval_count):
        super().__init__(control_eval_count, cost_eval_step, costs, evolution_time, hamiltonian,
                         interpolation_policy, ProgramType.EVOLVE, save_file_path,
                         system_eval_count)
        self.initial_states = initial_states
        self.magnus_policy = magnus_policy
        self.save_intermediate_states_ = save_file_path is not None and save_intermediate_states_

    def save_initial(self, controls):
        if self.save_file_path is None:
            return
        h5py, FileLock, Timeout = _h5()
        print("QOC is saving this evolution to {}.".format(self.save_file_path))
        try:
            with FileLock(self.save_file_lock_path):
                with h5py.File(self.save_file_path, "w") as f:
                    f["controls"] = controls
                    f["cost_eval_step"] = self.cost_eval_step
                    f["costs"] = np.array(["{}".format(c) for c in self.costs])
                    f["evolution_time"] = self.evolution_time
                    f["initial_states"] = self.initial_states
                    f["interpolation_policy"] = "{}".format(self.interpolation_policy)
                    if self.save_intermediate_states_:
                        f["intermediate_states"] = np.zeros(
                            (self.system_eval_count, *self.initial_states.shape),
                            dtype=np.complex128)
                    f["magnus_policy"] = "{}".format(self.magnus_policy)
                    f["method"] = self.method
                    f["program_type"] = self.program_type.value
                    f["system_eval_count"] = self.system_eval_count
        except Timeout:
            print("Timeout while locking {}.".format(self.save_file_lock_path))

    def save_all_intermediate_states(self, iteration, step_states):
        """step_states :: (system_eval_count x state_count x hilbert_size x 1)."""
        if self.save_file_path is None:
            return
        h5py, FileLock, Timeout = _h5()
        try:
            with FileLock(self.save_file_lock_path):
                with h5py.File(self.save_file_path, "a") as f:
                    f["intermediate_states"][...] = step_states.astype(np.complex128)
        except Timeout:
            print("Timeout while locking {} while saving intermediate states on iteration {}."
                  "".format(self.save_file_lock_path, iteration))


class EvolveSchroedingerResult(object):
    def __init__(self, error=None, final_states=None):
        super().__init__()
        self.error = error
        self.final_states = final_states


class GrapeSchroedingerDiscreteState(GrapeState):
    method = "grape_schroedinger_discrete"

    def __init__(self, complex_controls, control_count, control_eval_count, cost_eval_step, costs,
                 evolution_time, hamiltonian, impose_control_conditions, initial_controls,
                 initial_states, interpolation_policy, iteration_count, log_iteration_step,
                 max_control_norms, magnus_policy, min_error, optimizer, save_file_path,
                 save_intermediate_states_, save_iteration_step, system_eval_count):
        super().__init__(complex_controls, control_count, control_eval_count, cost_eval_step,
                         costs, evolution_time, hamiltonian, impose_control_conditions,
                         initial_controls, interpolation_policy, iteration_count,
                         log_iteration_step, max_control_norms, min_error, optimizer,
                         save_file_path, save_iteration_step, system_eval_count)
        self.hilbert_size = initial_states[0].shape[0]
        self.initial_states = initial_states
        self.magnus_policy = magnus_policy
        self.save_intermediate_states_ = self.should_save and save_intermediate_states_

    def _is_save_iteration(self, iteration):
        return self.should_save and (np.mod(iteration, self.save_iteration_step) == 0
                                     or iteration == self.final_iteration)

    def log_and_save(self, controls, error, final_states, grads, iteration):
        if iteration > self.final_iteration:
            return
        if self.should_log and (np.mod(iteration, self.log_iteration_step) == 0
                                or iteration == self.final_iteration):
            print("{:^6d} | {:^1.8e} | {:^1.8e}".format(iteration, error, np.linalg.norm(grads)))
        if self._is_save_iteration(iteration):
            h5py, FileLock, Timeout = _h5()
            save_step, _ = np.divmod(iteration, self.save_iteration_step)
            try:
                with FileLock(self.save_file_lock_path):
                    with h5py.File(self.save_file_path, "a") as f:
                        f["controls"][save_step, ] = controls
                        f["error"][save_step, ] = error
                        f["final_states"][save_step, ] = final_states
                        f["grads"][save_step, ] = grads
            except Timeout:
                print("Timeout while locking {} to save after iteration {}."
                      "".format(self.save_file_lock_path, iteration))

    def log_and_save_initial(self):
        if self.should_save:
            h5py, FileLock, Timeout = _h5()
            print("QOC is saving this optimization run to {}.".format(self.save_file_path))
            save_count, remainder = np.divmod(self.iteration_count, self.save_iteration_step)
            if remainder != 0:
                save_count += 1
            state_count = len(self.initial_states)
            ctype = self.initial_controls.dtype
            try:
                with FileLock(self.save_file_lock_path):
                    with h5py.File(self.save_file_path, "w") as f:
                        f["complex_controls"] = self.complex_controls
                        f["control_count"] = self.control_count
                        f["control_eval_count"] = self.control_eval_count
                        f["controls"] = np.zeros((save_count, self.control_eval_count,
                                                  self.control_count), dtype=ctype)
                        f["cost_eval_step"] = self.cost_eval_step
                        f["cost_names"] = np.array([np.bytes_("{}".format(c)) for c in self.costs])
                        f["error"] = np.repeat(np.finfo(np.float64).max, save_count)
                        f["evolution_time"] = self.evolution_time
                        f["final_states"] = np.zeros((save_count, state_count, self.hilbert_size, 1),
                                                     dtype=np.complex128)
                        f["grads"] = np.zeros((save_count, self.control_eval_count,
                                               self.control_count), dtype=ctype)
                        f["initial_controls"] = self.initial_controls
                        f["initial_states"] = self.initial_states
                        if self.save_intermediate_states_:
                            f["intermediate_states"] = np.zeros(
                                (save_count, self.system_eval_count, *self.initial_states.shape),
                                dtype=np.complex128)
                        f["interpolation_policy"] = "{}".format(self.interpolation_policy)
                        f["iteration_count"] = self.iteration_count
                        f["magnus_policy"] = "{}".format(self.magnus_policy)
                        f["max_control_norms"] = self.max_control_norms
                        f["method"] = self.method
                        f["optimizer"] = "{}".format(self.optimizer)
                        f["program_type"] = self.program_type.value
                        f["system_eval_count"] = self.system_eval_count
            except Timeout:
                print("Timeout while locking {}, could not perform initial save."
                      "".format(self.save_file_lock_path))
        if self.should_log:
            print("iter   |   total error  |    grads_l2   \n"
                  "=========================================")

    def save_all_intermediate_states(self, iteration, step_states):
        if iteration > self.final_iteration or not self._is_save_iteration(iteration):
            return
        h5py, FileLock, Timeout = _h5()
        save_step, _ = np.divmod(iteration, self.save_iteration_step)
        try:
            with FileLock(self.save_file_lock_path):
                with h5py.File(self.save_file_path, "a") as f:
                    f["intermediate_states"][save_step] = step_states.astype(np.complex128)
        except Timeout:
            print("Timeout while locking {}, could not save intermediate states on iteration {}."
                  "".format(self.save_file_lock_path, iteration))


class GrapeSchroedingerResult(object):
    def __init__(self, best_controls=None, best_error=np.finfo(np.float64).max,
                 best_final_states=None, best_iteration=None):
        super().__init__()
        self.best_controls = best_controls
        self.best_error = best_error
        self.best_final_states = best_final_states
        self.best_iteration = best_iteration
