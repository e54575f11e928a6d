"""
statecosts.py - the three built-in state costs of the Schroedinger path.

Values follow qoc/standard/costs/targetstateinfidelity.py:39-63,
targetstateinfidelitytime.py:46-73 and forbidstates.py:50-81 (same constructor arguments,
attributes and cost() results, including the reference's spelling `neglect_relative_pahse`).
`cost()` is the host (NumPy) evaluation used by evolve-style callers and tests;
`device_descriptor()` hands the same formula to the HIP engine, which evaluates the value and
its cotangent on the GPU during GRAPE (include/qocx.h: qocx_cost_desc).
"""

import numpy as np

from qoc_amd.models.cost import Cost
from qoc_amd.standard.functions import conjugate_transpose

_KIND_COHERENT, _KIND_INCOHERENT, _KIND_FORBID = 0, 1, 2


def _abs2(z):
    return np.real(z * np.conjugate(z))


class TargetStateInfidelity(Cost):
    """Infidelity between the evolved states and their targets, at the final time."""
    name = "target_state_infidelity"
    requires_step_evaluation = False

    def __init__(self, target_states, neglect_relative_pahse=False, cost_multiplier=1.):
        super().__init__(cost_multiplier=cost_multiplier)
        self.state_count = target_states.shape[0]
        self.target_states_dagger = conjugate_transpose(target_states)
        self.neglect_relative_pahse = neglect_relative_pahse
        self._targets = np.asarray(target_states, dtype=np.complex128)

    def _infidelity(self, states):
        overlaps = np.matmul(self.target_states_dagger, states)[:, 0, 0]
        if self.neglect_relative_pahse == False:  # noqa: E712 (mirrors the reference's test)
            total = np.sum(overlaps)
            fidelity = _abs2(total) / self.state_count ** 2
        else:
            self.inner_products = overlaps
            fidelity = np.sum(_abs2(overlaps)) / self.state_count
        return 1 - fidelity

    def cost(self, controls, states, system_eval_step):
        return self._infidelity(states) * self.cost_multiplier

    def _scale(self):
        return self.cost_multiplier

    def device_descriptor(self, state_count, hilbert_size):
        if self.state_count != state_count:
            raise ValueError("{}: {} target states for {} evolving states"
                             "".format(self, self.state_count, state_count))
        return dict(kind=_KIND_INCOHERENT if self.neglect_relative_pahse else _KIND_COHERENT,
                    step_cost=int(self.requires_step_evaluation), scale=float(self._scale()),
                    vectors=self._targets.reshape(self.state_count, hilbert_size))


class TargetStateInfidelityTime(TargetStateInfidelity):
    """The same infidelity, accumulated at every cost evaluation step."""
    name = "target_state_infidelity_time"
    requires_step_evaluation = True

    def __init__(self, system_eval_count, target_states, neglect_relative_pahse=False,
                 cost_eval_step=1, cost_multiplier=1.):
        super().__init__(np.stack(target_states), neglect_relative_pahse=neglect_relative_pahse,
                         cost_multiplier=cost_multiplier)
        self.cost_eval_count, _ = np.divmod(system_eval_count - 1, cost_eval_step)

    def cost(self, controls, states, system_eval_step):
        return self._infidelity(states) / self.cost_eval_count * self.cost_multiplier

    def _scale(self):
        return self.cost_multiplier / self.cost_eval_count


class ForbidStates(Cost):
    """Occupation of forbidden states, accumulated at every cost evaluation step."""
    name = "forbid_states"
    requires_step_evaluation = True

    def __init__(self, forbidden_states, system_eval_count, cost_eval_step=1, cost_multiplier=1.):
        super().__init__(cost_multiplier=cost_multiplier)
        state_count = forbidden_states.shape[0]
        cost_evaluation_count, _ = np.divmod(system_eval_count - 1, cost_eval_step)
        self.cost_normalization_constant = cost_evaluation_count * state_count
        self.forbidden_states_count = np.array([f.shape[0] for f in forbidden_states])
        self.forbidden_states_dagger = conjugate_transpose(forbidden_states)
        self._forbidden = [np.asarray(f, dtype=np.complex128) for f in forbidden_states]

    def cost(self, controls, states, system_eval_step):
        total = 0
        for i, daggers in enumerate(self.forbidden_states_dagger):
            occupation = 0
            for dagger in daggers:
                occupation = occupation + _abs2(np.matmul(dagger, states[i])[0, 0])
            total = total + occupation / self.forbidden_states_count[i]
        return total / self.cost_normalization_constant * self.cost_multiplier

    def device_descriptor(self, state_count, hilbert_size):
        if len(self._forbidden) != state_count:
            raise ValueError("{}: forbidden sets for {} states, {} evolving states"
                             "".format(self, len(self._forbidden), state_count))
        vectors = np.concatenate([f.reshape(f.shape[0], hilbert_size) for f in self._forbidden])
        return dict(kind=_KIND_FORBID, step_cost=1,
                    scale=float(self.cost_multiplier / self.cost_normalization_constant),
                    vectors=vectors, counts=[int(c) for c in self.forbidden_states_count])
