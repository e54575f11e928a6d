"""
densitycosts.py - the three built-in density-matrix costs of the Lindblad path.

Values follow qoc/standard/costs/targetdensityinfidelity.py:41-69,
targetdensityinfidelitytime.py:38-77 and forbiddensities.py:53-85 (same constructor
arguments, attributes and cost() results). Two reference behaviours kept on purpose:
the fidelity is |tr(T^H rho)| (not squared), and TargetDensityInfidelityTime has
requires_step_evaluation = False, i.e. it is evaluated once at the final time although it is
normalised by the cost evaluation count.
`device_descriptor()` hands the formulas to the HIP engine (include/qocx.h kinds 3 and 4).
"""

import numpy as np

from qoc_amd.models.cost import Cost
from qoc_amd.standard.functions import conjugate_transpose

_KIND_TARGET_DENSITY, _KIND_FORBID_DENSITY = 3, 4


class TargetDensityInfidelity(Cost):
    """Infidelity between the evolved densities and their targets, at the final time."""
    name = "target_density_infidelity"
    requires_step_evaluation = False

    def __init__(self, target_densities, cost_multiplier=1.):
        super().__init__(cost_multiplier=cost_multiplier)
        target_densities = np.stack(target_densities)
        self.density_count = target_densities.shape[0]
        self.hilbert_size = target_densities.shape[1]
        self.target_densities_dagger = conjugate_transpose(target_densities)
        self._targets = np.asarray(target_densities, dtype=np.complex128)

    def _infidelity(self, densities):
        overlaps = np.einsum("sij,sij->s", np.conjugate(self._targets), densities)
        fidelity = np.sum(np.abs(overlaps)) / (self.density_count * self.hilbert_size)
        return 1 - fidelity

    def _scale(self):
        return self.cost_multiplier

    def cost(self, controls, densities, system_eval_step):
        return self._infidelity(np.asarray(densities)) * self._scale()

    def device_descriptor(self, density_count, hilbert_size):
        if self.density_count != density_count or self.hilbert_size != hilbert_size:
            raise ValueError("{}: targets of shape {} for {} evolving densities of size {}"
                             "".format(self, self._targets.shape, density_count, hilbert_size))
        return dict(kind=_KIND_TARGET_DENSITY, step_cost=0, scale=float(self._scale()),
                    vectors=self._targets)


class TargetDensityInfidelityTime(TargetDensityInfidelity):
    name = "target_density_infidelity_time"
    requires_step_evaluation = False

    def __init__(self, system_eval_count, target_densities, cost_eval_step=1,
                 cost_multiplier=1.):
        super().__init__(target_densities, cost_multiplier=cost_multiplier)
        self.cost_eval_count, _ = np.divmod(system_eval_count - 1, cost_eval_step)

    def _scale(self):
        return self.cost_multiplier / self.cost_eval_count


class ForbidDensities(Cost):
    """Overlap with forbidden densities, accumulated at every cost evaluation step."""
    name = "forbid_densities"
    requires_step_evaluation = True

    def __init__(self, forbidden_densities, system_eval_count, cost_eval_step=1,
                 cost_multiplier=1.):
        super().__init__(cost_multiplier=cost_multiplier)
        density_count = len(forbidden_densities)
        cost_evaluation_count, _ = np.divmod(system_eval_count - 1, cost_eval_step)
        self.cost_normalization_constant = cost_evaluation_count * density_count
        self.forbidden_densities_count = np.array([len(f) for f in forbidden_densities])
        self._forbidden = [np.asarray(f, dtype=np.complex128) for f in forbidden_densities]
        self.forbidden_densities_dagger = [conjugate_transpose(f) for f in self._forbidden]
        self.hilbert_size = self._forbidden[0].shape[-1]

    def cost(self, controls, densities, system_eval_step):
        total = 0
        for i, forbidden in enumerate(self._forbidden):
            overlaps = np.einsum("fij,ij->f", np.conjugate(forbidden), densities[i])
            overlaps = overlaps / self.hilbert_size
            total = total + (np.sum(np.real(overlaps * np.conjugate(overlaps)))
                             / self.forbidden_densities_count[i])
        return total / self.cost_normalization_constant * self.cost_multiplier

    def device_descriptor(self, density_count, hilbert_size):
        if len(self._forbidden) != density_count or self.hilbert_size != hilbert_size:
            raise ValueError("{}: forbidden sets for {} densities of size {}, evolving {} of "
                             "size {}".format(self, len(self._forbidden), self.hilbert_size,
                                              density_count, hilbert_size))
        return dict(kind=_KIND_FORBID_DENSITY, step_cost=1,
                    scale=float(self.cost_multiplier / self.cost_normalization_constant),
                    vectors=np.concatenate(self._forbidden),
                    counts=[int(c) for c in self.forbidden_densities_count])
