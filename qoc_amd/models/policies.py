"""
policies.py - the enumerations callers pass to the entry points.

Mirrors (names, members, integer values and str() forms)
qoc/models/magnuspolicy.py:8-26, interpolationpolicy.py:8-20, programtype.py:7-25,
operationpolicy.py:8-28 and performancepolicy.py:8-25 of the reference, so that strings
written to save files and user code comparing members keep working.
"""

from enum import Enum


class _NamedEnum(Enum):
    """Enum whose str()/repr() is a fixed label per member."""

    def __str__(self):
        return self._labels()[self.value]

    def __repr__(self):
        return str(self)


class MagnusPolicy(_NamedEnum):
    """Order of the Magnus expansion used for the step generator (arXiv:1709.06483)."""
    M2 = 1
    M4 = 2
    M6 = 3

    def _labels(self):
        return {1: "magnus_m2", 2: "magnus_m4", 3: "magnus_m6"}

    @property
    def short(self):
        return {1: "M2", 2: "M4", 3: "M6"}[self.value]

    @property
    def nodes(self):
        """Quadrature nodes c_q (t = time + c_q dt), qoc/core/mathmethods.py:70, :96-97, :125-127."""
        return {1: (0.5,),
                2: (0.5 - 3 ** 0.5 / 6, 0.5 + 3 ** 0.5 / 6),
                3: (0.5 - 15 ** 0.5 / 10, 0.5, 0.5 + 15 ** 0.5 / 10)}[self.value]


class InterpolationPolicy(_NamedEnum):
    """How time-discrete controls are evaluated between their grid points."""
    LINEAR = 1

    def _labels(self):
        return {1: "interpolation_linear"}


class ProgramType(_NamedEnum):
    EVOLVE = 1
    GRAPE = 2

    def _labels(self):
        return {1: "evolve", 2: "grape"}


class OperationPolicy(_NamedEnum):
    """Kept for signature compatibility (Adam(operation_policy=...)); the engine is always GPU."""
    CPU = 1
    GPU = 2
    CPU_SPARSE = 3
    GPU_SPARSE = 4

    def _labels(self):
        return {1: "operation_policy_cpu", 2: "operation_policy_gpu",
                3: "operation_policy_cpu_sparse", 4: "operation_policy_gpu_sparse"}


class PerformancePolicy(_NamedEnum):
    TIME = 1
    MEMORY = 2

    def _labels(self):
        return {1: "performance_policy_time", 2: "performance_policy_memory"}
