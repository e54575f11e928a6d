"""models - data models of the host API (same public names as qoc.models)."""

from .cost import Cost
from .dummy import Dummy
from .lindbladmodels import (EvolveLindbladDiscreteState, EvolveLindbladResult,
                             GrapeLindbladDiscreteState, GrapeLindbladResult)
from .policies import (InterpolationPolicy, MagnusPolicy, OperationPolicy, PerformancePolicy,
                       ProgramType)
from .programstate import GrapeState, ProgramState
from .schroedingermodels import (EvolveSchroedingerDiscreteState, EvolveSchroedingerResult,
                                 GrapeSchroedingerDiscreteState, GrapeSchroedingerResult)

__all__ = [
    "Cost", "Dummy", "InterpolationPolicy", "MagnusPolicy", "OperationPolicy",
    "PerformancePolicy", "ProgramType", "ProgramState", "GrapeState",
    "EvolveLindbladDiscreteState", "EvolveLindbladResult", "GrapeLindbladDiscreteState",
    "GrapeLindbladResult",
    "EvolveSchroedingerDiscreteState", "EvolveSchroedingerResult",
    "GrapeSchroedingerDiscreteState", "GrapeSchroedingerResult",
]
