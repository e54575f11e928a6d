"""costs - the Cost plugins of the hot paths (same names as qoc.standard.costs)."""

from .controlcosts import ControlArea, ControlBandwidthMax, ControlNorm, ControlVariation
from .densitycosts import ForbidDensities, TargetDensityInfidelity, TargetDensityInfidelityTime
from .statecosts import ForbidStates, TargetStateInfidelity, TargetStateInfidelityTime

__all__ = [
    "ControlArea", "ControlBandwidthMax", "ControlNorm", "ControlVariation",
    "ForbidDensities", "ForbidStates", "TargetDensityInfidelity", "TargetDensityInfidelityTime",
    "TargetStateInfidelity", "TargetStateInfidelityTime",
]
