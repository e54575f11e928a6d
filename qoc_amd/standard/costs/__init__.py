"""costs - the Cost plugins of the Schroedinger path (same names as qoc.standard.costs)."""

from .controlcosts import ControlArea, ControlBandwidthMax, ControlNorm, ControlVariation
from .statecosts import ForbidStates, TargetStateInfidelity, TargetStateInfidelityTime

__all__ = [
    "ControlArea", "ControlBandwidthMax", "ControlNorm", "ControlVariation",
    "ForbidStates", "TargetStateInfidelity", "TargetStateInfidelityTime",
]
