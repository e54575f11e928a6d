"""core - the entry points of the hot path (same names as qoc.core)."""

from .schroedingerdiscrete import evolve_schroedinger_discrete, grape_schroedinger_discrete

__all__ = ["evolve_schroedinger_discrete", "grape_schroedinger_discrete"]
