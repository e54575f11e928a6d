"""core - the entry points of the hot path (same names as qoc.core)."""

from .lindbladdiscrete import evolve_lindblad_discrete, grape_lindblad_discrete
from .schroedingerdiscrete import (evolve_schroedinger_discrete, grape_schroedinger_discrete,
                                   grape_schroedinger_discrete_batch)

__all__ = ["evolve_lindblad_discrete", "grape_lindblad_discrete",
           "evolve_schroedinger_discrete", "grape_schroedinger_discrete",
           "grape_schroedinger_discrete_batch"]
