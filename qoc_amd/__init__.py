"""
qoc_amd - MI355X-native GRAPE propagation engine; drop-in for the evolve/grape hot path of
SchusterLab/qoc (host: NumPy; device: hand-written gfx950 HIP through a ctypes C ABI).

    from qoc_amd import evolve_schroedinger_discrete, grape_schroedinger_discrete
    from qoc_amd.standard import TargetStateInfidelity, Adam
"""

from .core import (evolve_lindblad_discrete, evolve_schroedinger_discrete,
                   grape_lindblad_discrete, grape_schroedinger_discrete,
                   grape_schroedinger_discrete_batch)

__all__ = ["evolve_lindblad_discrete", "evolve_schroedinger_discrete",
           "grape_lindblad_discrete", "grape_schroedinger_discrete",
           "grape_schroedinger_discrete_batch"]
