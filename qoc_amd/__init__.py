"""
qoc_amd - MI355X-native GRAPE propagation engine; drop-in for the evolve/grape hot path of
SchusterLab/qoc (host: NumPy; device: hand-written gfx950 HIP through a ctypes C ABI).
"""
