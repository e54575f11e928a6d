# one evaluation of the general path at dim 128 (64 seeds x 1000 steps) under rocprofv3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bench.DIM = dim
engine = Engine(0)
h0, g, psi0, target = bench.make_problem()
engine.set_schroedinger_problem(dim, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
    h0[None], np.stack(g)[None], psi0, costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
engine.upload_controls(bench.make_controls(0, 64))
for _ in range(2):
    engine.eval_resident(True)
engine.synchronize()
engine.close()
