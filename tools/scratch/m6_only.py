import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
h0, g, psi0, target = bench.make_problem()
u = bench.make_controls(0, 256)
eng = Engine(0)
eng.set_timing(True)
eng.set_schroedinger_problem(bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
    h0[None], np.stack(g)[None], psi0, costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)],
    magnus_policy="M6")
eng.upload_controls(u)
for _ in range(3): eng.eval_resident(True)
eng.reset_timing()
t0 = time.perf_counter()
for _ in range(3): eng.eval_resident(True)
wall = (time.perf_counter() - t0) / 3
tm = {k: round(v[1] / 3, 2) for k, v in eng.timing().items() if v[0]}
print(json.dumps(dict(policy="M6", ms_per_eval=round(wall * 1e3, 2), kernel_ms_per_eval=tm)))
