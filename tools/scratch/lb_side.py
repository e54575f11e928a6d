import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_DENSITY
h0, g, gam, ops, rho0, target = bench.lindblad_problem()
eng = Engine(0)
eng.set_lindblad_problem(bench.LB_DIM, 1, bench.K_CTRL, bench.LB_EVAL, bench.LB_EVAL, bench.DT * (bench.LB_EVAL - 1), h0, g, gam, ops, rho0,
    costs=[dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=1.0, vectors=target)])
eng.set_timing(True)
for seeds in (64, 32, 128):
    u = np.stack([0.1 * np.random.default_rng(1000 + b).standard_normal((bench.LB_EVAL, bench.K_CTRL)) for b in range(seeds)])
    for limit in (128, 0, 128, 0):
        eng.set_knob("lindblad_side_limit", limit)
        eng.evaluate_lindblad(u)
        eng.reset_timing()
        t0 = time.perf_counter()
        for _ in range(3): eng.evaluate_lindblad(u)
        wall = (time.perf_counter() - t0) / 3
        tm = eng.timing()
        print(json.dumps(dict(seeds=seeds, side_limit=limit, ms_per_eval=round(wall * 1e3, 3),
              pass_ms=round(tm["lindblad"][1] / max(tm["lindblad"][0], 1), 3), launches=tm["lindblad"][0])), flush=True)
