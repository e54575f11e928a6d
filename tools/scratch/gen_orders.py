import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
for dim in [int(x) for x in sys.argv[1].split(",")]:
    bench.DIM = dim
    engine = Engine(0)
    h0, g, psi0, target = bench.make_problem()
    engine.set_schroedinger_problem(dim, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
        h0[None], np.stack(g)[None], psi0, costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
    engine.upload_controls(bench.make_controls(0, 64))
    engine.set_timing(1)
    for po in (0, 13, 0, 13):
        engine.set_knob("pade_order", po)
        engine.eval_resident(True); engine.synchronize()
        engine.reset_timing()
        t = time.perf_counter()
        engine.eval_resident(True); engine.synchronize()
        ms = (time.perf_counter() - t) * 1e3
        print(json.dumps(dict(dim=dim, pade_order=po, ms=ms, orders=engine.pade_orders(), kernels={k: v[1] for k, v in engine.timing().items() if v[0]})), flush=True)
    engine.close()
