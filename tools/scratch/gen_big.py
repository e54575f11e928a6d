# the general path at its largest sizes: ms per forward + gradient evaluation, one control set, 10 steps, one state
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
engine = Engine(0)
for dim, steps, seeds in ((512, 10, 1), (1024, 10, 1), (1024, 10, 8)):
    bench.DIM = dim
    h0, g, psi0, target = bench.make_problem()
    engine.set_schroedinger_problem(dim, 1, bench.K_CTRL, steps + 1, steps + 1, bench.DT * steps, h0[None], np.stack(g)[None], psi0,
        costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
    engine.upload_controls(np.ascontiguousarray(bench.make_controls(0, seeds)[:, :steps + 1]))
    engine.set_timing(1)
    engine.eval_resident(True); engine.synchronize()
    engine.reset_timing()
    t = time.perf_counter()
    engine.eval_resident(True); engine.synchronize()
    ms = (time.perf_counter() - t) * 1e3
    print(json.dumps(dict(dim=dim, steps=steps, seeds=seeds, ms_per_evaluation=ms, ms_per_step=ms / (steps * seeds),
                          orders={k: v for k, v in engine.pade_orders().items() if v},
                          kernel_ms={k: v[1] for k, v in engine.timing().items() if v[0]})), flush=True)
engine.close()
