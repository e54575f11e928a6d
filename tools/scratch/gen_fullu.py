# full-propagator problems on the general path: n states of dimension n, forward + gradient
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
engine = Engine(0)
for dim, steps, seeds in ((100, 200, 1), (100, 200, 16), (128, 200, 1), (200, 100, 1)):
    bench.DIM = dim
    h0, g, _, _ = bench.make_problem()
    rng = np.random.default_rng(dim)
    q, _ = np.linalg.qr(rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim)))
    init = np.eye(dim, dtype=np.complex128)
    engine.set_schroedinger_problem(dim, dim, bench.K_CTRL, steps + 1, steps + 1, bench.DT * steps, h0[None], np.stack(g)[None], init,
        costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=q.T.copy())])
    engine.upload_controls(np.ascontiguousarray(bench.make_controls(0, seeds)[:, :steps + 1]))
    engine.set_timing(1)
    engine.eval_resident(True); engine.synchronize()
    engine.reset_timing()
    t = time.perf_counter()
    engine.eval_resident(True); engine.synchronize()
    ms = (time.perf_counter() - t) * 1e3
    print(json.dumps(dict(dim=dim, states=dim, steps=steps, seeds=seeds, ms_per_evaluation=ms, orders={k: v for k, v in engine.pade_orders().items() if v},
                          kernel_ms={k: v[1] for k, v in engine.timing().items() if v[0]})), flush=True)
engine.close()
