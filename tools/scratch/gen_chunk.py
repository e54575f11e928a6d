# memory chunking of the general path at a size that does not fit: n = 256, 1000 steps, 160 seeds (2.1 GB of factors per seed)
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
dim, seeds = 256, 160
bench.DIM = dim
engine = Engine(0)
h0, g, psi0, target = bench.make_problem()
engine.set_schroedinger_problem(dim, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
    h0[None], np.stack(g)[None], psi0, costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
controls = bench.make_controls(0, seeds)
t = time.perf_counter()
cost, grads, final = engine.evaluate(controls, True)
ms = (time.perf_counter() - t) * 1e3
c1, g1, f1 = engine.evaluate(controls[150:151], True)
print(json.dumps(dict(dim=dim, seeds=seeds, ms=ms, ksteps_per_s=seeds * 1000 / ms, same_cost=bool(c1[0] == cost[150]),
                      same_grads=bool(np.array_equal(g1[0], grads[150])), same_final=bool(np.array_equal(f1[0], final[150])),
                      finite=bool(np.all(np.isfinite(cost)) and np.all(np.isfinite(grads))))))
engine.close()
