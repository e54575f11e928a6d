"""scratch timing: per-iteration wall times (not part of the product)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT

h0, g, psi0, target = bench.make_problem()
eng = Engine(0)
eng.set_schroedinger_problem(bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL,
                             bench.DT * (bench.N_EVAL - 1), h0, np.stack(g)[None], psi0,
                             costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
u = bench.make_controls(0, 256)
eng.upload_controls(u)
eng.set_timing(True)
for mode in ("eval", "eval+download", "eval+download+sum"):
    ts = []
    for it in range(25):
        t0 = time.perf_counter()
        eng.eval_resident(True)
        t1 = time.perf_counter()
        if mode != "eval":
            c, gr, _ = eng.download_results(want_grad=True, want_final=False)
        t2 = time.perf_counter()
        if mode == "eval+download+sum":
            x = float(np.sum(c)), gr.sum(axis=0)
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    a = np.array(ts) * 1e3
    print(mode, "eval ms:", np.round(a[:, 0], 1).tolist())
    print("   download ms:", np.round(a[:, 1], 2).tolist())
    print("   sum ms:", np.round(a[:, 2], 2).tolist(), flush=True)
