import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
engine = Engine(0)
h0, g, psi0, target = bench.make_problem()
engine.set_schroedinger_problem(bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
    h0[None], np.stack(g)[None], psi0, costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
engine.upload_controls(bench.make_controls(0, bench.SEEDS_PER_GPU))
def run(label, nseg=0, **knobs):
    for k, v in knobs.items(): engine.set_knob(k, v)
    engine.set_pipeline(nseg)
    for _ in range(3): engine.eval_resident(True)
    engine.synchronize()
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        for _ in range(5): engine.eval_resident(True)
        engine.synchronize()
        ts.append((time.perf_counter() - t) * 200)
    print(json.dumps(dict(label=label, nseg=nseg, knobs=knobs, ms_median=sorted(ts)[len(ts)//2], ms_min=min(ts))), flush=True)
ref = None
for label, v in (("adj_first=0", 0), ("adj_first=1", 1), ("adj_first=0 again", 0), ("adj_first=1 again", 1), ("adj_first=0 third", 0), ("adj_first=1 third", 1)):
    run(label, bidir_adj_first=v)
    c, g, _ = engine.download_results(want_grad=True, want_final=False)
    if ref is None: ref = (c.copy(), g.copy())
    print("   identical:", bool(np.array_equal(c, ref[0]) and np.array_equal(g, ref[1])), flush=True)
engine.close()
