import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from tools import diaglib
diaglib.load()
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
engine = Engine(0)
h0, g, psi0, target = bench.make_problem()
engine.set_schroedinger_problem(bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
    h0[None], np.stack(g)[None], psi0, costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
engine.upload_controls(bench.make_controls(0, bench.SEEDS_PER_GPU))
def run(nseg, w):
    if w: os.environ["QOCX_SEG_WEIGHTS"] = w
    else: os.environ.pop("QOCX_SEG_WEIGHTS", None)
    engine.set_pipeline(nseg)
    for _ in range(3): engine.eval_resident(True)
    engine.synchronize()
    t = time.perf_counter()
    for _ in range(10): engine.eval_resident(True)
    engine.synchronize()
    print(json.dumps(dict(nseg=nseg, weights=w, ms=(time.perf_counter() - t) * 100)), flush=True)
run(0, "")
run(8, "1,1,1,0.5,0.5,1,1,1")
run(8, "0.5,1,1,0.5,0.5,1,1,0.5")
run(8, "0.5,1,1,0.75,0.75,1,1,0.5")
run(8, "0.6,1,1.2,0.5,0.5,1.2,1,0.6")
run(8, "1,1,1,0.3,0.3,1,1,1")
run(8, "1,1,1,1,1,1,1,1")
run(10, "1,1,1,1,0.5,0.5,1,1,1,1")
run(10, "0.5,1,1,1,0.5,0.5,1,1,1,0.5")
run(12, "1,1,1,1,1,0.5,0.5,1,1,1,1,1")
run(12, "0.5,1,1,1,1,0.5,0.5,1,1,1,1,0.5")
run(6, "1,1,0.5,0.5,1,1")
run(6, "0.7,1,0.5,0.5,1,0.7")
run(16, "1,1,1,1,1,1,1,0.5,0.5,1,1,1,1,1,1,1")
engine.close()
