import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
knob = sys.argv[1] if len(sys.argv) > 1 else "sweep_umode_batch"
dims = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16]
for dim in dims:
    bench.DIM = dim
    engine = Engine(0)
    h0, g, psi0, target = bench.make_problem()
    engine.set_schroedinger_problem(bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
        h0[None], np.stack(g)[None], psi0, costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
    engine.upload_controls(bench.make_controls(0, 256))
    ref = None
    for label, v in ((knob + "=0", 0), (knob + "=1", 1), (knob + "=0 again", 0), (knob + "=1 again", 1)):
        engine.set_knob(knob, v)
        for _ in range(3): engine.eval_resident(True)
        engine.synchronize()
        ts = []
        for _ in range(4):
            t = time.perf_counter()
            for _ in range(5): engine.eval_resident(True)
            engine.synchronize()
            ts.append((time.perf_counter() - t) * 200)
        c, gr, _ = engine.download_results(want_grad=True, want_final=False)
        if ref is None: ref = (c.copy(), gr.copy())
        print(json.dumps(dict(dim=dim, label=label, ms_median=sorted(ts)[len(ts)//2])), "max diff", float(np.max(np.abs(c - ref[0]))), float(np.max(np.abs(gr - ref[1]))), "of", float(np.max(np.abs(ref[1]))), flush=True)
    engine.close()
