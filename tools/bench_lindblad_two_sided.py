import json, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_DENSITY
e = Engine(0)
h0, g, gam, ops, rho0, target = bench.lindblad_problem()
e.set_lindblad_problem(bench.LB_DIM, 1, bench.K_CTRL, bench.LB_EVAL, bench.LB_EVAL, bench.DT*(bench.LB_EVAL-1), h0, g, gam, ops, rho0, costs=[dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=1.0, vectors=target)])
for B in (16, 64, 128, 192, 256, 512, 1024):
    u = np.stack([0.1*np.random.default_rng(1000+b).standard_normal((bench.LB_EVAL, bench.K_CTRL)) for b in range(B)])
    out = {}
    for ts in (0, 1, 2):
        e.set_knob("lindblad_two_sided", 1 if ts else 0)
        e.set_knob("lindblad_side_limit", 100000 if ts == 2 else 128)
        e.evaluate_lindblad(u)
        t0 = time.perf_counter()
        for _ in range(3): c, gr, f = e.evaluate_lindblad(u)
        out[ts] = (time.perf_counter()-t0)/3*1e3
        if ts == 1:
            e.set_timing(True); e.evaluate_lindblad(u); tl = e.timeline(); e.set_timing(False)
            out["tl"] = [(int(w), round(a, 2), round(b, 2)) for w, a, b in tl]
        out["g%d" % ts] = gr
    rel = np.max(np.abs(out["g0"]-out["g1"]))/np.max(np.abs(out["g0"]))
    print(json.dumps(dict(batch=B, classic_ms=round(out[0],2), two_sided_ms=round(out[1],2), two_sided_always_concurrent_ms=round(out[2],2), msteps_classic=round(B*500/out[0]/1e3,2), msteps_two_sided=round(B*500/out[1]/1e3,2), grad_rel_diff=float(rel), timeline_two_sided=out.get("tl"))), flush=True)
