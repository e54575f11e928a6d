"""
bench_lindblad_ab.py - GPU-BOX TOOLING: bench.py's secondary (BASELINE configs[3]: n = 16, 500 steps,
64 seeds, L = 2) under values of an engine knob, interleaved rounds in one process: wall ms per
forward + gradient evaluation (host buffers to host buffers), the kernels' ms per launch, and the
largest relative difference of the gradients against the first variant.
    python tools/bench_lindblad_ab.py --knob lindblad_q2 --values 0 1 [--seeds 64] [--rounds 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", default="lindblad_q2")
    ap.add_argument("--values", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--seeds", type=int, default=bench.LB_SEEDS)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--diag", action="store_true",
                    help="load libqocx_diag.so (for a diagnostic knob; its kernels carry the experiments' tests "
                         "and are slower than the product's)")
    args = ap.parse_args()
    if args.diag:
        from tools import diaglib
        diaglib.load()
    from qoc_amd.engine import Engine, COST_TARGET_DENSITY
    engine = Engine(0)
    h0, g, gam, ops, rho0, target = bench.lindblad_problem()
    engine.set_lindblad_problem(
        bench.LB_DIM, 1, bench.K_CTRL, bench.LB_EVAL, bench.LB_EVAL,
        bench.DT * (bench.LB_EVAL - 1), h0, g, gam, ops, rho0,
        costs=[dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=1.0, vectors=target)])
    u = np.stack([0.1 * np.random.default_rng(1000 + b).standard_normal((bench.LB_EVAL, bench.K_CTRL))
                  for b in range(args.seeds)])
    times = {v: [] for v in args.values}
    kern = {}
    ref = None
    diffs = {}
    for rnd in range(args.rounds + 1):
        for v in args.values:
            engine.set_knob(args.knob, v)
            engine.reset_timing()
            t0 = time.perf_counter()
            cost, grads, final = engine.evaluate_lindblad(u)
            dt = time.perf_counter() - t0
            if rnd == 0:
                if ref is None:
                    ref = (cost.copy(), grads.copy(), final.copy())
                diffs[v] = {"grad_rel": float(np.max(np.abs(grads - ref[1])) / np.max(np.abs(ref[1]))),
                            "cost_abs": float(np.max(np.abs(cost - ref[0]))),
                            "final_abs": float(np.max(np.abs(final - ref[2])))}
                continue
            times[v].append(dt * 1e3)
            kern[v] = {k: t / max(n, 1) for k, (n, t) in engine.timing().items() if n}
    spans = {}
    for v in args.values:  # one more evaluation per variant with the HIP-event timeline on
        engine.set_knob(args.knob, v)
        engine.set_timing(True)
        engine.evaluate_lindblad(u)
        spans[v] = [(int(w), round(a, 3), round(b, 3)) for w, a, b in engine.timeline() if int(w) in (5, 6)]
        engine.set_timing(False)
    for v in args.values:
        print(json.dumps({"knob": args.knob, "value": v, "seeds": args.seeds,
                          "timeline_ms (5: pass, 6: combine)": spans[v],
                          "ms_median": float(np.median(times[v])), "ms_min": float(np.min(times[v])),
                          "msteps_per_s": args.seeds * (bench.LB_EVAL - 1) / np.median(times[v]) / 1e3,
                          "kernel_ms_per_launch": kern[v], "against_first_variant": diffs[v]}), flush=True)
    engine.close()


if __name__ == "__main__":
    main()
