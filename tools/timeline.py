"""
timeline.py - GPU-BOX TOOLING: the kernel launches of one C3 evaluation (bench.py's workload) on
a common time axis, from HIP events on the launch streams (qocx_debug_timeline).

    python tools/timeline.py [--knob name=value ...] > gpurun_out/timeline.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = {0: "K1a", 1: "sweep", 2: "K3", 3: "scatter", 4: "K1b", 5: "lindblad"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", action="append", default=[])
    ap.add_argument("--seeds", type=int, default=bench.SEEDS_PER_GPU)
    ap.add_argument("--dim", type=int, default=bench.DIM, help="Hilbert size (GUE problem of bench.py)")
    ap.add_argument("--diag", action="store_true", help="load libqocx_diag.so (for a diagnostic knob)")
    args = ap.parse_args()
    bench.DIM = args.dim
    if args.diag:
        from tools import diaglib
        diaglib.load()  # the measurement build: diagnostic knobs (qocx_diag.h)
    from qoc_amd.engine import Engine, COST_TARGET_COHERENT
    engine = Engine(0)
    h0, g, psi0, target = bench.make_problem()
    engine.set_schroedinger_problem(
        bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
        h0[None], np.stack(g)[None], psi0,
        costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
    engine.upload_controls(bench.make_controls(0, args.seeds))
    for kv in args.knob:
        k, v = kv.split("=")
        engine.set_knob(k, int(v))
    engine.set_timing(True)
    for _ in range(4):
        engine.eval_resident(True)
    tl = engine.timeline()
    rows = [dict(kernel=NAMES[int(w)], start=round(a, 3), end=round(b, 3)) for w, a, b in tl]
    rows.sort(key=lambda r: r["start"])
    print(json.dumps(dict(knobs=args.knob, launches=rows)))
    for r in rows:
        print("{:8s} {:7.3f} -> {:7.3f}  ({:.3f})".format(r["kernel"], r["start"], r["end"],
                                                        r["end"] - r["start"]), file=sys.stderr)
    engine.close()


if __name__ == "__main__":
    main()
