"""
k1a_stamps.py - GPU-BOX TOOLING: where the two waves of the two-wave Pade kernel (K1a,
qocx_pade2.hip) spend their cycles inside the pipeline of the C3 workload (bench.py's). Runs the
stamped build (libqocx_diag.so, knob "k1a_stamps"; the product kernel executes no stamp) and prints
per wave the shader-clock cycles per matrix and phase, and the clock the chip held.
    python tools/k1a_stamps.py [--knob name=value ...]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PHASES = ["generator, norm, staging", "products", "barrier waits", "P / Q images",
          "factorisation (MFMA form: stores, fragments, MFMA issue)", "factorisation: panels through LDS",
          "factorisation: four pivots per block", "realtime"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", action="append", default=[])
    ap.add_argument("--seeds", type=int, default=bench.SEEDS_PER_GPU)
    args = ap.parse_args()
    from tools import diaglib
    diaglib.load()
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
    for _ in range(2):
        engine.eval_resident(True)
    engine.set_knob("k1a_stamps", 1)
    engine.eval_resident(True)
    engine.synchronize()
    # the evaluation zeroes the sums before its first K1a launch and every launch adds to them
    st = engine.read_stamps(1024, roles=2).astype(np.float64).sum(axis=0)  # 1024 sets of sums
    matrices = float(args.seeds) * (bench.N_EVAL - 1)
    out = {"knobs": args.knob, "matrices": matrices}
    for w in range(2):
        cyc = st[w, :7] / matrices
        life = cyc.sum()
        ghz = st[w, :7].sum() / (st[w, 7] * 10.0) if st[w, 7] > 0 else 0.0
        out["wave%d" % w] = {"cycles_per_matrix": life, "clock_GHz": ghz,
                             "phases": {PHASES[k]: cyc[k] for k in range(7)}}
    print(json.dumps(out, indent=1))
    engine.close()


if __name__ == "__main__":
    main()
