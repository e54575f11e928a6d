"""
sweep_stamps.py - GPU-BOX TOOLING: where the blocked sweep (qocx_sweep3.hip) spends its cycles.
Runs the C3 workload (bench.py's) with the stamped diagnostic build of the sweep (knob
"sweep3_stamps"; the product kernel executes no stamp) and prints, per role, the shader-clock
cycles per step and per phase of the role's loop, and the clock the chip held.
    python tools/sweep_stamps.py [--time-segments 1]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PHASES = {
    "compute": ["bookkeeping", "Q product (waits for Q)", "Q load issue", "solves + stores",
                "barrier", "epilogue", "-", "realtime"],
    "inverter_L": ["loop head", "inversion", "barrier", "-", "-", "-", "-", "realtime"],
    "loader": ["loop head", "issue", "landing wait", "barrier", "-", "-", "-", "realtime"],
    "inverter_U": ["loop head", "inversion", "barrier", "-", "-", "-", "-", "realtime"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time-segments", type=int, default=1)
    ap.add_argument("--seeds", type=int, default=bench.SEEDS_PER_GPU)
    ap.add_argument("--dbg", type=int, default=0)
    args = ap.parse_args()
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
    engine.set_pipeline(args.time_segments)
    engine.set_knob("sweep3_dbg", args.dbg)
    for _ in range(2):
        engine.eval_resident(True)
    engine.set_knob("sweep3_stamps", 1)
    engine.eval_resident(True)
    st = engine.read_stamps(args.seeds).astype(np.float64)
    steps = 2.0 * (bench.N_EVAL - 1)  # forward + adjoint passes
    out = {}
    for r, role in enumerate(("compute", "inverter_L", "loader", "inverter_U")):
        mean = st[:, r, :].mean(axis=0)
        cycles = mean[:7].sum()
        ghz = cycles / (mean[7] * 10.0) if mean[7] > 0 else 0.0  # realtime ticks are 10 ns
        out[role] = {"cycles_per_step": cycles / steps, "clock_GHz_over_stamped_part": ghz,
                     "phases_cycles_per_step": {PHASES[role][k]: mean[k] / steps
                                                for k in range(7) if PHASES[role][k] != "-"}}
    print(json.dumps(out, indent=1))
    engine.close()


if __name__ == "__main__":
    main()
