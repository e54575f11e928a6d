"""
bench_magnus.py - secondary measurement: the C3 workload of bench.py under MagnusPolicy M2 / M4 /
M6 (one evaluation = cost + gradient of 256 seeds x 1000 steps, n = 32).

    python tools/bench_magnus.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qoc_amd.engine import Engine, COST_TARGET_COHERENT  # noqa: E402


def main():
    h0, g, psi0, target = bench.make_problem()
    u = bench.make_controls(0, 256)
    eng = Engine(0)
    eng.set_timing(True)
    for policy in ("M2", "M4", "M6"):
        eng.set_schroedinger_problem(
            bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
            h0[None], np.stack(g)[None], psi0,
            costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)],
            magnus_policy=policy)
        eng.upload_controls(u)
        for _ in range(3):
            eng.eval_resident(True)
        eng.reset_timing()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            eng.eval_resident(True)
        wall = (time.perf_counter() - t0) / reps
        cost, grads, _ = eng.download_results(want_grad=True, want_final=False)
        tm = {k: round(v[1] / reps, 2) for k, v in eng.timing().items() if v[0]}
        print(json.dumps(dict(policy=policy, ms_per_eval=round(wall * 1e3, 2),
                              steps_per_s=round(256 * 1000 / wall), kernel_ms_per_eval=tm,
                              sum_cost=float(cost.sum()))), flush=True)


if __name__ == "__main__":
    main()
