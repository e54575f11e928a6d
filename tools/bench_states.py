"""
bench_states.py - secondary measurement: the C3 workload of bench.py with S initial states
(SURVEY.md 8d "secondary variant S = n": identity columns, random-unitary target columns).

    python tools/bench_states.py 1,4,32
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
    counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,4,32").split(",")]
    h0, g, _, _ = bench.make_problem()
    rng = np.random.default_rng(5)
    q, _ = np.linalg.qr(rng.standard_normal((bench.DIM, bench.DIM))
                        + 1j * rng.standard_normal((bench.DIM, bench.DIM)))
    u = bench.make_controls(0, 256)
    eng = Engine(0)
    eng.set_timing(True)
    for S in counts:
        psi0 = np.eye(bench.DIM, dtype=np.complex128)[:S]
        target = np.ascontiguousarray(q.T[:S])
        eng.set_schroedinger_problem(
            bench.DIM, S, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
            h0[None], np.stack(g)[None], psi0,
            costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
        eng.upload_controls(u)
        eng.eval_resident(True)
        eng.reset_timing()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            eng.eval_resident(True)
        wall = (time.perf_counter() - t0) / reps
        tm = {k: round(v[1] / reps, 2) for k, v in eng.timing().items() if v[0]}
        print(json.dumps(dict(states=S, ms_per_eval=round(wall * 1e3, 2),
                              steps_per_s=round(256 * 1000 / wall), kernel_ms_per_eval=tm)),
              flush=True)


if __name__ == "__main__":
    main()
