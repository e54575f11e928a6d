"""
exp_two_halves.py - GPU-BOX EXPERIMENT: would two half-batches in flight, the second one a factor
phase behind the first, beat one batch of 256 seeds? Two engine contexts with 128 seeds each,
evaluated from two host threads; the second starts `delay` ms after the first. Prints the wall
time until both are done (median of the rounds) for each delay, and the one-context time.
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qoc_amd.engine import Engine, COST_TARGET_COHERENT  # noqa: E402


def make(seeds, first):
    eng = Engine(0)
    h0, g, psi0, target = bench.make_problem()
    eng.set_schroedinger_problem(
        bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
        h0[None], np.stack(g)[None], psi0,
        costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
    eng.upload_controls(bench.make_controls(first, seeds))
    eng.set_pipeline(8)
    for _ in range(3):
        eng.eval_resident(True)
    return eng


def main():
    one = make(256, 0)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        one.eval_resident(True)
        ts.append(time.perf_counter() - t0)
    print(json.dumps(dict(mode="one context, 256 seeds", ms=round(float(np.median(ts)) * 1e3, 3))), flush=True)
    a, b = make(128, 0), make(128, 128)
    for delay_ms in (0.0, 1.5, 2.5, 3.0, 3.5, 4.0, 5.0):
        walls = []
        for _ in range(12):
            done = [0.0, 0.0]

            def run(eng, idx, wait):
                if wait > 0:
                    t_end = time.perf_counter() + wait
                    while time.perf_counter() < t_end:
                        pass
                eng.eval_resident(True)
                done[idx] = time.perf_counter()
            t0 = time.perf_counter()
            th = [threading.Thread(target=run, args=(a, 0, 0.0)),
                  threading.Thread(target=run, args=(b, 1, delay_ms * 1e-3))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            walls.append(max(done) - t0)
        print(json.dumps(dict(mode="two contexts, 128 seeds each", delay_ms=delay_ms,
                              ms=round(float(np.median(walls)) * 1e3, 3),
                              ms_min=round(float(np.min(walls)) * 1e3, 3))), flush=True)


if __name__ == "__main__":
    main()
