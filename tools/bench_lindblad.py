"""
bench_lindblad.py - secondary measurement (not the driver's bench contract): the Lindblad
engine on BASELINE.json configs[3] (SURVEY.md 8d "C4": n=16, N=501, L=2 operators a and a^dag a
with gamma=(0.05, 0.02), K=2 real controls, S=1, dt=0.05), fwd+grad, for several batch sizes.
Prints one JSON line per batch size. (The CPU restatement is timed by bench.py's cpu_baseline
only: nothing under tools/ touches oracle/.)

    python tools/bench_lindblad.py --batches 64,1024 --reps 3
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_DIM, N_EVAL, K_CTRL, DT = 16, 501, 2, 0.05


def gue(rng, n):
    g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    h = (g + g.conj().T) / 2
    return h / np.linalg.norm(h, 2)


def problem():
    rng = np.random.default_rng(2004)
    h0 = gue(rng, N_DIM)
    g = [gue(rng, N_DIM) for _ in range(K_CTRL)]
    a = np.diag(np.sqrt(np.arange(1, N_DIM)), 1).astype(np.complex128)
    ops = np.stack([a, a.conj().T @ a])
    gam = np.array([0.05, 0.02])
    rho0 = np.zeros((1, N_DIM, N_DIM), dtype=np.complex128)
    rho0[0, 0, 0] = 1
    target = np.zeros((1, N_DIM, N_DIM), dtype=np.complex128)
    target[0, 1, 1] = 1
    return h0, g, gam, ops, rho0, target


def controls(count):
    out = np.empty((count, N_EVAL, K_CTRL))
    for b in range(count):
        out[b] = 0.1 * np.random.default_rng(1000 + b).standard_normal((N_EVAL, K_CTRL))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="64,1024")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from qoc_amd.engine import Engine, COST_TARGET_DENSITY
    h0, g, gam, ops, rho0, target = problem()
    eng = Engine(0)
    eng.set_lindblad_problem(N_DIM, 1, K_CTRL, N_EVAL, N_EVAL, DT * (N_EVAL - 1), h0, g, gam, ops,
                             rho0, costs=[dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=1.0,
                                               vectors=target)])
    eng.set_timing(True)
    for batch in [int(x) for x in args.batches.split(",")]:
        u = controls(batch)
        eng.evaluate_lindblad(u)  # warm-up
        eng.reset_timing()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            cost, grads, final = eng.evaluate_lindblad(u)
        wall = (time.perf_counter() - t0) / args.reps
        tm = eng.timing()["lindblad"]
        print(json.dumps(dict(
            workload="c4_lindblad_n16_N501", batch=batch, ms_per_eval=wall * 1e3,
            kernel_ms=tm[1] / max(tm[0], 1),
            steps_per_s=batch * (N_EVAL - 1) / wall, trace_defect=float(
                np.max(np.abs(np.trace(final[:, 0], axis1=-2, axis2=-1) - 1))),
            cost0=float(cost[0]))), flush=True)


if __name__ == "__main__":
    main()
