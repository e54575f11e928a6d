"""
bench_lindblad_sizes.py - secondary measurement: the Lindblad engine away from BASELINE configs[3] -
Hilbert sizes 16 / 24 / 32 (64 seeds x 200 steps, two operators, one final target cost: above n = 16
the tile-per-wave kernels of qocx_lindblad4t.hip, forward and unit adjoint side by side) and operator
counts 1..4 at n = 16 (64 seeds x 500 steps; one operator runs as two, the second zero).

    python tools/bench_lindblad_sizes.py            # -> profiles/r04_lindblad_above_n16.jsonl,
                                                    #    profiles/r04_lindblad_operator_count.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qoc_amd.engine import Engine, COST_TARGET_DENSITY  # noqa: E402


def problem(engine, n, count, steps, seeds):
    rng = np.random.default_rng(7)
    h0 = bench.gue(rng, n)
    g = [bench.gue(rng, n) for _ in range(2)]
    a = np.diag(np.sqrt(np.arange(1, n)), 1).astype(np.complex128)
    pool = [a, a.conj().T @ a, a @ a, a.conj().T]
    ops = np.stack(pool[:count])
    gam = np.array([0.05, 0.02, 0.01, 0.01][:count])
    rho0 = np.zeros((1, n, n), complex)
    rho0[0, 0, 0] = 1
    tgt = np.zeros((1, n, n), complex)
    tgt[0, 1, 1] = 1
    N = steps + 1
    engine.set_lindblad_problem(n, 1, 2, N, N, bench.DT * steps, h0, g, gam, ops, rho0,
                                costs=[dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=1.0, vectors=tgt)])
    return np.stack([0.1 * np.random.default_rng(1000 + b).standard_normal((N, 2)) for b in range(seeds)])


def measure(engine, u, reps=3):
    engine.evaluate_lindblad(u)
    t0 = time.perf_counter()
    for _ in range(reps):
        engine.evaluate_lindblad(u)
    return (time.perf_counter() - t0) / reps


def main():
    engine = Engine(0)
    for n in (16, 24, 32):
        u = problem(engine, n, 2, 200, 64)
        dt = measure(engine, u)
        print(json.dumps(dict(sweep="hilbert size", n=n, operators=2, steps=200, seeds=64, ms=round(dt * 1e3, 2),
                              subintervals_per_step=engine.lindblad_last_subintervals() / 64 / 200,
                              msteps_per_s=round(64 * 200 / dt / 1e6, 3))), flush=True)
    for count in (1, 2, 3, 4):
        u = problem(engine, 16, count, 500, 64)
        dt = measure(engine, u)
        print(json.dumps(dict(sweep="operator count", n=16, operators=count, steps=500, seeds=64,
                              ms=round(dt * 1e3, 2), msteps_per_s=round(64 * 500 / dt / 1e6, 3))), flush=True)


if __name__ == "__main__":
    main()
