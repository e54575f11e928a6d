"""
lindblad_stamps.py - GPU-BOX TOOLING: where a stage of the multi-wave Lindblad kernel spends its
cycles (stamped diagnostic build, knob "lindblad_stamps"; the product kernel executes no stamp).
Workload: bench.py's secondary (BASELINE configs[3]: n = 16, 500 steps, 64 seeds, L = 2).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PHASES = ["stage value / kbar (axpys)", "generator + operands to LDS", "MFMA products",
          "partial result to LDS", "barrier", "sum of partial results",
          "control cotangents (two-sided stage loop: between sub-intervals)"]
WAVES = ["generator terms", "operator 1", "operator 2", "control cotangents"]


def main():
    from tools import diaglib
    diaglib.load()  # the measurement build: diagnostic knobs (qocx_diag.h)
    from qoc_amd.engine import Engine, COST_TARGET_DENSITY
    engine = Engine(0)
    h0, g, gam, ops, rho0, target = bench.lindblad_problem()
    engine.set_lindblad_problem(
        bench.LB_DIM, 1, bench.K_CTRL, bench.LB_EVAL, bench.LB_EVAL,
        bench.DT * (bench.LB_EVAL - 1), h0, g, gam, ops, rho0,
        costs=[dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=1.0, vectors=target)])
    seeds = bench.LB_SEEDS
    u = np.stack([0.1 * np.random.default_rng(1000 + b).standard_normal((bench.LB_EVAL, bench.K_CTRL))
                  for b in range(seeds)])
    engine.evaluate_lindblad(u)
    engine.set_knob("lindblad_stamps", 1)
    engine.evaluate_lindblad(u)
    raw = engine.read_stamps(2 * seeds, roles=6).astype(np.float64)
    subs = engine.lindblad_last_subintervals() / seeds
    out = {}
    # two-sided evaluation: the forward pass and the unit adjoint are launches of their own
    passes = [("forward pass", raw[:seeds], 12.0 * subs), ("unit adjoint", raw[seeds:], 12.0 * subs)]
    if raw[seeds:].sum() == 0:
        passes = [("forward + adjoint (one launch)", raw[:seeds], 24.0 * subs)]
    for title, st, stages in passes:
        out[title] = {}
        for wv, name in enumerate(WAVES):
            mean = st[:, wv, :].mean(axis=0)
            cycles = mean[:7].sum()
            out[title][name] = {"cycles_per_stage": cycles / stages,
                                "clock_GHz": cycles / (mean[7] * 10.0) if mean[7] > 0 else 0.0,
                                "phases": {PHASES[k]: round(mean[k] / stages, 1) for k in range(7)}}
    print(json.dumps(out, indent=1))
    engine.close()


if __name__ == "__main__":
    main()
