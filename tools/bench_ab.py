"""
bench_ab.py - BUILD-CONTAINER / GPU-BOX TOOLING: interleaved A/B rounds of the C3 workload
(bench.py's) in ONE process (cdna_hip_programming.md rule 24), switching engine knobs
(qocx_debug_set_knob) between rounds.

    python tools/bench_ab.py --knob sweep_loader --values 0 1 --rounds 6
Prints one JSON line per variant: median / min ms per evaluation and per-kernel ms per launch.
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
    ap.add_argument("--knob", action="append", default=[])
    ap.add_argument("--values", nargs="+", action="append", default=[],
                    help="one list per --knob; variants are the zip of the lists")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--evals", type=int, default=4)
    ap.add_argument("--seeds", type=int, default=bench.SEEDS_PER_GPU)
    ap.add_argument("--time-segments", type=int, default=0)
    ap.add_argument("--magnus", default="M2", choices=["M2", "M4", "M6"])
    ap.add_argument("--dim", type=int, default=bench.DIM, help="Hilbert size (GUE problem of bench.py)")
    ap.add_argument("--states", type=int, default=1,
                    help="S > 1: orthonormal random initial states and targets")
    args = ap.parse_args()
    from tools import diaglib
    diaglib.load()  # the measurement build: diagnostic knobs (qocx_diag.h)
    from qoc_amd.engine import Engine, COST_TARGET_COHERENT

    engine = Engine(0)
    bench.DIM = args.dim
    h0, g, psi0, target = bench.make_problem()
    if args.states > 1:
        rng = np.random.default_rng(7)
        q = lambda: np.linalg.qr(rng.standard_normal((bench.DIM, bench.DIM))  # noqa: E731
                                 + 1j * rng.standard_normal((bench.DIM, bench.DIM)))[0]
        psi0, target = q()[:args.states], q()[:args.states]
    engine.set_schroedinger_problem(
        bench.DIM, args.states, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL,
        bench.DT * (bench.N_EVAL - 1), h0[None], np.stack(g)[None], psi0,
        costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)],
        magnus_policy=args.magnus)
    engine.upload_controls(bench.make_controls(0, args.seeds))
    if args.time_segments:
        engine.set_pipeline(args.time_segments)
    variants = list(zip(*[[int(v) for v in vals] for vals in args.values])) or [()]
    engine.set_timing(True)
    results = {v: dict(ms=[], kernels={}, check=None) for v in variants}

    def apply(variant):
        for name, value in zip(args.knob, variant):
            engine.set_knob(name, value)

    for variant in variants:  # warm every variant
        apply(variant)
        for _ in range(2):
            engine.eval_resident(True)
    for _ in range(args.rounds):
        for variant in variants:
            apply(variant)
            engine.eval_resident(True)
            engine.reset_timing()
            engine.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.evals):
                engine.eval_resident(True)
                cost, grads, _ = engine.download_results(want_grad=True, want_final=False)
            engine.synchronize()
            results[variant]["ms"].append((time.perf_counter() - t0) / args.evals * 1e3)
            for k, (launches, ms) in engine.timing().items():
                if launches:
                    results[variant]["kernels"].setdefault(k, []).append(ms / launches)
            results[variant]["check"] = (float(cost.sum()), float(np.linalg.norm(grads.sum(axis=0))))
    for variant in variants:
        r = results[variant]
        print(json.dumps(dict(
            knobs=dict(zip(args.knob, variant)), ms_median=float(np.median(r["ms"])),
            ms_min=float(np.min(r["ms"])),
            kernel_ms_per_launch={k: float(np.median(v)) for k, v in r["kernels"].items()},
            check=r["check"], seeds=args.seeds)))
    engine.close()


if __name__ == "__main__":
    main()
