"""
two_groups.py - GPU-BOX TOOLING: the C3 workload (bench.py's) as G independent groups of seeds, each
group a context of its own on the same GPU, evaluations issued round-robin without a host
synchronisation in between: the latency-bound tail of one group's evaluation (two sweep chains, K3)
overlaps the throughput-bound factor phase of the next group's.
    python tools/two_groups.py [--groups 2] [--seeds 256] [--rounds 20]
Prints one JSON line per group count: ms per round (= one evaluation of every group) and
propagator-steps/s.
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
    ap.add_argument("--groups", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--seeds", type=int, default=bench.SEEDS_PER_GPU)
    ap.add_argument("--rounds", type=int, default=20)
    args = ap.parse_args()
    from qoc_amd.engine import Engine, COST_TARGET_COHERENT
    h0, g, psi0, target = bench.make_problem()
    controls = bench.make_controls(0, args.seeds)
    for G in args.groups:
        per = args.seeds // G
        engines = []
        for k in range(G):
            e = Engine(0)
            e.set_schroedinger_problem(
                bench.DIM, 1, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
                h0[None], np.stack(g)[None], psi0,
                costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
            e.upload_controls(controls[k * per:(k + 1) * per])
            engines.append(e)
        for _ in range(3):
            for e in engines:
                e.eval_resident(True)
        for e in engines:
            e.synchronize()
        # qocx_eval_resident returns when the evaluation's status word has arrived: one host thread per
        # group keeps the groups' evaluations in flight together (ctypes releases the GIL)
        import threading
        barrier = threading.Barrier(G + 1)

        def worker(e):
            barrier.wait()
            for _ in range(args.rounds):
                e.eval_resident(True)
            e.synchronize()

        threads = [threading.Thread(target=worker, args=(e,)) for e in engines]
        for t in threads:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        for t in threads:
            t.join()
        dt = (time.perf_counter() - t0) / args.rounds
        costs = np.concatenate([np.asarray(e.download_costs()).reshape(-1) for e in engines])
        print(json.dumps({"groups": G, "seeds_per_group": per, "ms_per_round": dt * 1e3,
                          "msteps_per_s": per * G * (bench.N_EVAL - 1) / dt / 1e6,
                          "sum_cost": float(costs.sum())}), flush=True)
        for e in engines:
            e.close()


if __name__ == "__main__":
    main()
