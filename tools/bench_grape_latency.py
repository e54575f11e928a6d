"""
bench_grape_latency.py - GPU-BOX TOOLING: wall time per iteration of the reference-shaped entry point
grape_schroedinger_discrete on ONE control set (what a user's script pays per GRAPE iteration):
BASELINE configs[1] (n = 8 transmon, 500 steps) and the C3 shape (n = 32, 1000 steps).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qoc_amd  # noqa: E402
from qoc_amd.standard import Adam, TargetStateInfidelity  # noqa: E402
from tests import cases as cases_mod  # noqa: E402


def main():
    for name in ("c2_transmon", "c3_subset"):
        case = cases_mod.case_by_name(name)
        targets = case.cost_specs[0][1]["target_states"]
        times = {}
        for iters in (5, 20, 220):  # (the first call pays library load and context creation)
            t0 = time.perf_counter()
            qoc_amd.grape_schroedinger_discrete(
                case.K, case.Nc, [TargetStateInfidelity(targets)], case.T, case.hamiltonian(),
                case.initial_states, case.N, iteration_count=iters, log_iteration_step=0,
                optimizer=Adam(learning_rate=1e-3), initial_controls=case.controls[0])
            times[iters] = time.perf_counter() - t0
        per_iter = (times[220] - times[20]) / 200
        print(json.dumps(dict(case=name, n=case.n, steps=case.N - 1,
                              ms_per_iteration=round(per_iter * 1e3, 3),
                              setup_ms=round((times[20] - 20 * per_iter) * 1e3, 1))), flush=True)


if __name__ == "__main__":
    main()
