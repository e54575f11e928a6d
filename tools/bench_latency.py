"""
bench_latency.py - secondary measurement: single-seed evaluation latency (what one GRAPE
iteration of an ordinary qoc script costs), BASELINE.json configs[1]: n = 8 transmon, 500
steps, one seed; plus the C3 shape with one seed.

    python tools/bench_latency.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qoc_amd.engine import Engine, COST_TARGET_COHERENT  # noqa: E402
from tests import cases as cases_mod  # noqa: E402
from tests import gpu_helpers as gh  # noqa: E402


def main():
    """Variants: the column-chain sweep in one launch (impl 1, pipeline 1), the blocked sweep
    (impl 3, round 2's latency mode) and the two-sided pipeline of round 3 (4 / 8 time segments:
    forward and adjoint sweep of the ONE seed side by side) on the column-chain sweep (impl 1) and
    on the blocked sweep (impl 3)."""
    eng = Engine(0)
    for name in ("c2_transmon", "c3_subset"):
        case = cases_mod.case_by_name(name)
        gh.setup_engine(eng, case)
        u = gh.real_controls(case, case.controls[:1])
        # the last variant is what the single-control-set entry points get (knob "latency": the
        # inverse-image sweep of qocx_sweepi.hip, two-sided on four time segments)
        for impl, pipe, latency in ((1, 1, 0), (3, 1, 0), (1, 4, 0), (1, 8, 0), (3, 4, 0), (3, 8, 0),
                                    (3, 0, 1)):
            eng.set_knob("sweep_impl", impl)
            eng.set_knob("latency", latency)
            eng.set_pipeline(pipe)
            for _ in range(5):
                eng.evaluate(u, True)
            for want_grad in (False, True):
                t0 = time.perf_counter()
                reps = 50
                for _ in range(reps):
                    eng.evaluate(u, want_grad)
                wall = (time.perf_counter() - t0) / reps
                print(json.dumps(dict(case=name, n=case.n, steps=case.N - 1, want_grad=want_grad,
                                      sweep_impl=impl, time_segments=pipe, latency_mode=latency,
                                      ms_per_eval=round(wall * 1e3, 3),
                                      us_per_step=round(wall * 1e6 / (case.N - 1), 2))), flush=True)
        eng.set_knob("latency", 0)


if __name__ == "__main__":
    main()
