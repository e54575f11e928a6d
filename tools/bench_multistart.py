"""
bench_multistart.py - GPU-BOX TOOLING: wall time per iteration of grape_schroedinger_discrete_batch
on bench.py's C3 problem (256 seeds, 1001 x 2 real controls, Adam): the device-resident driver
against the host route (a subclass of Adam counts as "another plugin": one NumPy optimizer object
per seed; and the [B, P] host arrays when the device route is refused).

    python tools/bench_multistart.py > gpurun_out/r03_multistart.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import qoc_amd  # noqa: E402
from qoc_amd.standard import Adam, TargetStateInfidelity  # noqa: E402


def main():
    h0, g, psi0, target = bench.make_problem()
    ham = lambda u, t: h0 + u[0] * g[0] + u[1] * g[1]  # noqa: E731
    costs = [TargetStateInfidelity(target[:, :, None])]
    u0 = bench.make_controls(0, bench.SEEDS_PER_GPU)
    iterations = 12

    class PluginAdam(Adam):
        pass

    def conditions(controls):
        return controls
    variants = [("device resident", Adam(learning_rate=1e-2), None),
                ("host arrays [B, P] (a control condition keeps the loop on the host)",
                 Adam(learning_rate=1e-2), conditions),
                ("one plugin object per seed", PluginAdam(learning_rate=1e-2), None)]
    def run(opt, cond, count):
        t0 = time.perf_counter()
        res = qoc_amd.grape_schroedinger_discrete_batch(
            bench.K_CTRL, bench.N_EVAL, costs, bench.DT * (bench.N_EVAL - 1), ham,
            psi0[:, :, None], bench.N_EVAL, u0.copy(), iteration_count=count,
            log_iteration_step=0, optimizer=opt, impose_control_conditions=cond,
            max_control_norms=np.ones(bench.K_CTRL))
        return time.perf_counter() - t0, res
    for label, opt, cond in variants:
        run(opt, cond, 2)  # warm
        # set-up (probing the callable at every quadrature time, ...) cancels in the difference
        short, _ = run(opt, cond, iterations)
        long, res = run(opt, cond, 3 * iterations)
        print(json.dumps(dict(route=label, seeds=bench.SEEDS_PER_GPU, iterations=iterations,
                              ms_per_iteration=round((long - short) / (2 * iterations) * 1e3, 2),
                              setup_ms=round((short - (long - short) / 2) * 1e3, 1),
                              best_error=float(res.best.best_error))), flush=True)


if __name__ == "__main__":
    main()
