"""
bench_sizes.py - GPU-BOX TOOLING: forward + gradient throughput of the Schroedinger engine against
the Hilbert size (the GUE problem of bench.py, 256 seeds x 1000 steps, M2, S = 1).

    python tools/bench_sizes.py 8 16 32 48 64
One JSON line per size: ms per evaluation, M propagator-steps/s, per-kernel ms per launch.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 48, 64]
    for dim in sizes:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_ab.py"), "--rounds", "8",
                              "--evals", "4", "--dim", str(dim)], capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(json.dumps(dict(dim=dim, error=out.stderr[-300:])), flush=True)
            continue
        d = json.loads(line[-1])
        print(json.dumps(dict(dim=dim, seeds=d["seeds"], ms_per_evaluation=d["ms_median"],
                              msteps_per_s=d["seeds"] * 1000 / d["ms_median"] / 1e3,
                              kernel_ms_per_launch=d["kernel_ms_per_launch"])), flush=True)


if __name__ == "__main__":
    main()
