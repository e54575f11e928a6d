import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
NAMES = {0: "K1a", 1: "sweep", 2: "K3", 3: "scatter", 4: "K1b", 5: "lindblad"}
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
h0, g, _, _ = bench.make_problem()
rng = np.random.default_rng(5)
q, _ = np.linalg.qr(rng.standard_normal((bench.DIM, bench.DIM)) + 1j * rng.standard_normal((bench.DIM, bench.DIM)))
u = bench.make_controls(0, 256)
eng = Engine(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); eng.set_knob(k, int(v))
eng.set_timing(True)
psi0 = np.eye(bench.DIM, dtype=np.complex128)[:S]
target = np.ascontiguousarray(q.T[:S])
eng.set_schroedinger_problem(bench.DIM, S, bench.K_CTRL, bench.N_EVAL, bench.N_EVAL, bench.DT * (bench.N_EVAL - 1),
    h0[None], np.stack(g)[None], psi0, costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
eng.upload_controls(u)
for _ in range(3): eng.eval_resident(True)
t0 = time.perf_counter()
for _ in range(3): eng.eval_resident(True)
print("ms_per_eval", (time.perf_counter() - t0) / 3 * 1e3)
rows = sorted([(a, b, NAMES[int(w)]) for w, a, b in eng.timeline()])
for a, b, n in rows: print("{:8s} {:7.3f} -> {:7.3f}  ({:.3f})".format(n, a, b, b - a))
