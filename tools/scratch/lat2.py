import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qoc_amd.engine import Engine, COST_TARGET_COHERENT
e = Engine(0)
n, steps = int(sys.argv[1]), int(sys.argv[2])
if n == 8:
    a = np.diag(np.sqrt(np.arange(1, n)), 1).astype(np.complex128); ad = a.conj().T
    h0 = 2 * np.pi * 0.05 * ad @ a + 0.5 * 2 * np.pi * (-0.2) * ad @ ad @ a @ a
    g = [a + ad, 1j * (a - ad)]
    psi0 = np.eye(n, dtype=np.complex128)[:1]; target = np.eye(n, dtype=np.complex128)[1:2]
else:
    bench.DIM = n
    h0, g, psi0, target = bench.make_problem()
e.set_schroedinger_problem(n, 1, bench.K_CTRL, steps + 1, steps + 1, bench.DT * steps, h0[None], np.stack(g)[None], psi0,
    costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
u = 0.1 * np.random.default_rng(77).standard_normal((1, steps + 1, bench.K_CTRL))
e.set_knob("latency", 1); e.set_knob("sweep_impl", 3)
ref = None
for um in (0, 1, 0, 1):
    e.set_knob("bidir_adj_first", um)
    for _ in range(5): out = e.evaluate(u, True)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): out = e.evaluate(u, True)
    wall = (time.perf_counter() - t0) / 300
    if ref is None: ref = out
    print(json.dumps(dict(bidir_adj_first=um, ms=round(wall * 1e3, 4))), "max diff", [float(np.max(np.abs(a - b))) for a, b in zip(ref, out)], "scale", float(np.max(np.abs(ref[1]))), flush=True)
