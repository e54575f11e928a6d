import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import cases as cases_mod, gpu_helpers as gh
from qoc_amd.engine import Engine
engine = Engine(0)
for n, dt, scale in [(32, 0.05, 1.0), (20, 0.4, 6.0), (8, 0.05, 1.0)]:
    case = cases_mod.case_random("sweep1_n%d" % n, n=n, N=41, seeds=3, h_seed=7300 + n, S=1, K=2, Nc=17, dt=dt, sigma=0.6)
    case.h0 = case.h0 * scale
    engine.set_knob("sweep_inverse_small", 0)
    gh.setup_engine(engine, case)
    u = np.concatenate([case.controls, -0.7 * case.controls])
    for unit in (1, 0):
        engine.set_knob("unit_adjoint", unit)
        for pipe in (1, 4):
            engine.set_pipeline(pipe)
            engine.set_knob("sweep_one", 0)
            g = engine.evaluate(u, True)
            engine.set_knob("sweep_one", 1)
            o = engine.evaluate(u, True)
            o2 = engine.evaluate(u, True)
            print(n, "unit", unit, "pipe", pipe, [float(np.max(np.abs(a - b))) for a, b in zip(g, o)], "repeat", [float(np.max(np.abs(a - b))) for a, b in zip(o, o2)], "gmax", float(np.max(np.abs(g[1]))))
engine.close()
