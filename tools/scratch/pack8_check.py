import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import cases as cases_mod, gpu_helpers as gh
from qoc_amd.engine import Engine
engine = Engine(0)
case = cases_mod.case_random("pack8_n8", n=8, N=41, seeds=3, h_seed=7708, S=1, K=1, Nc=41, dt=0.05, sigma=0.6)
case.h0 = case.h0 * 0.1
gh.setup_engine(engine, case)
swing = case.controls[0] * np.where((np.arange(case.controls[0].shape[0]) // 2) % 2 == 0, 0.01, 2.5)[:, None]
u = np.concatenate([case.controls, swing[None]])
engine.set_knob("pack8", 0); ref = engine.evaluate(u, True)
engine.set_knob("pack8", 1); out = engine.evaluate(u, True)
for a, b in zip(ref, out):
    print([bool(np.array_equal(a[i], b[i])) for i in range(len(a))], float(np.max(np.abs(a - b))))
