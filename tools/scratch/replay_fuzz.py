import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import fuzz_lindblad as fl
from qoc_amd.engine import Engine
eng = Engine(0)
seed = int(sys.argv[1]); count = int(sys.argv[2])
rng = np.random.default_rng(seed)
for index in range(count):
    state = rng.bit_generator.state
    worst, tag = (fl.one_two_sided if index % 2 else fl.one)(eng, rng, index)
    if worst > 1.0:
        print("index", index, worst, tag)
        for knobs in ({}, {"lindblad_chain": 0}, {"lindblad_hermitian": 0}, {"lindblad_two_sided": 0}, {"lindblad_real_ops": 0}):
            r2 = np.random.default_rng(); r2.bit_generator.state = state
            for k, v in knobs.items(): eng.set_knob(k, v)
            w2, _ = (fl.one_two_sided if index % 2 else fl.one)(eng, r2, index)
            for k in knobs: eng.set_knob(k, 1)
            print("   knobs", knobs, "->", w2)
        # repeat default a few times (race?)
        for rep in range(3):
            r2 = np.random.default_rng(); r2.bit_generator.state = state
            w2, _ = (fl.one_two_sided if index % 2 else fl.one)(eng, r2, index)
            print("   repeat ->", w2)
