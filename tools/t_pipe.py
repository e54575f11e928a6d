import time, sys
sys.path.insert(0, '.')
import numpy as np
t=time.time()
from qoc_amd.engine import Engine
e=Engine(0); print("create1", time.time()-t)
t=time.time(); e2=Engine(0); print("create2", time.time()-t); e2.close()
import bench
h0,g,psi0,target=bench.make_problem()
from qoc_amd.engine import COST_TARGET_COHERENT
e.set_schroedinger_problem(32,1,2,1001,1001,0.05*1000,h0[None],np.stack(g)[None],psi0,costs=[dict(kind=COST_TARGET_COHERENT,step_cost=0,scale=1.0,vectors=target)])
u=bench.make_controls(0,256)
e.upload_controls(u)
for pipe in (1,2,4,8):
    e.set_pipeline(pipe)
    e.eval_resident(True)
    t=time.time()
    for _ in range(3): e.eval_resident(True)
    print("pipe",pipe,(time.time()-t)/3*1e3,"ms")
e.set_timing(True)
for pipe in (1,8):
    e.set_pipeline(pipe); e.reset_timing()
    t=time.time()
    for _ in range(3): e.eval_resident(True)
    print("timed pipe",pipe,(time.time()-t)/3*1e3,"ms", e.timing())
