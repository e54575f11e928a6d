import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, json
from qoc_amd.engine import Engine
e = Engine(0)
print(json.dumps(bench.latency_secondary(e, reps=50)))
