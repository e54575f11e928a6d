"""print the interesting fields of a bench.py JSON line read from stdin (build tooling)"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 2), round(d["value"] / 1e6, 2),
      {k: round(v, 3) for k, v in d["kernel_ms_per_launch"].items() if v},
      round(d["roofline"]["frac"], 3), d["check"]["grad_l2"])
