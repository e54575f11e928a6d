"""
pmc_counters.py - BUILD TOOLING: per-kernel averages of every counter found in the
counter_collection CSVs of one or more rocprofv3 --pmc passes (one directory per pass).

    python tools/pmc_counters.py <pass_dir> [<pass_dir> ...] > profiles/rNN_xx_pmc_sq.json
"""

import csv
import glob
import json
import os
import sys


def main():
    out = {}
    for directory in sys.argv[1:]:
        for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                    if not name.startswith("qocx::"):
                        continue
                    slot = out.setdefault(name, {}).setdefault(row["Counter_Name"], [0.0, 0])
                    slot[0] += float(row["Counter_Value"])
                    slot[1] += 1
    summary = {k: {c: v[0] / max(v[1], 1) for c, v in sorted(cs.items())} | {"dispatches": max(v[1] for v in cs.values())}
               for k, cs in sorted(out.items())}
    json.dump(summary, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
