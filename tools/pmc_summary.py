"""
pmc_summary.py - BUILD TOOLING: fold the counter_collection CSVs of two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; separate runs of the same bench.py command) into the per-kernel HBM
traffic summary that bench.py reads (profiles/rNN_*_pmc_hbm.json).

gfx950 correction (MI355X_MICROARCH.md, "HBM / rocprofv3"): FETCH_SIZE under-reports wide
coalesced reads by 2x (128-B requests tallied at 64 B); WRITE_SIZE is exact. Both are in KiB.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <units_per_dispatch> <note> > out.json
"""

import csv
import glob
import json
import os
import sys


def collect(directory, counter):
    sums, counts = {}, {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                sums[name] = sums.get(name, 0.0) + float(row["Counter_Value"])
                counts[name] = counts.get(name, 0) + 1
    return sums, counts


def main():
    fetch_dir, write_dir, units, note = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fs, fc = collect(fetch_dir, "FETCH_SIZE")
    ws, wc = collect(write_dir, "WRITE_SIZE")
    out = {"note": note, "units_per_dispatch": units, "kernels": {}}
    for name in sorted(set(fs) | set(ws)):
        if not name.startswith("qocx::"):
            continue
        f_avg = fs.get(name, 0.0) / max(fc.get(name, 0), 1)
        w_avg = ws.get(name, 0.0) / max(wc.get(name, 0), 1)
        out["kernels"][name] = {
            "FETCH_SIZE_KiB_avg": f_avg, "WRITE_SIZE_KiB_avg": w_avg,
            "dispatches": max(fc.get(name, 0), wc.get(name, 0)),
            "hbm_bytes_per_dispatch_corrected": (2.0 * f_avg + w_avg) * 1024.0,
        }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
