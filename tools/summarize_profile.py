#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory: per-kernel stats and PMC sums for ukf_kernel."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
summary = {}
print("== PMC (ukf_kernel dispatches only; mean per dispatch)")
for f in find("pmc_*/**/*counter_collection.csv"):
    acc, cnt = {}, {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "ukf_kernel" not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]
            acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"])
            cnt[k] = cnt.get(k, 0) + 1
    for k in sorted(acc):
        summary[k] = acc[k] / cnt[k]
        print(f"{k:32s} {acc[k] / cnt[k]:.6g}  (dispatches {cnt[k]})")
try:
    line = open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    print("== bench line under trace:", json.dumps({"value": d["value"], "roofline": d["roofline"]}))
    summary["_bench"] = d
except Exception as e:
    print("no bench line", e)
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
