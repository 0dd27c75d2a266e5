#!/usr/bin/env python3
"""Turns the PMC passes of tools/traffic.sh into the `roofline.traffic` figure: HBM bytes per launch of the
fused kernel = FETCH_SIZE * k_read + WRITE_SIZE * k_write, with k_* = known bytes / counted bytes measured
on the calibration kernels that replay the engine's own access pattern (counters are in KiB)."""
import csv
import glob
import json
import os
import sys

out, prec, n = sys.argv[1], sys.argv[2], int(sys.argv[3])


def mean_counter(d, counter, kernel_substr):
    vals = []
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


known = json.loads(open(os.path.join(out, "calib_FETCH_SIZE.json")).read().strip().splitlines()[-1])["bytes_per_launch"]
fetch_cal = mean_counter("calib_FETCH_SIZE", "FETCH_SIZE", "read_rows") * 1024.0
write_cal = mean_counter("calib_WRITE_SIZE", "WRITE_SIZE", "write_rows") * 1024.0
k_read, k_write = known / fetch_cal, known / write_cal
fetch = mean_counter("bench_FETCH_SIZE", "FETCH_SIZE", "ukf_kernel") * 1024.0
write = mean_counter("bench_WRITE_SIZE", "WRITE_SIZE", "ukf_kernel") * 1024.0
bench = json.loads(open(os.path.join(out, "bench_FETCH_SIZE.json")).read().strip().splitlines()[-1])
tsize = 8 if prec == "f64" else 4
res = {
    "kernel": bench["roofline"]["kernel"], "filters_per_launch": n, "precision": prec,
    "calibration": {"known_bytes": known, "FETCH_SIZE_bytes": fetch_cal, "WRITE_SIZE_bytes": write_cal,
                    "k_read": k_read, "k_write": k_write},
    "raw": {"FETCH_SIZE_bytes": fetch, "WRITE_SIZE_bytes": write},
    "hbm_read_bytes_per_launch": fetch * k_read, "hbm_write_bytes_per_launch": write * k_write,
    "hbm_bytes_per_launch": fetch * k_read + write * k_write,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "moved_layout_bytes_per_launch": n * (2 * 91 + 15) * tsize,
}
print(json.dumps(res, indent=1))
