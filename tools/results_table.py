#!/usr/bin/env python3
"""Prints the rows of BASELINE.md's results table from the end-of-round bench lines of a build
(profiles/<prefix>_*.json, written by tools/gpu_bench.sh end ...): filter-cycles/s, HBM GB/s = the line's PMC traffic over its
kernel time, roofline.achieved fraction, CPU oracle, largest error against the oracle replay.
usage: tools/results_table.py [prefix=profiles/r04_final]"""
import json
import sys

prefix = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_final"


def line(name):
    return json.loads(open(f"{prefix}_{name}.json").read().strip().splitlines()[-1])


def row(label, prec, gpus, name, err_note=None):
    d = line(name)
    r = d["roofline"]
    ms = r["kernel_ms_per_launch"]
    bw = f"{r['traffic'] / (ms * 1e-3) / 1e9:.0f}" if r.get("traffic") else "n/a"
    cb = d.get("cpu_baseline")
    cpu = f"{cb['value'] / 1e6:.2f} M ({cb['cores']} threads), {cb['single_core']['value'] / 1e6:.2f} M (1 core)" if cb else "—"
    p = d["parity"]
    err = err_note or f"{max(p['max_abs_mu'], p['max_abs_cov']):.1e}"
    print(f"| {label} | {prec} | {gpus} | {d['value'] / 1e6:.0f} M | {bw} | {r['frac']:.3f} | {cpu} | {err} |")


def f32_err(name):
    d = line(name)
    p, q = d["parity"], d["parity_recent"]
    fo = p.get("float_oracle_vs_fp64") or {}
    extra = f" (float oracle: {max(fo.get('max_abs_mu', 0), fo.get('max_abs_cov', 0)):.1e})" if fo else ""
    return f"{max(p['max_abs_mu'], p['max_abs_cov']):.1e} after all cycles{extra}; {max(q['max_abs_mu'], q['max_abs_cov']):.1e} over the last 16"


print(f"lib_sha16 {line('full')['lib_sha16']}")
row("2: 65 536 Pose", "fp64", "1", "cfg2")
row("3: 1 048 576 Pose (one GPU: all of it)", "fp32", "1", "f32", f32_err("f32"))
row("3: one GPU's share at N = 8 (131 072 Pose)", "fp32", "1 of 8", "cfg3", "as above")
row("metric: one GPU's share at N = 8 (131 072 Pose)", "fp64", "1 of 8", "shard8")
row("4: 4 194 304 Orient", "fp32", "1", "cfg4", f32_err("cfg4"))
row("3 with `wide_arithmetic` (fp32 arrays, fp64 arithmetic)", "fp32 / fp64", "1", "cfg3w")
row("4 with `wide_arithmetic`", "fp32 / fp64", "1", "cfg4w")
row("5: 262 144 Pose mixed", "fp64", "1", "cfg5")
row("metric: 1 048 576 Pose (headline, `python bench.py`)", "fp64", "1", "full")
row("the same, the driver's `--steps 20 --warmup 5`", "fp64", "1", "f64_20")
print("others:")
for n in ("full", "f32", "cfg4", "cfg4_64"):
    d = line(n)
    m = d.get("multi_cycle")
    print(f"  {n}: {d['value'] / 1e6:.0f} M per cycle" + (f", {m['filter_cycles_per_s_per_gpu'] / 1e6:.0f} M at {m['cycles_per_launch']} cycles per launch" if m else ""))
for n in ("f64_1000", "cfg5_order", "cfg2_single", "shard8_single", "group2", "uni262k"):
    print(f"  {n}: {line(n)['value'] / 1e6:.0f} M")
