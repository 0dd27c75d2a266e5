#!/usr/bin/env python3
"""Turns the passes of tools/pmc_configs.sh into profiles/traffic_latest.json (HBM bytes per launch per config) and
profiles/pmc_latest.json (VALU wave-instructions per wave, issue cycles per instruction, sustained clock)."""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]


def rows(d):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            yield row


def mean_counter(d, counter, kernel_substr):
    vals = [float(r["Counter_Value"]) for r in rows(d) if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
    return sum(vals) / len(vals) if vals else None


def kernel_ns(d, kernel_substr):
    """mean duration of the kernel's dispatches from the pass's own kernel trace"""
    vals = []
    for f in glob.glob(os.path.join(out, d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                vals.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return sum(vals) / len(vals) if vals else None


# DPP-fused fp32 arithmetic on the default path of each kernel (static listing, tools/isa_phases.py ):
# the SQ counts them as FMA / ADD, the SIMD issues them at the 4-cycle rate
DPP_FUSED_F32 = {"ukf_kernel16<f32,pose,cycle>": 243.0, "ukf_kernel16<f32,orient,cycle>": 300.0,
                 # (the plain-launch instantiations run the same factorisations)
                 "ukf_kernel16<f32,pose,cycle-plain>": 243.0, "ukf_kernel16<f32,orient,cycle-plain>": 300.0}

calib = {}
for prec in ("f64", "f32"):
    known = json.loads(open(os.path.join(out, f"calib_{prec}_FETCH_SIZE.json")).read().strip().splitlines()[-1])["bytes_per_launch"]
    fc = mean_counter(f"calib_{prec}_FETCH_SIZE", "FETCH_SIZE", "read_rows") * 1024.0
    wc = mean_counter(f"calib_{prec}_WRITE_SIZE", "WRITE_SIZE", "write_rows") * 1024.0
    calib[prec] = {"known_bytes": known, "FETCH_SIZE_bytes": fc, "WRITE_SIZE_bytes": wc, "k_read": known / fc, "k_write": known / wc}

traffic, pmc, lines = [], [], []
for name in ("headline_f64", "headline_f32", "cfg2", "cfg3", "cfg4", "cfg5", "multi8_f64", "multi8_f32", "wide_f32", "cfg4_wide"):
    try:
        bench = json.loads(open(os.path.join(out, f"{name}_FETCH_SIZE.json")).read().strip().splitlines()[-1])
    except Exception as e:
        lines.append(f"{name}: no bench line ({e})")
        continue
    prec = bench["config"].get("hbm_format", bench["dtype"])   # HBM format (wide arithmetic: fp32 arrays, dtype f64)
    arith = bench["dtype"]                                      # what the instructions are
    kern = bench["roofline"]["kernel"]
    n = bench["config"]["filters_per_gpu"]
    fetch = mean_counter(f"{name}_FETCH_SIZE", "FETCH_SIZE", "ukf_kernel") * 1024.0
    write = mean_counter(f"{name}_WRITE_SIZE", "WRITE_SIZE", "ukf_kernel") * 1024.0
    c = calib[prec]
    hbm = fetch * c["k_read"] + write * c["k_write"]
    cpl = float(bench["roofline"].get("cycles_per_launch", 1.0))   # multi-cycle launches: counters are per launch of cpl cycles
    ident = {"lib_sha16": bench.get("lib_sha16"), "build_head": bench.get("build_head")}   # the binary the counters belong to
    traffic.append({"config": name, **ident, "kernel": kern, "filters_per_launch": n, "cycles_per_launch": cpl, "precision": prec, "calibration": c,
                    "raw": {"FETCH_SIZE_bytes": fetch, "WRITE_SIZE_bytes": write},
                    "hbm_read_bytes_per_launch": fetch * c["k_read"], "hbm_write_bytes_per_launch": write * c["k_write"],
                    "hbm_bytes_per_launch": hbm,
                    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]})
    d = f"{name}_SQ_WAVES"
    waves = mean_counter(d, "SQ_WAVES", "ukf_kernel")
    valu = mean_counter(d, "SQ_INSTS_VALU", "ukf_kernel")
    act = mean_counter(d, "SQ_ACTIVE_INST_VALU", "ukf_kernel")
    lds = mean_counter(d, "SQ_INSTS_LDS", "ukf_kernel")
    wcyc = mean_counter(d, "SQ_WAVE_CYCLES", "ukf_kernel")
    grbm = mean_counter(d, "GRBM_GUI_ACTIVE", "ukf_kernel")
    ns = kernel_ns(d, "ukf_kernel")
    clock_mhz = grbm / 8.0 / (ns * 1e-9) / 1e6
    # instruction classes (two more passes) -> issue cycles per wave with the per-class costs measured by tools/valu_tput.hip
    # and tools/gen_valu_tput2.py on this chip (profiles/r02_valu_throughput_*.txt): a wave64 instruction occupies the SIMD for
    # 4 cycles when it is fp64, DPP, 64-bit, a select / compare / shift / 3-operand integer or reads an SGPR; 2 cycles when
    # it is a plain fp32 FMA / MUL / ADD or v_mov_b32; ~2.7 for and / or / add_u32; 6.5 (fp32) / 13 (fp64) for rcp / rsq / sqrt
    cls = {}
    for cname in ("FMA_F32", "MUL_F32", "ADD_F32", "TRANS_F32", "INT32", "CVT", "FMA_F64", "MUL_F64", "ADD_F64", "TRANS_F64", "INT64"):
        dd = f"{name}_SQ_INSTS_VALU_FMA_F32" if cname in ("FMA_F32", "MUL_F32", "ADD_F32", "TRANS_F32", "INT32", "CVT") else f"{name}_SQ_INSTS_VALU_FMA_F64"
        v = mean_counter(dd, "SQ_INSTS_VALU_" + cname, "ukf_kernel")
        cls[cname] = (v / waves) if v is not None else None
    weighted = None
    if all(v is not None for v in cls.values()):
        per_wave = valu / waves
        fp32 = cls["FMA_F32"] + cls["MUL_F32"] + cls["ADD_F32"]
        fp64 = cls["FMA_F64"] + cls["MUL_F64"] + cls["ADD_F64"]
        other = per_wave - fp32 - fp64 - cls["TRANS_F32"] - cls["TRANS_F64"] - cls["INT32"] - cls["INT64"] - cls["CVT"]
        dpp_fused = DPP_FUSED_F32.get(kern, 0.0) if arith == "f32" else 0.0    # counted as FMA / ADD by the SQ, issue like fp64
        weighted = (2.0 * fp32 + 2.0 * dpp_fused + 4.0 * fp64 + 6.5 * cls["TRANS_F32"] + 13.0 * cls["TRANS_F64"] + 3.3 * cls["INT32"]
                    + 4.0 * cls["INT64"] + 4.0 * cls["CVT"] + 3.5 * other)
        cls["other(mov,select,compare,dpp mov)"] = other
        cls["dpp_fused_fp32_static"] = dpp_fused
    e = {"config": name, **ident, "kernel": kern, "filters_per_launch": n, "cycles_per_launch": cpl, "precision": prec,
         "valu_classes_per_wave": cls, "issue_cycles_per_wave_weighted": weighted,
         "valu_issue_frac_weighted_in_pass": (weighted * waves / (1024 * clock_mhz * 1e6 * ns * 1e-9)) if weighted else None,
         "valu_insts_per_wave": valu / waves, "lds_insts_per_wave": lds / waves,
         # issue cost of one wave64 VALU instruction on a SIMD (MI355X_MICROARCH.md, cycle constants): fp64 moves 16 lanes
         # per cycle = 4 cycles; fp32 sustains one instruction per 2 cycles when several waves interleave
         "issue_cycles_per_valu_inst": 4.0 if arith == "f64" else 2.0,
         "sq_active_quadcycles_per_inst": act / valu,             # per-wave activity (overlaps between waves of a SIMD)
         "valu_active_cycles_per_wave": 4.0 * act / waves, "wave_cycles_per_wave": 4.0 * wcyc / waves,
         "clock_mhz": clock_mhz, "kernel_ms_in_pass": ns * 1e-6,
         "valu_issue_frac_in_pass": (valu * (4.0 if arith == "f64" else 2.0)) / (1024 * clock_mhz * 1e6 * ns * 1e-9),
         "source": f"rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... GRBM_GUI_ACTIVE, tools/pmc_configs.sh ({tag})"}
    pmc.append(e)
    lines.append(f"{name:13s} {kern:34s} n={n:8d} traffic {hbm / 1e9:7.3f} GB/launch (alg {bench['roofline']['algorithmic_bytes_per_launch'] / 1e9:.3f}) "
                 f"VALU/wave {e['valu_insts_per_wave']:.0f} LDS/wave {e['lds_insts_per_wave']:.0f} cyc/inst {e['issue_cycles_per_valu_inst']:.0f} "
                 f"clock {clock_mhz:.0f} MHz kernel {ns * 1e-6:.3f} ms VALU issue frac {e['valu_issue_frac_in_pass']:.3f}"
                 + (f" weighted {e['valu_issue_frac_weighted_in_pass']:.3f} ({weighted:.0f} cycles/wave)" if weighted else ""))
json.dump({"note": "HBM bytes per launch of the fused kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), corrected with "
                   "factors calibrated on the engine's own access pattern (tools/calib_traffic.hip, tools/pmc_configs.sh)",
           "entries": traffic}, open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
json.dump({"note": "per-wave instruction counters and sustained clock of the fused kernel, one SQ/GRBM pass per BASELINE configuration "
                   "(tools/pmc_configs.sh); bench.py turns them into roofline.valu with the kernel time it measures live",
           "entries": pmc}, open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
for t in ("default", "f32", "cfg2"):
    for f in glob.glob(os.path.join(out, f"trace_{t}", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, f"{tag}_{t}_kernel_stats.csv"))
        for row in csv.DictReader(open(f)):
            if "ukf_kernel" in row["Name"]:
                lines.append(f"trace_{t}: {row['Name'][:60]} calls {row['Calls']} avg {float(row['AverageNs']) / 1e6:.4f} ms "
                             f"min {float(row['MinNs']) / 1e6:.4f} max {float(row['MaxNs']) / 1e6:.4f}")
    try:
        b = json.loads(open(os.path.join(out, f"trace_{t}.json")).read().strip().splitlines()[-1])
        lines.append(f"trace_{t}: bench under trace value {b['value'] / 1e6:.1f} M/s kernel_ms {b['roofline']['kernel_ms_per_launch']:.4f}")
        json.dump(b, open(os.path.join(out, f"{tag}_{t}_bench_under_rocprof.json"), "w"))
    except Exception as e:
        lines.append(f"trace_{t}: {e}")
print("\n".join(lines))
