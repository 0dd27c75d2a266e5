#!/usr/bin/env python3
"""Static per-phase instruction accounting of the tuned kernel (no GPU needed).

Compiles one launch TU with -DUKFB_PHASE_MARKS (assembly comments at the phase boundaries of
ukf_kernel16.hpp), then walks the chosen kernel's assembly in layout order and counts instructions per phase
and per class (VALU / fp64-transcendental / DPP / cndmask / LDS / SALU / VMEM).  Blocks the compiler laid out
after `s_endpgm`-less cold jumps are attributed to the phase whose marker precedes them in the text; loops
are counted once (the covariance loop prints its trip count in the source).  Use for RELATIVE accounting;
the dynamic totals come from tools/phase_pmc.sh (SQ_INSTS_VALU).

usage: tools/isa_phases.py [pose|orient] [f64|f32] [cycle|predict|update|multi|indirect|plain|multi-plain|streams|bucketed-streams|
                            predict-plain|update-plain|update-streams] [extra hipcc flags...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        if "cndmask" in op:
            return "cndmask"
        if op.endswith("_dpp") or "_dpp" in op:
            return "dpp"
        if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div")):
            return "trans64"
        if op.startswith(("v_mov", "v_accvgpr")):
            return "mov"
        if op.startswith(("v_cmp", "v_and", "v_or", "v_xor", "v_lsh", "v_add_u", "v_sub_u", "v_mad_u", "v_mul_u", "v_min_i",
                          "v_max_i", "v_add3", "v_bfe", "v_mul_lo", "v_mul_hi", "v_add_co", "v_addc", "v_ashr", "v_mbcnt",
                          "v_readlane", "v_readfirstlane", "v_writelane", "v_sub_co", "v_subb", "v_not")):
            return "int"
        return "fp"
    return "other"


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "pose"
    prec = sys.argv[2] if len(sys.argv) > 2 else "f64"
    mode = sys.argv[3] if len(sys.argv) > 3 else "cycle"
    extra = sys.argv[4:]
    tu = os.path.join(ROOT, "slam-pose_estimation_amd", "csrc", f"ukf_launch_{model}_{prec}.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize",
                               "-mllvm", "-disable-machine-licm", "-DUKFB_PHASE_MARKS", "-S", "--cuda-device-only", "-o", out, tu] + extra,
                              stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    flags = {"cycle": "Lb1ELb1ELb0ELb0ELi0E", "predict": "Lb1ELb0ELb0ELb0ELi0E", "update": "Lb0ELb1ELb0ELb0ELi0E", "multi": "Lb1ELb1ELb1ELb0ELi0E",
             "indirect": "Lb1ELb1ELb0ELb1ELi0E", "plain": "Lb1ELb1ELb0ELb0ELi2E", "multi-plain": "Lb1ELb1ELb1ELb0ELi2E",
             "streams": "Lb1ELb1ELb0ELb0ELi1E", "bucketed-streams": "Lb1ELb1ELb0ELb1ELi1E", "predict-plain": "Lb1ELb0ELb0ELb0ELi2E",
             "update-plain": "Lb0ELb1ELb0ELb0ELi2E", "update-streams": "Lb0ELb1ELb0ELb0ELi1E"}[mode]
    start = None
    for i, line in enumerate(text):
        if line.startswith("_ZN4ukfb12ukf_kernel16") and flags in line and line.rstrip().endswith(":") is False and ":" in line:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    phases = collections.OrderedDict()
    cur = "entry"
    phases[cur] = collections.Counter()
    ops = collections.Counter()
    for line in text[start + 1:]:
        s = line.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".section"):
            break
        m = re.match(r"; @@PHASE (\S+)", s)
        if m:
            cur = m.group(1)
            phases.setdefault(cur, collections.Counter())
            continue
        if not s or s.startswith((";", ".", "_Z")) or s.endswith(":"):
            continue
        op = s.split()[0]
        phases[cur][classify(op)] += 1
        ops[(cur, op)] += 1
    cols = ["fp", "trans64", "dpp", "cndmask", "mov", "int", "lds", "vmem", "salu"]
    print(f"{'phase':18s}" + "".join(f"{c:>9s}" for c in cols) + f"{'VALU':>9s}")
    tot = collections.Counter()
    for ph, c in phases.items():
        valu = sum(c[k] for k in ("fp", "trans64", "dpp", "cndmask", "mov", "int"))
        print(f"{ph:18s}" + "".join(f"{c[k]:9d}" for k in cols) + f"{valu:9d}")
        tot.update(c)
    valu = sum(tot[k] for k in ("fp", "trans64", "dpp", "cndmask", "mov", "int"))
    print(f"{'TOTAL (static)':18s}" + "".join(f"{tot[k]:9d}" for k in cols) + f"{valu:9d}")
    if os.environ.get("ISA_OPS"):
        ph = os.environ["ISA_OPS"]
        for (p, op), n in sorted(ops.items(), key=lambda kv: -kv[1]):
            if p == ph:
                print(f"   {op:28s} {n}")


if __name__ == "__main__":
    main()
