#!/usr/bin/env python3
"""Approximate VGPR liveness per phase of the tuned kernel from its assembly (no GPU needed).

Backward scan over the kernel's instructions in layout order, treating the code as straight-line (loops and the few
wave-uniform branches make this an approximation: a value defined before a loop and used inside it is live throughout,
which a single backward pass over the layout order sees as long as the use follows the definition in the text).  Prints the
maximum number of simultaneously live VGPRs inside each phase (UKFB_PHASE_MARKS) -- where the register budget is spent.

usage: tools/isa_liveness.py [pose|orient] [f64|f32] [cycle|predict|update|multi] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


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
    flags = {"cycle": "Lb1ELb1ELb0ELb0E", "predict": "Lb1ELb0ELb0ELb0E", "update": "Lb0ELb1ELb0ELb0E", "multi": "Lb1ELb1ELb1ELb0E"}[mode]
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN4ukfb12ukf_kernel16") and flags in l and ":" in l)
    ins = []   # (phase, defs, uses)
    cur = "entry"
    for line in text[start + 1:]:
        s = line.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"; @@PHASE (\S+)", s)
        if m:
            cur = m.group(1)
            continue
        if not s or s.startswith((";", ".", "_Z")) or s.endswith(":"):
            continue
        s = s.split(";")[0]
        op, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        if not ops or not ops[0]:
            continue
        stores = op.startswith(("ds_write", "ds_store", "global_store", "scratch_store", "buffer_store", "flat_store")) or \
            op.startswith(("v_cmp", "s_")) and not op.startswith("v_cmpx")
        if stores:
            d, u = set(), set().union(*[regs(o) for o in ops])
        else:
            d = regs(ops[0])
            u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if op.startswith(("v_fmac", "v_mac", "v_pk_fmac")) or "dpp" in op or op.startswith("v_cndmask"):
                u |= d          # destructive forms read their destination (a DPP move keeps lanes it does not write)
        ins.append((cur, d, u))
    live = set()
    peak = {}
    for ph, d, u in reversed(ins):
        live -= d
        live |= u
        peak[ph] = max(peak.get(ph, 0), len(live))
    order = []
    for ph, _, _ in ins:
        if ph not in order:
            order.append(ph)
    for ph in order:
        print(f"{ph:18s} max live VGPRs ~{peak[ph]}")


if __name__ == "__main__":
    main()
