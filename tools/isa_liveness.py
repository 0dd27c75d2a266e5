#!/usr/bin/env python3
"""VGPR liveness per phase of the tuned kernel from its assembly (no GPU needed).

Builds the control-flow graph of the kernel's instructions (labels, s_branch / s_cbranch_*), runs the backward liveness
dataflow to its fixpoint and prints the maximum number of simultaneously live VGPRs inside each phase (UKFB_PHASE_MARKS) --
where the register budget is spent -- and, for a multi-cycle kernel, what is live at the head of the cycle loop.  Exec-masked
writes are treated as full definitions (an under-estimate where a divergent region writes part of a register).

usage: tools/isa_liveness.py [pose|orient] [f64|f32] [cycle|predict|update|multi|indirect|plain|multi-plain] [extra hipcc flags...]"""
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
    flags = {"cycle": "Lb1ELb1ELb0ELb0ELb0E", "predict": "Lb1ELb0ELb0ELb0ELb0E", "update": "Lb0ELb1ELb0ELb0ELb0E", "multi": "Lb1ELb1ELb1ELb0ELb0E",
             "indirect": "Lb1ELb1ELb0ELb1ELb0E", "plain": "Lb1ELb1ELb0ELb0ELb1E", "multi-plain": "Lb1ELb1ELb1ELb0ELb1E"}[mode]
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN4ukfb12ukf_kernel16") and flags in l and ":" in l)
    ins = []   # (phase, defs, uses)
    cur = "entry"
    labels, back_edges, pending = {}, [], []   # label -> index of the next instruction; (branch index, target index) of backward branches; all branches
    for line in text[start + 1:]:
        s = line.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"; @@PHASE (\S+)", s)
        if m:
            cur = m.group(1)
            continue
        if s.startswith(".LBB") and s.split(";")[0].strip().endswith(":"):
            labels[s.split(":")[0]] = len(ins)
            continue
        if not s or s.startswith((";", ".", "_Z")) or s.endswith(":"):
            continue
        s = s.split(";")[0]
        op, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        if not ops or not ops[0]:
            continue
        if op.startswith(("s_cbranch", "s_branch")):
            pending.append((len(ins), ops[0], op.startswith("s_branch")))
            if ops[0] in labels:
                back_edges.append((len(ins), labels[ops[0]]))
        if op == "s_endpgm":
            pending.append((len(ins), None, True))
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
    branch_at = {idx: ((labels.get(lbl) if lbl is not None else None), uncond) for idx, lbl, uncond in pending}
    # proper liveness over the control-flow graph: basic blocks end at branches and begin at labels; a conditional branch
    # falls through and jumps, s_branch only jumps; backward dataflow to the fixpoint, then the per-instruction live sets
    n = len(ins)
    leaders = {0} | set(labels.values()) | {b + 1 for b in branch_at}
    leaders = sorted(x for x in leaders if x < n)
    block_of = {}
    blocks = []
    for bi, st in enumerate(leaders):
        en = leaders[bi + 1] if bi + 1 < len(leaders) else n
        blocks.append((st, en))
        for k in range(st, en):
            block_of[k] = bi
    succ = [[] for _ in blocks]
    for bi, (st, en) in enumerate(blocks):
        last = en - 1
        if last in branch_at:
            tgt, uncond = branch_at[last]
            if tgt is not None and tgt < n:
                succ[bi].append(block_of[tgt])
            if not uncond and en < n:
                succ[bi].append(block_of[en])
        elif en < n:
            succ[bi].append(block_of[en])
    live_in = [set() for _ in blocks]
    changed = True
    while changed:
        changed = False
        for bi in range(len(blocks) - 1, -1, -1):
            st, en = blocks[bi]
            live = set()
            for sb in succ[bi]:
                live |= live_in[sb]
            for k in range(en - 1, st - 1, -1):
                live = (live - ins[k][1]) | ins[k][2]
            if live != live_in[bi]:
                live_in[bi] = live
                changed = True
    peak = {}
    where = {}
    for bi, (st, en) in enumerate(blocks):
        live = set()
        for sb in succ[bi]:
            live |= live_in[sb]
        for k in range(en - 1, st - 1, -1):
            live = (live - ins[k][1]) | ins[k][2]
            ph = ins[k][0]
            if len(live) > peak.get(ph, 0):
                peak[ph] = len(live)
    loop = max(back_edges, key=lambda e: e[0] - e[1]) if back_edges else None
    if loop and loop[0] - loop[1] >= 1000:
        head = live_in[block_of[loop[1]]]
        print(f"cycle loop: instructions {loop[1]}..{loop[0]}, {len(head)} VGPRs live at its head: " + " ".join(f"v{r}" for r in sorted(head)))
    order = []
    for ph, _, _ in ins:
        if ph not in order:
            order.append(ph)
    for ph in order:
        print(f"{ph:18s} max live VGPRs ~{peak[ph]}")


if __name__ == "__main__":
    main()
