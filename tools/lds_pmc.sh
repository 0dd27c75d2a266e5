#!/bin/bash
# Runs ON THE GPU BOX: LDS counters of the headline kernels (bank conflicts, address conflicts, LDS busy) per wavefront.
# usage: tools/lds_pmc.sh [tag]   -> gpurun_out/lds_pmc/<tag>_summary.txt
set -u
tag=${1:-r03}
out=$PWD/gpurun_out/lds_pmc; mkdir -p $out
export TMPDIR=/tmp
B="--steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extra-regions"
for prec in f64 f32; do
  for pass in "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
    p=$(echo $pass | cut -d' ' -f2)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/${prec}_$p -- python3 bench.py $B --precision $prec > $out/${prec}_$p.json 2> $out/${prec}_$p.err
    echo "$prec $p rc=$?"
  done
done
python3 - $out $tag <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
lines = []
for prec in ("f64", "f32"):
    tot = collections.Counter(); n = collections.Counter()
    for f in glob.glob(os.path.join(out, prec + "_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ukf_kernel16" not in r.get("Kernel_Name", ""):
                continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    waves = tot["SQ_WAVES"] / max(1, n["SQ_WAVES"]) * 1.0
    lines.append(f"{prec}: per wavefront (counter sum over the kernel's dispatches / SQ_WAVES of the same pass)")
    for k in sorted(tot):
        if k == "SQ_WAVES":
            continue
        per_disp = tot[k] / n[k]
        lines.append(f"   {k:28s} {per_disp / waves:12.1f}")
open(os.path.join(out, tag + "_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
