#!/bin/bash
# Runs ON THE GPU BOX: per-kernel (predict-only / update-only / fused) LDS and VALU counters.
out=$PWD/gpurun_out/phase_pmc; mkdir -p $out; export TMPDIR=/tmp
python3 tools/phase_split.py > $out/times.txt 2>&1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -- python3 tools/phase_split.py > /dev/null 2> $out/pmc_$name.err
done
python3 - <<'PY' > $out/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/phase_pmc/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "ukf_kernel" not in k: continue
        k = k.replace("void ukfb::", "").replace("ukfb::", "").split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    w = acc[k]["SQ_WAVES"] / max(1, cnt[k]["SQ_WAVES"])
    for c in sorted(acc[k]):
        v = acc[k][c] / cnt[k][c]
        print(f"   {c:28s} {v:14.6g}   per wave {v / w:10.1f}")
PY
cat $out/times.txt $out/summary.txt
