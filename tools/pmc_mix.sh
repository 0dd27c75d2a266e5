#!/bin/bash
# Runs ON THE GPU BOX: dynamic VALU instruction mix of the headline kernel (fp64 and fp32): SQ_INSTS_VALU_* classes per wavefront.
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_mix; rm -rf $out; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*\|SQ_ACTIVE_INST[A-Z0-9_]*\|SQ_INSTS_[A-Z0-9_]*\|SQ_THREAD_CYCLES[A-Z0-9_]*\|SQ_WAIT[A-Z0-9_]*\|SQ_IFETCH[A-Z0-9_]*\|SQ_INST_LEVEL[A-Z0-9_]*\|SQ_LDS[A-Z0-9_]*" | sort -u > $out/avail.txt
passes=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"
        "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32"
        "SQ_WAVES SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
        "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
        "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH")
for prec in f64 f32; do
  i=0
  for p in "${passes[@]}"; do
    rocprofv3 --kernel-trace --pmc $p --output-format csv -d $out/${prec}_$i -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra-regions --precision $prec > $out/${prec}_$i.json 2> $out/${prec}_$i.err || echo "pass $i failed" >> $out/${prec}.log
    i=$((i+1))
  done
  python3 - $out $prec <<'PY' > $out/$prec.txt
import csv, glob, sys, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/" + sys.argv[2] + "_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ukf_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
w = acc["SQ_WAVES"] / max(1, cnt["SQ_WAVES"])
for k in sorted(acc):
    print("%-28s %10.1f per wave" % (k, acc[k] / cnt[k] / w))
PY
  cat $out/$prec.txt
done
