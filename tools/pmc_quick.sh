#!/bin/bash
# Runs ON THE GPU BOX: one SQ pass over the headline workload (fp64 and fp32), VALU / LDS instructions per wavefront.
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_quick; rm -rf $out; mkdir -p $out
for prec in f64 f32; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/$prec -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-extra-regions --precision $prec > $out/$prec.json 2> $out/$prec.err
  python3 - $out/$prec <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ukf_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
w = acc["SQ_WAVES"] / max(1, cnt["SQ_WAVES"])
print(sys.argv[1].split("/")[-1], {k: round(acc[k] / cnt[k] / w, 1) for k in acc if k != "SQ_WAVES"})
PY
done
