#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 15): VALU instructions per wavefront of config 5's main kernel, dispatch by dispatch, with the
# four-launch grouping of commit 95fc77a~1 and with the two-launch grouping (same kernel code: are the launches the same work?)
export TMPDIR=/tmp
out=gpurun_out/r04_job15; mkdir -p $out
L=$PWD/slam-pose_estimation_amd/lib
for v in old new; do
  if [ $v = old ]; then export UKFB_LIB=$L/ab/oldgroup.so; else export UKFB_LIB=$L/libukf_batch.so; fi
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d $out/pmc_$v -- python3 bench.py --no-cpu-baseline --no-parity --no-extra-regions --workload pose-mixed --filters 262144 --steps 200 --warmup 10 > $out/pmc_$v.json 2> $out/pmc_$v.err || tail -3 $out/pmc_$v.err
done
python3 - $out <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
res={}
for v in ("old","new"):
    f=glob.glob(f"{out}/pmc_{v}/**/*counter_collection.csv",recursive=True)[0]
    rows=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "ukf_kernel16" not in r["Kernel_Name"]: continue
        rows.setdefault(int(r["Dispatch_Id"]),{})[r["Counter_Name"]]=float(r["Counter_Value"])
    seq=[d["SQ_INSTS_VALU"]/d["SQ_WAVES"] for d in rows.values() if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d]
    res[v]=seq
    print(v, "dispatches", len(seq), "mean", round(sum(seq)/len(seq),1), "first 5", [round(x) for x in seq[:5]], "last 5", [round(x) for x in seq[-5:]])
n=min(len(res["old"]),len(res["new"]))
d=[abs(a-b) for a,b in zip(res["old"][:n],res["new"][:n])]
print("max |old-new| per dispatch over", n, ":", round(max(d),2))
open(f"{out}/valu_per_wave_by_dispatch.txt","w").write("\n".join(f"{i} {a:.1f} {b:.1f}" for i,(a,b) in enumerate(zip(res["old"][:n],res["new"][:n]))))
PY
