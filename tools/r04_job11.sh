#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 11): the persistent form of the plain fused cycle (Pose fp64) -- parity inside a bench run,
# the GPU suite on the candidate, then same-box A/B against the shipped library
set -o pipefail
out=gpurun_out/r04_job11; mkdir -p $out
export TMPDIR=/tmp
L=slam-pose_estimation_amd/lib
UKFB_LIB=$PWD/$L/ab/persist.so timeout -k 10 300 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra-regions > $out/bench_persist.json 2> $out/bench_persist.err || { tail -5 $out/bench_persist.err; exit 1; }
python3 - $out/bench_persist.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("persist bench:", round(d["value"]/1e6,1), "M/s", d["roofline"]["kernel"], "status", d["status_or"], "parity", d["parity"]["max_abs_mu"], d["parity"]["max_abs_cov"], d["parity"]["ok"])
sys.exit(0 if d["parity"]["ok"] and d["status_or"] == 0 else 1)
PY
[ $? -ne 0 ] && exit 1
UKFB_LIB=$PWD/$L/ab/persist.so timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc -ne 0 ] && exit $rc
tools/ab.sh f64 4 $L/libukf_batch.so $L/ab/persist.so
AB_STEPS=20 tools/ab.sh f64 3 $L/libukf_batch.so $L/ab/persist.so
AB_ARGS="--filters 524288" tools/ab.sh f64 3 $L/libukf_batch.so $L/ab/persist.so
for w in 10 11 12; do echo "WG_CU=$w"; UKFB_PERSIST_WG_CU=$w tools/ab.sh f64 2 $L/ab/persist.so; done
