#!/bin/bash
set -o pipefail
out=gpurun_out/r04_job9; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -12 $out/pytest.log
A=slam-pose_estimation_amd/lib/ab/r4final.so; B=slam-pose_estimation_amd/lib/ab/v_short2.so
AB_STEPS=200 AB_ARGS="--workload pose-mixed --filters 262144" tools/ab.sh f64 4 $A $B > $out/ab_cfg5.txt 2>&1; cat $out/ab_cfg5.txt
AB_STEPS=200 AB_ARGS="--filters 262144" tools/ab.sh f64 3 $A $B > $out/ab_uni262k.txt 2>&1; cat $out/ab_uni262k.txt
timeout -k 10 300 python3 tests/fuzz_parity.py 1500 62 > $out/fuzz.txt 2>&1; tail -2 $out/fuzz.txt
