#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 8): short update factorisation -- GPU suite, A/B against the measured final library
set -o pipefail
out=gpurun_out/r04_job8; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -12 $out/pytest.log
A=slam-pose_estimation_amd/lib/ab/r4final.so; B=slam-pose_estimation_amd/lib/ab/v_short.so
AB_STEPS=200 tools/ab.sh f64 4 $A $B > $out/ab_f64.txt 2>&1; cat $out/ab_f64.txt
AB_STEPS=200 tools/ab.sh f32 3 $A $B > $out/ab_f32.txt 2>&1; cat $out/ab_f32.txt
AB_STEPS=100 AB_ARGS="--workload orient --filters 4194304" tools/ab.sh f32 3 $A $B > $out/ab_cfg4.txt 2>&1; cat $out/ab_cfg4.txt
AB_STEPS=100 AB_ARGS="--workload orient" tools/ab.sh f64 3 $A $B > $out/ab_orient64.txt 2>&1; cat $out/ab_orient64.txt
timeout -k 10 300 python3 tests/fuzz_parity.py 1500 61 > $out/fuzz.txt 2>&1; tail -2 $out/fuzz.txt
