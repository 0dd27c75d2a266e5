#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 14): model-class grouping in two launches (count, scatter with its own prefix sums) instead of four
set -o pipefail
out=gpurun_out/r04_job14; mkdir -p $out
export TMPDIR=/tmp
L=slam-pose_estimation_amd/lib
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tests/fuzz_round3.py 300 11 > $out/fuzz_round3.txt 2>&1; echo "fuzz_round3 rc=$?"; tail -2 $out/fuzz_round3.txt
AB_ARGS="--workload pose-mixed --filters 262144" tools/ab.sh f64 4 $L/ab/base.so $L/libukf_batch.so
AB_ARGS="--workload pose-mixed --filters 1048576" tools/ab.sh f64 3 $L/ab/base.so $L/libukf_batch.so
AB_ARGS="--workload pose-mixed --filters 4194304" tools/ab.sh f32 2 $L/ab/base.so $L/libukf_batch.so
AB_ARGS="--filters 262144" tools/ab.sh f64 2 $L/libukf_batch.so
