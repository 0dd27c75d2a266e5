#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 5): paired logarithms -- GPU suite, A/B against the round's base
set -o pipefail
out=gpurun_out/r04_job5; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -8 $out/pytest.log
cp slam-pose_estimation_amd/lib/libukf_batch.so slam-pose_estimation_amd/lib/ab/product.so
AB_STEPS=200 tools/ab.sh f64 4 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/product.so > $out/ab_f64.txt 2>&1
cat $out/ab_f64.txt
AB_STEPS=200 tools/ab.sh f32 3 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/product.so > $out/ab_f32.txt 2>&1
cat $out/ab_f32.txt
AB_STEPS=100 AB_ARGS="--workload orient --filters 4194304" tools/ab.sh f32 3 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/product.so > $out/ab_cfg4.txt 2>&1
cat $out/ab_cfg4.txt
AB_STEPS=100 AB_ARGS="--workload orient" tools/ab.sh f64 3 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/product.so > $out/ab_orient64.txt 2>&1
cat $out/ab_orient64.txt
