#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 7): OrientationState lane tables -- GPU suite, A/B
set -o pipefail
out=gpurun_out/r04_job7; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -8 $out/pytest.log
AB_STEPS=100 AB_ARGS="--workload orient --filters 4194304" tools/ab.sh f32 4 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/v_asm.so slam-pose_estimation_amd/lib/ab/v_otab.so > $out/ab_cfg4.txt 2>&1
cat $out/ab_cfg4.txt
AB_STEPS=100 AB_ARGS="--workload orient" tools/ab.sh f64 4 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/v_asm.so slam-pose_estimation_amd/lib/ab/v_otab.so > $out/ab_orient64.txt 2>&1
cat $out/ab_orient64.txt
AB_STEPS=100 AB_ARGS="--workload orient --filters 4194304 --cycles-per-launch 8" tools/ab.sh f32 2 slam-pose_estimation_amd/lib/ab/v_asm.so slam-pose_estimation_amd/lib/ab/v_otab.so > $out/ab_cfg4_multi.txt 2>&1
cat $out/ab_cfg4_multi.txt
