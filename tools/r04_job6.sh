#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 6): GPU suite on the product; the fp64 one-wavefront-per-filter layouts (GENERIC_F64 build) through
# the (prec, G) cases that skip on the shipped library; same-box A/B of the last cuts
set -o pipefail
out=gpurun_out/r04_job6; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -6 $out/pytest.log
UKFB_LIB=$PWD/slam-pose_estimation_amd/lib/ab/generic64.so timeout -k 10 900 python -m pytest tests -m gpu -q -k "G or lanes or layout" > $out/pytest_generic64.log 2>&1; echo "generic64 rc=$?" | tee -a $out/pytest_generic64.log
tail -6 $out/pytest_generic64.log
AB_STEPS=200 tools/ab.sh f64 4 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/v_pair.so slam-pose_estimation_amd/lib/ab/v_pair32.so slam-pose_estimation_amd/lib/ab/v_asm.so > $out/ab_f64.txt 2>&1
cat $out/ab_f64.txt
AB_STEPS=200 tools/ab.sh f32 3 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/v_pair32.so slam-pose_estimation_amd/lib/ab/v_asm.so > $out/ab_f32.txt 2>&1
cat $out/ab_f32.txt
AB_STEPS=100 AB_ARGS="--workload orient --filters 4194304" tools/ab.sh f32 3 slam-pose_estimation_amd/lib/ab/r4base.so slam-pose_estimation_amd/lib/ab/v_pair32.so slam-pose_estimation_amd/lib/ab/v_asm.so > $out/ab_cfg4.txt 2>&1
cat $out/ab_cfg4.txt
