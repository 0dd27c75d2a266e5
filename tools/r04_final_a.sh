#!/bin/bash
# Runs ON THE GPU BOX (round 4, final A): GPU suite, randomised parity, stamps and trip counts on the final library
set -o pipefail
out=gpurun_out/r04_final_a; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
for seed in 51 52; do
  timeout -k 10 420 python3 tests/fuzz_parity.py 3000 $seed > $out/fuzz_parity_$seed.txt 2>&1; echo "fuzz_parity seed $seed rc=$?"; tail -3 $out/fuzz_parity_$seed.txt
done
timeout -k 10 300 python3 tests/fuzz_round3.py 300 7 > $out/fuzz_round3.txt 2>&1; echo "fuzz_round3 rc=$?"; tail -3 $out/fuzz_round3.txt
for w in "pose f64" "pose f32" "orient f32" "orient f64"; do
  set -- $w
  UKFB_LIB=$PWD/slam-pose_estimation_amd/lib/ab/stamps.so timeout -k 10 300 python3 tools/phase_stamps.py $1 $2 262144 > $out/stamps_$1_$2.txt 2>&1 || tail -5 $out/stamps_$1_$2.txt
done
UKFB_LIB=$PWD/slam-pose_estimation_amd/lib/ab/counts.so timeout -k 10 300 python3 tools/phase_stamps.py pose f64 262144 --counts > $out/counts_pose_f64.txt 2>&1
cat $out/stamps_pose_f64.txt $out/stamps_orient_f32.txt
