#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 1): GPU suite on the cleaned-up library, trip-count histograms, phase stamps, a bench line.
set -o pipefail
out=gpurun_out/r04_job1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -4 $out/pytest.log
for w in "pose f64" "pose f32" "orient f32" "orient f64"; do
  set -- $w
  UKFB_LIB=$PWD/slam-pose_estimation_amd/lib/ab/counts.so timeout -k 10 300 python3 tools/phase_stamps.py $1 $2 262144 --counts > $out/counts_$1_$2.txt 2>&1 || tail -5 $out/counts_$1_$2.txt
  UKFB_LIB=$PWD/slam-pose_estimation_amd/lib/ab/stamps.so timeout -k 10 300 python3 tools/phase_stamps.py $1 $2 262144 > $out/stamps_$1_$2.txt 2>&1 || tail -5 $out/stamps_$1_$2.txt
done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_20.json 2> $out/bench_20.err
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
cat $out/counts_pose_f64.txt $out/stamps_pose_f64.txt $out/bench_20.json
