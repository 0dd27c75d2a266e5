#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 10): size of the first mean delta on the bench workloads (counts build), then the
# first-iteration rebase against the shipped library
set -o pipefail
out=gpurun_out/r04_job10; mkdir -p $out
export TMPDIR=/tmp
L=slam-pose_estimation_amd/lib
for w in "pose f64" "orient f32"; do
  set -- $w
  UKFB_LIB=$PWD/$L/ab/counts.so timeout -k 10 300 python3 tools/phase_stamps.py $1 $2 262144 --counts > $out/counts_$1_$2.txt 2>&1 || tail -5 $out/counts_$1_$2.txt
  cut -c1-260 $out/counts_$1_$2.txt
done
tools/ab_round.sh rb0 $L/ab/rb0.so "AB_ARGS=\"--workload orient --filters 4194304\" tools/ab.sh f32 3 $L/libukf_batch.so $L/ab/rb0.so;AB_ARGS=\"--workload orient\" tools/ab.sh f64 3 $L/libukf_batch.so $L/ab/rb0.so;f64 3 $L/libukf_batch.so $L/ab/rb0.so;f32 3 $L/libukf_batch.so $L/ab/rb0.so"
