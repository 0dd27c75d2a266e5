#!/bin/bash
# Runs ON THE GPU BOX: GPU test suite, then per-phase stamps and reference bench lines (files under gpurun_out/).
set -o pipefail
out=gpurun_out/r02_$1; mkdir -p $out
shift
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
for what in "$@"; do
  case $what in
    stamps)
      for p in f64 f32; do
        UKFB_LIB=$PWD/slam-pose_estimation_amd/lib/ab/stamps.so timeout -k 10 300 python3 tools/phase_stamps.py pose $p > $out/stamps_pose_$p.txt 2>&1 || tail -5 $out/stamps_pose_$p.txt
      done
      cat $out/stamps_pose_f64.txt ;;
    bench)
      timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_20.json 2> $out/bench_20.err
      timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
      timeout -k 10 300 python3 bench.py --no-cpu-baseline --precision f32 > $out/bench_f32.json 2> $out/bench_f32.err
      cat $out/bench_20.json $out/bench_default.json $out/bench_f32.json ;;
  esac
done
