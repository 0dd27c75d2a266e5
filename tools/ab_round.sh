#!/bin/bash
# Runs ON THE GPU BOX: parity tests on a candidate engine build, then an interleaved same-box A/B against other builds.
# usage: tools/ab_round.sh <tag> <candidate.so> "<ab.sh argument sets, ';'-separated>"   e.g. "f64 3;f32 3"
set -o pipefail
tag=$1; cand=$2; sets=$3
out=gpurun_out/r02_$tag; mkdir -p $out
export TMPDIR=/tmp
UKFB_LIB=$PWD/$cand timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_configs.py tests/test_gpu_engine_behaviour.py -m gpu -x -q > $out/pytest.log 2>&1
rc=$?; tail -3 $out/pytest.log; [ $rc -ne 0 ] && exit $rc
IFS=';' read -ra S <<< "$sets"
for s in "${S[@]}"; do
  echo "== $s"
  case "$s" in
    AB_*) eval "$s" | tee -a $out/ab.txt ;;           # e.g. AB_ARGS="--workload orient" tools/ab.sh f32 2 a.so b.so
    *) eval "tools/ab.sh $s" | tee -a $out/ab.txt ;;
  esac
done
