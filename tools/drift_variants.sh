#!/bin/bash
# Runs ON THE GPU BOX: tests/drift_f32.py (fp32 engine beside the fp64 and the float oracle) for the product library and
# every diagnostic build given.   usage: tools/drift_variants.sh <outdir> <filters> <cycles> [lib.so ...]
out=$1; n=$2; cyc=$3; shift 3
mkdir -p $out
export TMPDIR=/tmp
for wl in pose orient; do
  timeout -k 10 500 python3 tests/drift_f32.py $wl $n $cyc > $out/drift_${wl}_product.txt 2> $out/drift_${wl}_product.err || tail -3 $out/drift_${wl}_product.err
  for lib in "$@"; do
    name=$(basename $lib .so)
    UKFB_LIB=$PWD/$lib timeout -k 10 500 python3 tests/drift_f32.py $wl $n $cyc > $out/drift_${wl}_$name.txt 2> $out/drift_${wl}_$name.err || tail -3 $out/drift_${wl}_$name.err
  done
done
tail -n +1 $out/drift_*.txt
