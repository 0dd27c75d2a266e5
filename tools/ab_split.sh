#!/bin/bash
# Runs ON THE GPU BOX: split launches (two half launches on two streams) against single launches, interleaved, per batch size.
# usage: tools/ab_split.sh <f64|f32> <rounds> <filters> [<filters> ...]
prec=$1; rounds=$2; shift 2
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1))"; }
for n in "$@"; do
  for r in $(seq $rounds); do
    for sp in 1 0; do
      v=$(timeout -k 10 300 python3 bench.py --precision $prec --filters $n --split-streams $sp --steps ${AB_STEPS:-300} --warmup 20 --no-cpu-baseline --no-parity --no-extra-regions $AB_ARGS 2>/dev/null | val)
      echo "$prec n=$n split=$sp $v"
    done
  done
done | sort -k1,3 -s | awk '{k=$1" "$2" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k]}' | sort
