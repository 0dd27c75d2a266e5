#!/bin/bash
# Runs ON THE GPU BOX: interleaved A/B of ENVIRONMENT settings on the product library (same device), sorted values.
# usage: [AB_ARGS="..."] tools/ab_env.sh <f64|f32> <rounds> "VAR=a" "VAR=b" ...
prec=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for kv in "$@"; do
    v=$(env $kv timeout -k 10 200 python bench.py --precision $prec --steps ${AB_STEPS:-100} --warmup 10 --no-cpu-baseline --no-parity --no-extra-regions $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1))")
    echo "$kv $v"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}'
