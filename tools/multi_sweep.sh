#!/bin/bash
# Runs ON THE GPU BOX: filter-cycles/s of the headline workload as launches of C cycles (ukfb_cycle_multi_dev), C swept,
# interleaved rounds on one device.  usage: tools/multi_sweep.sh <f64|f32> <rounds> [bench args...]
prec=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for c in 1 2 4 8 16 32; do
    v=$(timeout -k 10 200 python bench.py --precision $prec --steps 192 --warmup 32 --cycles-per-launch $c --no-cpu-baseline --no-parity --no-extra-regions "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), d['roofline']['kernel'].split(',')[-1])")
    echo "C=$c $v"
  done
done | sort -t= -k2 -n | awk '{a[$1]=a[$1]" "$2; k[$1]=$3} END{for(x in a) print x, a[x], k[x]}' | sort -t= -k2 -n
