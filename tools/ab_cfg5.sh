#!/bin/bash
# Runs ON THE GPU BOX: config 5 (262 144 Pose filters, per-filter model ids) grouped by update class against filter order,
# and the uniform headline workload at the same size, interleaved on one box.  usage: tools/ab_cfg5.sh <rounds> [filters]
rounds=${1:-3}; n=${2:-262144}
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), d['roofline']['kernel'])"; }
for r in $(seq $rounds); do
  for what in "buckets:--workload pose-mixed --bucket-models 1" "order:--workload pose-mixed --bucket-models 0" "uniform:--workload pose"; do
    name=${what%%:*}; args=${what#*:}
    v=$(timeout -k 10 300 python3 bench.py --filters $n --steps ${AB_STEPS:-200} --warmup 20 --no-cpu-baseline --no-parity --no-extra-regions $args 2>/dev/null | val)
    echo "$name $v"
  done
done
