#!/bin/bash
# Runs ON THE GPU BOX: N back-to-back bench.py runs, one line each (value, kernel ms, wall ms/step).
# usage: tools/repeat_bench.sh <n> [bench args...]
n=$1; shift
for i in $(seq $n); do
  timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%8.1f M/s   kernel %.3f ms   step %.3f ms' % (d['value'] / 1e6, d['roofline']['kernel_ms_per_launch'], d['ms_per_step']))"
done
