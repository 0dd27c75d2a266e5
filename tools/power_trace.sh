#!/bin/bash
# Runs ON THE GPU BOX: samples rocm-smi power / clocks while bench.py runs a long region.
# usage: tools/power_trace.sh [bench args...]
python bench.py --no-cpu-baseline --steps 6000 --warmup 10 "$@" > /tmp/pt_bench.json 2>/dev/null &
pid=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|junction" | tr '\n' ' '; echo
  sleep 1
done
wait $pid
python -c "import json; d=json.load(open('/tmp/pt_bench.json')); print(round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms/step')"
