#!/bin/bash
# Runs ON THE GPU BOX: calibrates FETCH_SIZE / WRITE_SIZE on the engine's access pattern, then measures the
# fused kernel of bench.py's default workload and writes profiles-ready JSON to gpurun_out/traffic_<prec>.json
# usage: tools/traffic.sh <f64|f32> [filters]
set -u
prec=${1:-f64}; n=${2:-1048576}
out=$PWD/gpurun_out/traffic_$prec; mkdir -p $out
export TMPDIR=/tmp
hipcc -O2 --offload-arch=gfx950 tools/calib_traffic.hip -o $out/calib_traffic || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/calib_$c -- $out/calib_traffic $n $prec > $out/calib_$c.json 2> $out/calib_$c.err
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/bench_$c -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --filters $n --precision $prec > $out/bench_$c.json 2> $out/bench_$c.err
done
python3 tools/traffic_report.py $out $prec $n | tee $out/traffic.json
