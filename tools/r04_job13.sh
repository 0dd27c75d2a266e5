#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 13): rocprofv3 kernel traces (--kernel-trace --stats only) of configs 4, 5, OrientationState fp64 and
# the wide-arithmetic config 3 on the final library, with the bench line of the same run
export TMPDIR=/tmp
out=gpurun_out/r04_job13; mkdir -p $out
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 bench.py --no-cpu-baseline --no-parity --no-extra-regions "$@" > $out/trace_$name.json 2> $out/trace_$name.err || tail -3 $out/trace_$name.err
  f=$(find $out/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/r04_${name}_kernel_stats.csv && head -4 $f | cut -c1-200
  python3 - $out/trace_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "bench under trace", round(d["value"]/1e6,1), "M/s kernel_ms", round(d["roofline"]["kernel_ms_per_launch"],4))
PY
}
run cfg4 --workload orient --precision f32 --filters 4194304
run cfg5 --workload pose-mixed --filters 262144
run orient_f64 --workload orient
run cfg3w --precision f32 --wide-arithmetic 1
