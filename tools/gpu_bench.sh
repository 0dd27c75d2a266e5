#!/bin/bash
# Runs ON THE GPU BOX: the reference bench lines of a build (short, default and fp32), compact summary on stdout.
out=gpurun_out/${UKFB_ROUND:-r04}_$1; mkdir -p $out; shift
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || tail -3 $out/$name.err;
  python3 - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{sys.argv[2]:14s} {d['value']/1e6:8.1f} M/s  ms/step {d['ms_per_step']:.4f}  kernel {r['kernel_ms_per_launch']:.4f}  sustained {r.get('kernel_ms_per_launch_sustained')}  burst {r.get('kernel_ms_per_launch_burst')}  frac {r['frac']:.3f}  status {d['status_or']}  parity {d.get('parity') and (d['parity']['max_abs_mu'], d['parity']['max_abs_cov'], d['parity']['ok'])}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for what in "$@"; do
  case $what in
    f64_20) run f64_20 --steps 20 --warmup 5 --no-cpu-baseline ;;
    f64) run f64 --no-cpu-baseline ;;
    f32) run f32 --no-cpu-baseline --precision f32 ;;
    f32_20) run f32_20 --steps 20 --warmup 5 --no-cpu-baseline --precision f32 ;;
    cfg2) run cfg2 --no-cpu-baseline --filters 65536 ;;
    cfg3) run cfg3 --no-cpu-baseline --filters 131072 --precision f32 ;;
    cfg4) run cfg4 --no-cpu-baseline --workload orient --precision f32 --filters 4194304 ;;
    cfg4_64) run cfg4_64 --no-cpu-baseline --workload orient --filters 1048576 ;;
    cfg5) run cfg5 --no-cpu-baseline --workload pose-mixed --filters 262144 ;;
    cfg5_order) run cfg5_order --no-cpu-baseline --workload pose-mixed --filters 262144 --bucket-models 0 ;;
    cfg2_single) run cfg2_single --no-cpu-baseline --filters 65536 --split-streams 0 ;;
    shard8) run shard8 --no-cpu-baseline --filters 131072 ;;
    shard8_single) run shard8_single --no-cpu-baseline --filters 131072 --split-streams 0 ;;
    f64_1000) run f64_1000 --no-cpu-baseline --steps 1000 --no-extra-regions ;;
    track_1000) run track_1000 --no-cpu-baseline --steps 1000 --no-extra-regions --inputs tracking ;;
    group2) run group2 --no-cpu-baseline --launcher group --gpus 2 --group-devices 0,0 ;;
    cfg3w) run cfg3w --no-cpu-baseline --precision f32 --wide-arithmetic 1 ;;
    cfg4w) run cfg4w --no-cpu-baseline --workload orient --precision f32 --filters 4194304 --wide-arithmetic 1 ;;
    uni262k) run uni262k --no-cpu-baseline --filters 262144 ;;
    full) run full ;;
    cabi) make -s -C tests/cpp build/cabi_bench 2>/dev/null; for a in "1048576 500 f64" "1048576 500 f32" "131072 500 f64" "1048576 500 f64 2"; do tests/cpp/build/cabi_bench $a | tee -a $out/cabi.txt; done ;;
  esac
done
