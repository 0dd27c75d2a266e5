#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 2): the wide-arithmetic mode -- its tests, the whole GPU suite, bench lines of configs 3 / 4
set -o pipefail
out=gpurun_out/r04_job2; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_wide_arithmetic.py -x -q -s > $out/pytest_wide.log 2>&1; echo "wide rc=$?" | tee -a $out/pytest_wide.log
tail -30 $out/pytest_wide.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_wide_arithmetic.py > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -4 $out/pytest.log
timeout -k 10 400 python3 bench.py --precision f32 --wide-arithmetic 1 --no-cpu-baseline > $out/bench_cfg3_wide.json 2> $out/bench_cfg3_wide.err
timeout -k 10 400 python3 bench.py --precision f32 --no-cpu-baseline > $out/bench_cfg3_f32.json 2> $out/bench_cfg3_f32.err
timeout -k 10 500 python3 bench.py --workload orient --precision f32 --filters 4194304 --wide-arithmetic 1 --no-cpu-baseline > $out/bench_cfg4_wide.json 2> $out/bench_cfg4_wide.err
timeout -k 10 500 python3 bench.py --workload orient --precision f32 --filters 4194304 --no-cpu-baseline > $out/bench_cfg4_f32.json 2> $out/bench_cfg4_f32.err
timeout -k 10 300 python3 bench.py --workload orient --filters 4194304 --no-cpu-baseline --steps 100 > $out/bench_orient_f64.json 2> $out/bench_orient_f64.err
for f in cfg3_wide cfg3_f32 cfg4_wide cfg4_f32 orient_f64; do python3 - $out/bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d.get("parity") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"]/1e6,1), "M", d["dtype"], d["roofline"]["kernel"], "frac", round(d["roofline"]["frac"],3), "parity", p.get("ok"), p.get("max_abs_mu"), p.get("max_abs_cov"), p.get("horizon_cycles"), p.get("within_horizon"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
done
