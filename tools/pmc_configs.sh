#!/bin/bash
# Runs ON THE GPU BOX: per-configuration PMC passes for bench.py's roofline object (all five BASELINE configs).
#   - FETCH_SIZE and WRITE_SIZE in separate passes, corrected with factors calibrated on the engine's own access
#     pattern (tools/calib_traffic.hip; MI355X_MICROARCH.md: FETCH_SIZE halves wide coalesced reads on gfx950)
#   - one SQ/GRBM pass: VALU wave-instructions, VALU busy quad-cycles, wave cycles, GRBM_GUI_ACTIVE (clock)
#   - two SQ passes: VALU instructions by class (FMA / MUL / ADD / TRANS per precision, INT32, INT64, CVT) for the weighted issue model
#   - rocprofv3 --kernel-trace --stats of the default bench command
# Output: gpurun_out/pmc_configs/{traffic_latest.json,pmc_latest.json,*_kernel_stats.csv,summary.txt}
# usage: tools/pmc_configs.sh [tag] [part]   part = all | 1 | 2: a gpurun call is limited to 20 minutes, the whole set takes
#        about 25 -- run parts 1 and 2 in two calls (their outputs merge in gpurun_out/pmc_configs) and then
#        `python3 tools/pmc_report.py gpurun_out/pmc_configs <tag>` in the container; copy the json / csv files into profiles/
set -u
tag=${1:-r04}
part=${2:-all}
out=$PWD/gpurun_out/pmc_configs; mkdir -p $out
export TMPDIR=/tmp
if [ $part != 2 ]; then
hipcc -O2 --offload-arch=gfx950 tools/calib_traffic.hip -o $out/calib_traffic || exit 1
for prec in f64 f32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/calib_${prec}_$c -- $out/calib_traffic 1048576 $prec > $out/calib_${prec}_$c.json 2> $out/calib_${prec}_$c.err
  done
done
rm -f $out/calib_traffic
fi
# (a fixed pre-roll of 384 cycles instead of bench.py's 0.4 s: the workload is not stationary -- later cycles cost more
#  instructions -- and a time span makes the set of profiled launches depend on the box and on the pass; earlier rounds' passes
#  saw 330...400 pre-roll cycles on the headline, which is why 384: cycles 0...383, then 0...49 again)
B="--steps 40 --warmup 10 --clock-warmup-cycles 384 --no-cpu-baseline --no-parity --no-extra-regions"
cfg() {  # name, bench args
  name=$1; shift
  for pass in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
      "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
      "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU"; do
    p=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/${name}_$p -- python3 bench.py $B "$@" > $out/${name}_$p.json 2> $out/${name}_$p.err
    echo "$name $p rc=$?"
  done
}
if [ $part != 2 ]; then
cfg headline_f64
cfg headline_f32 --precision f32
# (small launches: one launch per cycle for the counters -- the default runs them as two half launches on two streams,
#  whose per-dispatch counters would describe half a batch)
cfg cfg2 --filters 65536 --split-streams 0
cfg cfg3 --filters 131072 --precision f32 --split-streams 0
cfg multi8_f64 --cycles-per-launch 8 --warmup 16
fi
if [ $part = 1 ]; then exit 0; fi
cfg cfg4 --workload orient --precision f32 --filters 4194304
cfg cfg5 --workload pose-mixed --filters 262144
cfg multi8_f32 --cycles-per-launch 8 --warmup 16 --precision f32
cfg wide_f32 --precision f32 --wide-arithmetic 1
cfg cfg4_wide --workload orient --precision f32 --filters 4194304 --wide-arithmetic 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_cfg2 -- python3 bench.py --no-cpu-baseline --no-parity --no-extra-regions --filters 65536 --split-streams 0 > $out/trace_cfg2.json 2> $out/trace_cfg2.err
# kernel stats of the default command (the driver's own invocation and the 500-step default)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_default -- python3 bench.py --no-cpu-baseline --no-parity --no-extra-regions > $out/trace_default.json 2> $out/trace_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_f32 -- python3 bench.py --no-cpu-baseline --no-parity --no-extra-regions --precision f32 > $out/trace_f32.json 2> $out/trace_f32.err
if [ $part = all ]; then python3 tools/pmc_report.py $out $tag | tee $out/summary.txt; fi
