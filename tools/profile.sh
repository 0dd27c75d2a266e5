#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace/stats + PMC passes over bench.py.
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/...
# PMC passes are separate runs without any trace domain besides --kernel-trace (gpurun rule).
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $ARGS > $out/bench_trace.json 2> $out/trace.err
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -- python3 bench.py $ARGS > /dev/null 2> $out/pmc_$name.err
done
python3 tools/summarize_profile.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
