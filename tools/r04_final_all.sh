#!/bin/bash
# Runs ON THE GPU BOX (round 4): the end-of-round evidence on the final library in one call -- GPU suite, randomised parity, stamps and
# trip counts (tools/r04_final_a.sh), the bench lines of every configuration (tools/gpu_bench.sh final ...), kernel traces of the
# configurations the PMC script does not trace (tools/r04_job13.sh)
tools/r04_final_a.sh > gpurun_out/r04_final_a.log 2>&1; tail -25 gpurun_out/r04_final_a.log | cut -c1-200
tools/gpu_bench.sh final f64_20 f64 f32 cfg2 cfg3 cfg4 cfg4_64 cfg5 cfg5_order cfg2_single shard8 shard8_single f64_1000 group2 cfg3w cfg4w uni262k full cabi 2>&1 | tee gpurun_out/r04_final_bench.log
tools/r04_job13.sh 2>&1 | tee gpurun_out/r04_final_traces.log
