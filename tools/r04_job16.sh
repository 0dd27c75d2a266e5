#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 16): four more seeds of the randomised parity runs on the final library 813288388fa3e027
set -o pipefail
out=gpurun_out/r04_job16; mkdir -p $out
export TMPDIR=/tmp
for seed in 61 62 63 64; do
  timeout -k 10 420 python3 tests/fuzz_parity.py 3000 $seed > $out/fuzz_parity_$seed.txt 2>&1; echo "fuzz_parity seed $seed rc=$?"; tail -1 $out/fuzz_parity_$seed.txt
done
