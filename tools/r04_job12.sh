#!/bin/bash
# Runs ON THE GPU BOX (round 4, job 12): more seeds of the randomised parity runs on the final library
set -o pipefail
out=gpurun_out/r04_job12; mkdir -p $out
export TMPDIR=/tmp
for seed in 53 54 55 56 57 58; do
  timeout -k 10 420 python3 tests/fuzz_parity.py 3000 $seed > $out/fuzz_parity_$seed.txt 2>&1; echo "fuzz_parity seed $seed rc=$?"; tail -1 $out/fuzz_parity_$seed.txt
done
for seed in 8 9 10; do
  timeout -k 10 300 python3 tests/fuzz_round3.py 300 $seed > $out/fuzz_round3_$seed.txt 2>&1; echo "fuzz_round3 seed $seed rc=$?"; tail -1 $out/fuzz_round3_$seed.txt
done
