#!/bin/bash
# Runs ON THE GPU BOX: the headline kernel at reduced occupancy (extra dynamic LDS per workgroup, UKFB_LDS_PAD_BYTES),
# interleaved: how much does a wavefront per SIMD buy?   usage: tools/occ_sweep.sh <f64|f32> <rounds> pad0 pad1 ...
prec=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for pad in "$@"; do
    v=$(UKFB_LDS_PAD_BYTES=$pad timeout -k 10 200 python3 bench.py --precision $prec --steps ${AB_STEPS:-150} --warmup 10 --no-cpu-baseline --no-parity --no-extra-regions $AB_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), d['roofline']['lds_bytes_per_workgroup'])")
    echo "pad=$pad $v"
  done
done | sort -s -k1,1 | awk '{a[$1" lds="$3]=a[$1" lds="$3]" "$2} END{for(k in a) print k, a[k]}' | sort
