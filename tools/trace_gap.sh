#!/bin/bash
# Runs ON THE GPU BOX: repeat bench under rocprofv3 kernel+HIP-API trace until a run shows the late-completion
# anomaly, keep that run's traces under gpurun_out/trace_gap/.
export TMPDIR=/tmp
out=$PWD/gpurun_out/trace_gap; rm -rf $out; mkdir -p $out
for i in 1 2 3 4 5 6 7 8; do
  rm -rf $out/run; 
  UKFB_BENCH_DEBUG=1 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $out/run -- python3 bench.py --precision f32 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/err.txt
  line=$(grep launch $out/err.txt); echo "$i $line"
  ms=$(echo "$line" | sed 's/.*event sync \([0-9.]*\) ms.*/\1/')
  if python3 -c "import sys; sys.exit(0 if float('$ms') > 50 else 1)"; then echo "slow run captured"; break; fi
done
python3 - <<'PY'
import csv, glob
kt = glob.glob("gpurun_out/trace_gap/run/**/*kernel_trace.csv", recursive=True)
ht = glob.glob("gpurun_out/trace_gap/run/**/*hip_api_trace.csv", recursive=True)
print(kt, ht)
ks = [r for f in kt for r in csv.DictReader(open(f)) if "ukf_kernel16" in r["Kernel_Name"]]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
last = ks[-30:]
t0 = int(last[0]["Start_Timestamp"])
print("timed kernels: first start 0, last end %.3f ms" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e6))
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(last[:-1], last[1:])]
print("max gap between timed kernels %.1f us" % max(gaps))
rows = [r for f in ht for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# API calls around the timed region
sel = [r for r in rows if int(r["End_Timestamp"]) > t0 - 2_000_000]
import collections
c = collections.Counter()
first_q = None; last_q = None
for r in sel:
    c[r["Function"]] += 1
print(dict(c))
for r in sel[:8]: print(r["Function"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6)
q = [r for r in sel if r["Function"] in ("hipEventQuery", "hipEventSynchronize", "hipEventElapsedTime", "hipDeviceSynchronize", "hipStreamSynchronize")]
if q:
    print("first wait call at %.3f ms, last wait call ends at %.3f ms" % ((int(q[0]["Start_Timestamp"]) - t0) / 1e6, (int(q[-1]["End_Timestamp"]) - t0) / 1e6))
    for r in q[-6:]: print(r["Function"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6)
PY
