#!/usr/bin/env python3
"""Runs ON THE GPU BOX with a -DUKFB_STAMPS engine build (tools/build_variant.sh stamps -DUKFB_STAMPS):
mean shader cycles between consecutive phase markers of ukf_kernel16 (s_memtime, lane 0 of every wavefront),
i.e. where a wavefront's life goes, at several points of the bench trajectory (the orientation spread of the
headline workload grows with the cycle count, which moves the SO(3) log onto its half-angle path).

usage: UKFB_LIB=slam-pose_estimation_amd/lib/ab/stamps.so python3 tools/phase_stamps.py [pose|orient|pose-mixed] [f64|f32] [filters]

With a -DUKFB_COUNTS build and --counts as LAST argument (tools/build_variant.sh counts -DUKFB_COUNTS): per launch, the histogram over
wavefronts of the manifold-mean trip counts (slot0: wave trips of the iteration loop = max over the wavefront's four filters;
slot1: iterations of one filter incl. the first; slot2: final deltas re-based (1) or by a third round of logarithms (0); slot3: size of the first mean delta's rotation part,
largest of the wavefront -- bin b = (1e-(8-b), 1e-(7-b)] rad, bin 0 = up to 1e-7) and how
many of the wavefront-level exp / log calls took their wide-angle paths (dbg = exp calls, exp angle-doubling, exp mod-2pi,
log_n calls, log_n half-angle path, log calls, log half-angle path)."""
import torch  # noqa: F401
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import slam_pose_estimation_amd as spe  # noqa: E402

COUNTS = sys.argv[-1] == "--counts"
if COUNTS:
    sys.argv.pop()
workload = sys.argv[1] if len(sys.argv) > 1 else "pose"
prec = spe.F64 if (len(sys.argv) <= 2 or sys.argv[2] == "f64") else spe.F32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
names = re.findall(r'UKFB_MARK\("([a-z0-9_]+)"\)', open(os.path.join(ROOT, "slam-pose_estimation_amd/csrc/ukf_kernel16.hpp")).read())
out = os.path.join(ROOT, "gpurun_out", "stamps_raw.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
if os.path.exists(out):
    os.remove(out)
td = torch.float64 if prec == spe.F64 else torch.float32
sy = spe.synth
orient = workload == "orient"
if orient:
    e = spe.BatchOrientationUKF(n, sy.ORIENT_TAU, sy.ORIENT_TAU, sy.ORIENT_LATITUDE, precision=prec)
    e.set_process_noise(sy.orient_process_noise())
else:
    e = spe.BatchPoseUKF(n, precision=prec)
ring = []
CH = 131072
bufs = [dict(a=torch.empty((n, 3), dtype=td, device="cuda"), g=torch.empty((n, 3), dtype=td, device="cuda"),
             z=torch.empty((n, 3), dtype=td, device="cuda"), Q=torch.empty((n, 9), dtype=td, device="cuda"),
             m=torch.empty((n,), dtype=torch.int32, device="cuda")) for _ in range(4)]
for lo in range(0, n, CH):
    hi = min(n, lo + CH)
    mu, cov = (sy.orient_initial if orient else sy.pose_initial)(hi - lo, first=lo)
    e.initialize(mu, cov, first=lo)
    for k in range(4):
        b = bufs[k]
        if orient:
            gyro, acc, z, Q = sy.orient_cycle_inputs(hi - lo, k, mu[:, :4], first=lo)
            b["g"][lo:hi] = torch.from_numpy(gyro).to("cuda", td)
        else:
            acc, z, Q = sy.pose_cycle_inputs(hi - lo, k, mu[:, :3], first=lo, random_q=workload == "pose-mixed")
            if workload == "pose-mixed":
                models = sy.pose_mixed_models(hi - lo, k, first=lo)
                z = sy.pose_measurement_for_model(mu, models, z - mu[:, :3])
                b["m"][lo:hi] = torch.from_numpy(models).to("cuda")
        b["a"][lo:hi] = torch.from_numpy(acc).to("cuda", td)
        b["z"][lo:hi] = torch.from_numpy(z).to("cuda", td)
        b["Q"][lo:hi] = torch.from_numpy(Q.reshape(-1, 9)).to("cuda", td)
if not orient:
    e.set_acceleration(None, 0.01 * np.eye(3))
torch.cuda.synchronize()


def step(k):
    b = bufs[k % 4]
    if orient:
        e.bind_orient_inputs_dev(b["g"], b["a"])
        e.cycle_dev(0.01, spe.MEAS_ORIENT_BODYVEL3, b["z"], b["Q"])
    else:
        e.bind_acceleration_dev(b["a"])
        e.cycle_dev(0.01, spe.MEAS_POS3, b["z"], b["Q"], meas_model_dev=b["m"] if workload == "pose-mixed" else None)


report_at = [0, 1, 3, 10, 30, 60, 100, 150, 200, 300, 400, 500, 600] if COUNTS else [3, 60, 150, 300, 600]
k = 0
for target in report_at:
    os.environ.pop("UKFB_STAMP_OUT", None)
    while k < target:
        step(k); k += 1
    os.environ["UKFB_STAMP_OUT"] = out
    step(k); k += 1
os.environ.pop("UKFB_STAMP_OUT", None)
print("status_or", e.status_summary())
lines = open(out).read().strip().splitlines()
if COUNTS:
    print(f"{workload} {'f64' if prec == spe.F64 else 'f32'} n={n}: per-wavefront counters of the launch at cycle index")
    for cyc, line in zip(report_at, lines):
        print(f"@{cyc:4d} {line}")
    sys.exit(0)
print(f"{workload} {'f64' if prec == spe.F64 else 'f32'} n={n}: mean cycles between markers (s_memtime), per launch at cycle index")
table = {}
for cyc, line in zip(report_at, lines):
    parts = line.split()
    life = float(parts[2].split("=")[1])
    d = {int(p.split(":")[0]): float(p.split(":")[1]) for p in parts[3:]}
    table[cyc] = (life, d)
idx = sorted({i for _, d in table.values() for i in d})
print(f"{'phase':18s}" + "".join(f"{('@' + str(c)):>10s}" for c in report_at))
for i in idx:
    print(f"{names[i] if i < len(names) else i:18s}" + "".join(f"{table[c][1].get(i, 0):10.0f}" for c in report_at))
print(f"{'wave life':18s}" + "".join(f"{table[c][0]:10.0f}" for c in report_at))
