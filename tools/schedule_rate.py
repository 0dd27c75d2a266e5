#!/usr/bin/env python3
"""Runs ON THE GPU BOX: an IMU-rate filter with a slower aiding sensor (BASELINE config 2's shape: 100 Hz predictions, a 10 Hz
position fix), replayed from device-resident buffers -- ten launches per fix (nine predictions + one fused cycle) against ONE
scheduled multi-cycle launch per fix (ukfb_cycle_schedule_dev).  Prints filter-predictions/s for both.

usage: python3 tools/schedule_rate.py [filters=1048576] [f64|f32] [fixes=30]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import slam_pose_estimation_amd as spe  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
prec = spe.F32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else spe.F64
fixes = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tdt = torch.float64 if prec == spe.F64 else torch.float32
RATE, SLOTS = 10, 10
CH = 131072


def build():
    e = spe.BatchPoseUKF(n, precision=prec)
    for lo in range(0, n, CH):
        mu, cov = spe.synth.pose_initial(min(CH, n - lo), first=lo)
        e.initialize(mu, cov, first=lo)
    e.set_acceleration(None, 0.01 * np.eye(3))
    return e


acc = torch.empty((SLOTS, n, 3), dtype=tdt, device="cuda").uniform_(-0.5, 0.5)
z = torch.empty((SLOTS, n, 3), dtype=tdt, device="cuda").uniform_(-10, 10)
Q = (0.05 ** 2 * torch.eye(3, dtype=tdt, device="cuda")).reshape(1, 1, 9).repeat(SLOTS, n, 1).contiguous()
dts = np.full(RATE, 0.01)
models = np.full(RATE, -1, dtype=np.int32)
models[-1] = spe.MEAS_POS3
torch.cuda.synchronize()


def separate(e):
    for c in range(RATE):
        e.bind_acceleration_dev(acc[c])
        if models[c] < 0:
            e.predict(0.01)
        else:
            e.cycle_dev(0.01, int(models[c]), z[c], Q[c])


def scheduled(e):
    e.cycle_schedule_dev(dts, models, z, Q, SLOTS, 0, in_a_dev=acc)


res = {}
for name, fn in (("ten launches per fix", separate), ("one scheduled launch per fix", scheduled)):
    e = build()
    for _ in range(5):
        fn(e)
    e.sync()
    t0 = time.perf_counter()
    for _ in range(fixes):
        fn(e)
    e.sync()
    dt = time.perf_counter() - t0
    res[name] = n * RATE * fixes / dt
    assert e.status_summary() == 0
    print(f"{name:30s} {res[name] / 1e6:8.1f} M filter-predictions/s  ({dt / fixes * 1e3:.3f} ms per fix of {RATE} IMU samples, {n} filters)")
    e.close()
print(f"ratio {res['one scheduled launch per fix'] / res['ten launches per fix']:.3f}")
