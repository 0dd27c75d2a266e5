"""PCIe-inclusive rate of the HOST-pointer entry points (ukfb_cycle with numpy arrays: z, Q uploaded every call) next
to the device-pointer entry (what bench.py times).  usage: python3 tools/host_path_rate.py [filters] [f64|f32]"""
import torch  # noqa: F401
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_pose_estimation_amd as spe

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
prec = spe.F32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else spe.F64
td = torch.float64 if prec == spe.F64 else torch.float32
e = spe.BatchPoseUKF(n, precision=prec)
CH = 131072
zs, Qs, accs = [], [], []
for lo in range(0, n, CH):
    hi = min(n, lo + CH)
    mu, cov = spe.synth.pose_initial(hi - lo, first=lo); e.initialize(mu, cov, first=lo)
    acc, z, Q = spe.synth.pose_cycle_inputs(hi - lo, 0, mu[:, :3], first=lo)
    zs.append(z); Qs.append(Q); accs.append(acc)
z = np.concatenate(zs); Q = np.concatenate(Qs); acc = np.concatenate(accs)
e.set_acceleration(acc, 0.01 * np.eye(3))
Q1 = Q[0].copy()     # the synthetic Q is the same 3x3 for every filter
for name, fn in (("host pointers (ukfb_cycle: uploads z, Q)", lambda: e.cycle(0.01, spe.MEAS_POS3, z, Q)),
                 ("host pointers, one Q for the batch (ukfb_cycle_uniform_q: uploads z)", lambda: e.cycle_uniform_q(0.01, spe.MEAS_POS3, z, Q1))):
    fn(); e.sync()
    t0 = time.perf_counter(); k = 5
    for _ in range(k): fn()
    e.sync(); dt = (time.perf_counter() - t0) / k
    print(f"{name}: {dt * 1e3:.2f} ms per cycle -> {n / dt / 1e6:.1f} M filter-cycles/s")
z_t = torch.from_numpy(z).to("cuda", td); Q_t = torch.from_numpy(Q.reshape(-1, 9)).to("cuda", td); torch.cuda.synchronize()
e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t); e.sync()
t0 = time.perf_counter(); k = 50
for _ in range(k): e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t)
e.sync(); dt = (time.perf_counter() - t0) / k
print(f"device pointers (ukfb_cycle_dev): {dt * 1e3:.3f} ms per cycle -> {n / dt / 1e6:.1f} M filter-cycles/s")
# upload only (what a pinned, double-buffered caller would overlap with the previous launch)
zp = torch.from_numpy(z).pin_memory(); Qp = torch.from_numpy(Q.reshape(-1, 9)).pin_memory()
zd = torch.empty_like(z_t, dtype=torch.float64); Qd = torch.empty((n, 9), dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): zd.copy_(zp, non_blocking=True); Qd.copy_(Qp, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"pinned H2D of z + Q alone ({(z.nbytes + Q.nbytes) / 1e6:.0f} MB): {dt * 1e3:.2f} ms = {(z.nbytes + Q.nbytes) / dt / 1e9:.1f} GB/s")
# the same host arrays through a device group (ukfb_group_cycle: every shard uploads and launches its range).  On a one-GPU
# box the shards share device 0 and its PCIe link: the figure shows what the fan-out costs, not what several links give.
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if shards > 0:
    e.close()
    g = spe.UKFGroup(spe.MODEL_POSE, prec, n, [0] * shards)
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        mu, cov = spe.synth.pose_initial(hi - lo, first=lo); g.initialize(mu, cov, first=lo)
    g.set_acceleration(acc, 0.01 * np.eye(3))
    g.cycle(0.01, spe.MEAS_POS3, z, Q); g.sync()
    t0 = time.perf_counter(); k = 5
    for _ in range(k): g.cycle(0.01, spe.MEAS_POS3, z, Q)
    g.sync(); dt = (time.perf_counter() - t0) / k
    print(f"group of {shards} shards on device 0, host pointers (ukfb_group_cycle): {dt * 1e3:.2f} ms per cycle -> {n / dt / 1e6:.1f} M filter-cycles/s")
    g.close()
