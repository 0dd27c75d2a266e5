"""Diagnostic: where does the wall time of a short timed region go (launch loop / event sync / device sync)?"""
import torch  # noqa: F401
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_pose_estimation_amd as spe
n = 1048576
prec = spe.F32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else spe.F64
td = torch.float32 if prec == spe.F32 else torch.float64
e = spe.BatchPoseUKF(n, precision=prec)
CH = 131072
z_t = torch.empty((n, 3), dtype=td, device="cuda"); Q_t = torch.empty((n, 9), dtype=td, device="cuda"); a_t = torch.empty((n, 3), dtype=td, device="cuda")
for lo in range(0, n, CH):
    hi = min(n, lo + CH)
    mu, cov = spe.synth.pose_initial(hi - lo, first=lo); e.initialize(mu, cov, first=lo)
    acc, z, Q = spe.synth.pose_cycle_inputs(hi - lo, 0, mu[:, :3], first=lo)
    z_t[lo:hi] = torch.from_numpy(z).to("cuda", td); Q_t[lo:hi] = torch.from_numpy(Q.reshape(-1, 9)).to("cuda", td); a_t[lo:hi] = torch.from_numpy(acc).to("cuda", td)
e.set_acceleration(None, 0.01 * np.eye(3)); e.bind_acceleration_dev(a_t)
torch.cuda.synchronize()
for rep in range(8):
    for _ in range(5): e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t)
    torch.cuda.synchronize()
    e.timer_begin(); t0 = time.perf_counter()
    for _ in range(30): e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t)
    t1 = time.perf_counter()
    k = e.timer_end(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("launch loop %.2f ms   event sync %.2f ms   device sync %.2f ms   gpu %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, k))
