"""Times predict-only, update-only and fused launches of the bench workload (same device, HIP events)."""
import torch  # noqa: F401
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_pose_estimation_amd as spe
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
for prec in (spe.F64, spe.F32):
    td = torch.float64 if prec == spe.F64 else torch.float32
    e = spe.BatchPoseUKF(n, precision=prec)
    CH = 131072
    z_t = torch.empty((n, 3), dtype=td, device="cuda"); Q_t = torch.empty((n, 9), dtype=td, device="cuda"); a_t = torch.empty((n, 3), dtype=td, device="cuda")
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        mu, cov = spe.synth.pose_initial(hi - lo, first=lo); e.initialize(mu, cov, first=lo)
        acc, z, Q = spe.synth.pose_cycle_inputs(hi - lo, 0, mu[:, :3], first=lo)
        z_t[lo:hi] = torch.from_numpy(z).to("cuda", td); Q_t[lo:hi] = torch.from_numpy(Q.reshape(-1, 9)).to("cuda", td); a_t[lo:hi] = torch.from_numpy(acc).to("cuda", td)
    e.set_acceleration(None, 0.01 * np.eye(3)); e.bind_acceleration_dev(a_t)
    def timeit(fn, k=20):
        for _ in range(5): fn()
        e.sync(); e.timer_begin()
        for _ in range(k): fn()
        return e.timer_end() / k
    tp = timeit(lambda: e.predict(0.01))
    tu = timeit(lambda: e.update_dev(spe.MEAS_POS3, z_t, Q_t))
    tc = timeit(lambda: e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t))
    print("f64" if prec == spe.F64 else "f32", "predict %.3f ms  update %.3f ms  fused %.3f ms  (sum %.3f)  status %d" % (tp, tu, tc, tp + tu, e.status_summary()))
