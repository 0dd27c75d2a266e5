import torch, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_pose_estimation_amd as spe
n = 1048576
for prec in (spe.F64, spe.F32):
    e = spe.BatchPoseUKF(n, precision=prec)
    mu, cov = spe.synth.pose_initial(131072)
    mu = np.tile(mu, (8, 1)); cov = np.tile(cov, (8, 1, 1))
    t0 = time.perf_counter(); e.initialize(mu, cov); t1 = time.perf_counter()
    m, c, i = e.state(); t2 = time.perf_counter()
    ok = np.abs(c - cov).max() < (1e-15 if prec == spe.F64 else 1e-7) and np.abs(m - mu).max() < (1e-15 if prec == spe.F64 else 1e-5)
    print("f64" if prec == spe.F64 else "f32", "initialize 1M: %.3f s, state 1M: %.3f s, round trip ok: %s" % (t1 - t0, t2 - t1, ok))
