import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_pose_estimation_amd as spe
for n in (1, 1024):
    mu, cov = spe.synth.pose_initial(n)
    acc, z, Q = spe.synth.pose_cycle_inputs(n, 0, mu[:, :3])
    e = spe.BatchPoseUKF(n); e.initialize(mu, cov); e.set_acceleration(acc, 0.01 * np.eye(3))
    for _ in range(50): e.cycle(0.01, spe.MEAS_POS3, z, Q)
    e.sync(); t0 = time.perf_counter()
    K = 500
    for _ in range(K): e.cycle(0.01, spe.MEAS_POS3, z, Q); e.sync()
    dt = (time.perf_counter() - t0) / K
    print("batch of %d: host-pointer cycle + sync: %.1f us per call" % (n, dt * 1e6))
