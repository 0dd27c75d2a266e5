import torch; torch.cuda.init()
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_pose_estimation_amd as spe
from oracle import capi
n = 8
mu, cov = spe.synth.pose_initial(n)
acc, z, Q = spe.synth.pose_cycle_inputs(n, 0, mu[:, :3])
acc_cov = 0.01 * np.eye(3); R = spe.synth.pose_default_process_noise()
m1, c1, _ = capi.pose_predict(mu, cov, R, acc, acc_cov, 0.01)
m2, c2, _ = capi.pose_update(m1, c1, 0, z, Q)
for G in (16, 64):
    def mk():
        e = spe.BatchPoseUKF(n, precision=0, lanes_per_filter=G); e.initialize(mu, cov); e.set_acceleration(acc, acc_cov); return e
    B = mk(); B.cycle(0.01, 0, z, Q); mb, cb, _ = B.state()
    print(G, "cycle err vs oracle", np.abs(mb-m2).max(), np.abs(cb-c2).max(), B.last_launch_info()["kernel"])
    D = mk()
    zt = torch.from_numpy(z).cuda(); Qt = torch.from_numpy(Q.reshape(-1, 9)).cuda(); mt = torch.full((n,), -1, dtype=torch.int32).cuda()
    D.cycle_dev(0.01, 0, zt, Qt, meas_model_dev=mt); md_, cd_, _ = D.state()
    print(G, "cycle(inactive) vs oracle predict", np.abs(md_-m1).max(), np.abs(cd_-c1).max())
