#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the NumPy restatement (oracle/ukf_numpy.py).

PARITY UNPINNED: the reference (rock-slam/slam-pose_estimation) holds no numerical fixture for the UKF
path and cannot be built here (Eigen/boost/MTK/base-types absent), so these vectors come from the
build's own restatement of the algorithm, not from the reference binary.  They pin (a) the C++ oracle
against an independently written implementation and (b) the GPU engine against both.

    python tests/golden/make_golden.py        # rewrites the .npz files (deterministic)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import slam_pose_estimation_amd as spe  # noqa: E402
from oracle import ukf_numpy as un  # noqa: E402


def pose_single_steps(n=12):
    s = spe.synth
    mu, cov = s.pose_initial(n, seed=s.SEED_BASE + 101)
    acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3], seed=s.SEED_BASE + 101, random_q=True)
    R = s.pose_default_process_noise()
    acc_cov = np.array([[0.02, 0.001, 0.0], [0.001, 0.03, -0.002], [0.0, -0.002, 0.01]])
    out = {"mu": mu, "cov": cov, "acc": acc, "acc_cov": acc_cov, "R": R, "Q": Q, "dt": np.array(0.02)}
    out["pred_acc_mu"], out["pred_acc_cov"], _ = un.pose_predict(mu, cov, R, acc, acc_cov, 0.02)
    out["pred_cv_mu"], out["pred_cv_cov"], _ = un.pose_predict(mu, cov, R, None, None, 0.02)
    for model in range(9):
        zz = s.pose_measurement_for_model(mu, np.full(n, model), z - mu[:, :3])
        m, c, st = un.pose_update(mu, cov, model, zz, Q)
        assert (st == 0).all()
        out[f"z_{model}"], out[f"upd_{model}_mu"], out[f"upd_{model}_cov"] = zz, m, c
    return out


def pose_trajectory(n=4, cycles=100):
    """Config 1 shape: IMU-rate predict (acc branch) + 3D position update, 100 cycles."""
    s = spe.synth
    seed = s.SEED_BASE + 1
    mu, cov = s.pose_initial(n, seed=seed)
    R = s.pose_default_process_noise()
    acc_cov = 0.01 * np.eye(3)
    out = {"mu0": mu, "cov0": cov, "R": R, "acc_cov": acc_cov, "dt": np.array(0.01), "cycles": np.array(cycles)}
    accs, zs = [], []
    m, c = mu.copy(), cov.copy()
    for k in range(cycles):
        acc, z, Q = s.pose_cycle_inputs(n, k, m[:, :3], seed=seed)
        m, c, s1 = un.pose_predict(m, c, R, acc, acc_cov, 0.01)
        m, c, s2 = un.pose_update(m, c, un.MEAS_POS3, z, Q)
        assert (s1 == 0).all() and (s2 == 0).all()
        accs.append(acc); zs.append(z)
        if k in (0, 9, 49):
            out[f"mu_{k + 1}"], out[f"cov_{k + 1}"] = m.copy(), c.copy()
    out["acc"], out["z"], out["Q"] = np.stack(accs), np.stack(zs), Q
    out["mu_final"], out["cov_final"] = m, c
    return out


def orient_steps(n=8):
    s = spe.synth
    mu, cov = s.orient_initial(n, seed=s.SEED_BASE + 104)
    gyro, acc, z, Q = s.orient_cycle_inputs(n, 0, mu[:, :4], seed=s.SEED_BASE + 104)
    R = s.orient_process_noise()
    earth = un.earth_rotation(s.ORIENT_LATITUDE)
    out = {"mu": mu, "cov": cov, "gyro": gyro, "acc": acc, "z": z, "Q": Q, "R": R, "earth": earth,
           "tau": np.array(s.ORIENT_TAU), "dt": np.array(0.01)}
    m, c, s1 = un.orient_predict(mu, cov, R, acc, gyro, s.ORIENT_TAU, s.ORIENT_TAU, earth, 0.01)
    out["pred_mu"], out["pred_cov"] = m, c
    m2, c2, s2 = un.orient_update(m, c, z, Q)
    assert (s1 == 0).all() and (s2 == 0).all()
    out["upd_mu"], out["upd_cov"] = m2, c2
    out["rotation_rate"] = un.orient_rotation_rate(m2, gyro, earth)
    return out


def main():
    np.savez(os.path.join(HERE, "pose_steps.npz"), **pose_single_steps())
    np.savez(os.path.join(HERE, "pose_trajectory.npz"), **pose_trajectory())
    np.savez(os.path.join(HERE, "orient_steps.npz"), **orient_steps())
    for f in ("pose_steps.npz", "pose_trajectory.npz", "orient_steps.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
