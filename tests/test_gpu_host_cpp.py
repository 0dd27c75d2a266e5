"""The host C++ mirror of the reference's class surface (include/pose_estimation/...), driven like a Rock
component would drive PoseUKF / OrientationUKF, against the CPU oracle.  The driver is
tests/cpp/host_classes.cpp (built by __graft_entry__.build() / tests/cpp/Makefile)."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from conftest import max_abs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "build", "host_classes")


def pad(z, Q):
    zz = np.zeros((1, 3)); zz[0, :len(z)] = z
    QQ = np.eye(3)[None].copy(); QQ[0, :len(z), :len(z)] = Q
    return zz, QQ


def test_host_cpp_classes_match_the_oracle(oracle, onp):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert r["pose_ok"] and r["threw_negative"] and r["threw_nonfinite"] and r["state_size"] == 12
    # C++ BatchOrientationUKF latches the constructor's inputs: equal to the scalar class, velocity held steady
    assert r["batch_orient_vs_scalar"] <= 1e-12 and abs(r["batch_orient_dvz"]) < 1e-3
    assert r["batch_cycles_bit_equal"] is True      # BatchUKF::cycles (one launch) == the same samples one cycle() at a time
    # ShardedBatchPoseUKF (ukfb_group_*: one engine per shard) == one BatchPoseUKF over the same filters, bit for bit
    assert r["sharded_orient_bit_equal"] is True
    assert r["sharded_bit_equal"] is True and r["sharded_moved"] > 1e-4 and r["sharded_shard1"] == [0, 4, 3]
    # ---- PoseUKF sequence restated with the oracle
    mu = np.array([[1.0, -2.0, 0.5, 0.0, 0.3826834323650898, 0.0, 0.9238795325112867, 0.3, 0.1, -0.2, 0.05, -0.02, 0.1]])
    i, j = np.meshgrid(np.arange(12), np.arange(12), indexing="ij")
    cov = (np.where(i == j, 0.04, 0.0) + 0.001 / (1.0 + i + j))[None]
    R = np.diag([0.01] * 3 + [0.001] * 3 + [0.00001] * 6)
    m, c, st = oracle.pose_predict(mu, cov, R, None, None, 0.02)
    acc = np.array([[0.2, -0.1, 0.05]])
    m, c, st = oracle.pose_predict(m, c, R, acc, 0.01 * np.eye(3), 0.01)
    z, Q = pad([1.02, -1.97, 0.49], 0.0025 * np.eye(3)); m, c, _ = oracle.pose_update(m, c, onp.MEAS_POS3, z, Q)
    z, Q = pad([0.31, 0.09], 0.01 * np.eye(2)); m, c, _ = oracle.pose_update(m, c, onp.MEAS_XVEL_YAWVEL, z, Q)
    z, Q = pad([0.01, 0.79, -0.02], 0.001 * np.eye(3)); m, c, _ = oracle.pose_update(m, c, onp.MEAS_ORIENT_SO3, z, Q)
    assert max_abs(np.array(r["pose"]["mu"]), m[0]) <= 1e-9
    assert max_abs(np.array(r["pose"]["cov"]).reshape(12, 12), c[0]) <= 1e-9
    # ---- OrientationUKF sequence
    mo = np.array([[0.0, 0.0, 0.13052619222005157, 0.9914448613738104, 0.1, 0.0, -0.05, 1e-4, -2e-4, 5e-5, 1e-3, 2e-3,
                    -1e-3, 9.81]])
    sd = np.array([0.05] * 3 + [0.1] * 3 + [1e-3] * 3 + [1e-2] * 3 + [1e-2])
    co = np.diag(sd ** 2)[None]
    Rn = np.diag([1e-6] * 3 + [1e-4] * 3 + [1e-10] * 3 + [1e-8] * 3 + [1e-12])
    earth = onp.earth_rotation(0.92698121)
    gyro = np.array([[0.01, -0.02, 0.15]]); a = np.array([[0.1, -0.05, 9.79]])
    m2, c2, _ = oracle.orient_predict(mo, co, Rn, a, gyro, 3600.0, 1800.0, earth, 0.01)
    m2, c2, _ = oracle.orient_update(m2, c2, np.array([[0.09, 0.03, -0.04]]), 0.0025 * np.eye(3)[None])
    assert max_abs(np.array(r["orient"]["mu"]), m2[0]) <= 1e-9
    assert max_abs(np.array(r["orient"]["cov"]).reshape(13, 13), c2[0]) <= 1e-9
    assert max_abs(np.array(r["rotation_rate"]), oracle.orient_rotation_rate(m2, gyro, earth)[0]) <= 1e-12
    # GravitationalModel::WGS_84 (GravitationalModel.hpp:33-44)
    lat, alt = 0.92698121, 10.0
    g = 9.7803267714 * ((1 + 0.00193185138639 * math.sin(lat) ** 2) / math.sqrt(1 - 0.0818191908426 ** 2 * math.sin(lat) ** 2))
    g *= (6378137.0 / (6378137.0 + alt)) ** 2
    assert abs(r["wgs84"] - g) < 1e-12


def test_cabi_bench_runs_the_headline_loop_from_cpp():
    """tests/cpp/cabi_bench.cpp: the bench loop (device-resident samples, ukfb_cycle_dev, HIP-event timing) from a C++ host with
    nothing but include/ukf_batch.h -- one engine, and a two-shard ukfb_group on the same device.  Small here; the full-size
    lines are in profiles/ (tools/gpu_bench.sh cabi)."""
    exe = os.path.join(ROOT, "tests", "cpp", "build", "cabi_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    for args, kernel in (["16384", "40", "f64"], "ukf_kernel16<f64,pose,cycle-plain>"), (["16384", "40", "f32"], "ukf_kernel16<f32,pose,cycle-plain>"), \
            (["16387", "40", "f64", "2"], None):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr + out.stdout
        r = json.loads(out.stdout.strip().splitlines()[-1])
        assert r["status_or"] == 0 and r["filter_cycles_per_s"] > 1e7 and r["cycles"] == 40
        if kernel:
            assert r["kernel"] == kernel
