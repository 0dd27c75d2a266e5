"""Deterministic synthetic inputs for the UKF hot path.

Counter-based SplitMix64 so that any consumer (NumPy here, C++ elsewhere) draws the same
numbers from (seed, filter id, slot) without shared state.  All outputs are float64 AoS arrays in
the layout of include/ukf_batch.h (quaternions x, y, z, w).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
SLOTS = 512  # stream index = filter id * SLOTS + slot

SEED_BASE = 0x5EED0000


def splitmix64(x):
    """SplitMix64 output function applied to the 64-bit counters x (vectorised)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, ids, slots, lo=0.0, hi=1.0):
    """U(lo, hi) from the top 53 bits; result shape = broadcast(ids[:, None], slots[None, :])."""
    ids = np.asarray(ids, dtype=np.uint64).reshape(-1, 1)
    slots = np.asarray(slots, dtype=np.uint64).reshape(1, -1)
    with np.errstate(over="ignore"):
        ctr = (ids * np.uint64(SLOTS) + slots) * _GOLDEN + splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF))
    u = (splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def _quat_exp(a):
    """MTK SO3::exp(a) with scale 1 (plain sin/cos form; generator only, not part of parity)."""
    n = np.linalg.norm(a, axis=-1, keepdims=True)
    half = 0.5 * n
    s = np.where(n > 1e-12, np.sin(half) / np.where(n > 1e-12, n, 1.0), 0.5)
    return np.concatenate([a * s, np.cos(half)], axis=-1)


def _spd(seed, ids, dof, diag_std, slot0=16):
    """Sigma0 = D^1/2 (I + 0.1 G G^T / dof) D^1/2 with G ~ U(-1,1)^(dof x dof)."""
    n = len(ids)
    G = uniform(seed, ids, np.arange(slot0, slot0 + dof * dof), -1.0, 1.0).reshape(n, dof, dof)
    A = np.eye(dof)[None] + 0.1 * (G @ np.swapaxes(G, 1, 2)) / float(dof)
    d = np.asarray(diag_std, dtype=np.float64)
    C = d[None, :, None] * A * d[None, None, :]
    return 0.5 * (C + np.swapaxes(C, 1, 2))  # exactly symmetric


# ------------------------------------------------------------------------------- Pose
POSE_DIAG_STD = np.array([0.1] * 3 + [0.05] * 3 + [0.1] * 3 + [0.02] * 3)


def pose_default_process_noise():
    """PoseUKF ctor defaults (PoseUKF.cpp:103-107 of the reference)."""
    return np.diag([0.01] * 3 + [0.001] * 3 + [0.00001] * 3 + [0.00001] * 3)


def pose_initial(n: int, seed: int = SEED_BASE + 2, first: int = 0):
    """mu [n,13], cov [n,12,12]."""
    ids = np.arange(first, first + n)
    p = uniform(seed, ids, [0, 1, 2], -10.0, 10.0)
    q = _quat_exp(uniform(seed, ids, [3, 4, 5], -1.0, 1.0))
    v = uniform(seed, ids, [6, 7, 8], -1.0, 1.0)
    w = uniform(seed, ids, [9, 10, 11], -0.2, 0.2)
    mu = np.concatenate([p, q, v, w], axis=-1)
    cov = _spd(seed, ids, 12, POSE_DIAG_STD)
    return mu, cov


def pose_cycle_inputs(n: int, cycle: int, mu_pos=None, seed: int = SEED_BASE + 2, first: int = 0, random_q=False):
    """Per-cycle inputs: acc [n,3], z [n,3] (= position + noise when mu_pos is given, else U(-10,10)+noise),
    Q [n,3,3] (0.05^2 I, or random SPD when random_q)."""
    ids = np.arange(first, first + n)
    s = seed + 0x10000 * (cycle + 1)
    acc = uniform(s, ids, [0, 1, 2], -0.5, 0.5)
    noise = uniform(s, ids, [3, 4, 5], -0.05, 0.05)
    base = mu_pos if mu_pos is not None else uniform(seed, ids, [0, 1, 2], -10.0, 10.0)
    z = base + noise
    if random_q:
        G = uniform(s, ids, np.arange(6, 15), -1.0, 1.0).reshape(n, 3, 3)
        Q = 0.05 ** 2 * (np.eye(3)[None] + 0.3 * (G @ np.swapaxes(G, 1, 2)) / 3.0)
        Q = 0.5 * (Q + np.swapaxes(Q, 1, 2))
    else:
        Q = np.broadcast_to(0.05 ** 2 * np.eye(3), (n, 3, 3)).copy()
    return acc, z, Q


def pose_mixed_models(n: int, cycle: int, seed: int = SEED_BASE + 5, first: int = 0, inactive_frac=0.25):
    """Config 5: per-filter measurement model id uniform over the 9 Pose models, -1 for inactive."""
    ids = np.arange(first, first + n)
    s = seed + 0x10000 * (cycle + 1)
    m = np.floor(uniform(s, ids, [15], 0.0, 9.0)).astype(np.int32).reshape(-1)
    m = np.clip(m, 0, 8)
    off = uniform(s, ids, [16]).reshape(-1) < inactive_frac
    return np.where(off, -1, m).astype(np.int32)


def pose_measurement_for_model(mu, models, noise):
    """z [n,3] consistent with each filter's model id (first m entries used)."""
    n = mu.shape[0]
    z = np.zeros((n, 3))
    sel = {0: [0, 1, 2], 1: [0, 1], 2: [2], 4: [7, 8, 9], 5: [7, 8], 6: [9], 7: [7, 12], 8: [10, 11, 12]}
    for mid, idx in sel.items():
        i = np.nonzero(models == mid)[0]
        z[np.ix_(i, range(len(idx)))] = mu[np.ix_(i, idx)] + noise[np.ix_(i, range(len(idx)))]
    i = np.nonzero(models == 3)[0]
    if len(i):
        # axis-angle of the current orientation plus noise (log map of a unit quaternion)
        q = mu[i, 3:7]
        nv = np.linalg.norm(q[:, :3], axis=-1, keepdims=True)
        ang = 2.0 * np.arctan2(nv, q[:, 3:4])
        axis = q[:, :3] / np.where(nv > 1e-12, nv, 1.0)
        z[i] = axis * ang + noise[i]
    return z


# ------------------------------------------------------------------------------- Orient
ORIENT_DIAG_STD = np.array([0.05] * 3 + [0.1] * 3 + [0.001] * 3 + [0.01] * 3 + [0.01])
ORIENT_LATITUDE = 0.92698121
ORIENT_TAU = 3600.0
ORIENT_G = 9.81


def orient_process_noise():
    """SURVEY.md section 8(d) config 4 (the reference default is zero)."""
    return np.diag([1e-6] * 3 + [1e-4] * 3 + [1e-10] * 3 + [1e-8] * 3 + [1e-12])


def orient_initial(n: int, seed: int = SEED_BASE + 4, first: int = 0):
    """mu [n,14] = q, v, bias_gyro, bias_acc, gravity; cov [n,13,13]."""
    ids = np.arange(first, first + n)
    q = _quat_exp(uniform(seed, ids, [0, 1, 2], -1.0, 1.0))
    v = uniform(seed, ids, [3, 4, 5], -1.0, 1.0)
    bg = uniform(seed, ids, [6, 7, 8], -1e-3, 1e-3)
    ba = uniform(seed, ids, [9, 10, 11], -1e-2, 1e-2)
    g = np.full((n, 1), ORIENT_G)
    mu = np.concatenate([q, v, bg, ba, g], axis=-1)
    cov = _spd(seed, ids, 13, ORIENT_DIAG_STD, slot0=16)
    return mu, cov


def orient_cycle_inputs(n: int, cycle: int, mu_q=None, seed: int = SEED_BASE + 4, first: int = 0):
    """gyro [n,3], acc [n,3] = q^-1 (0,0,g) + U(-0.1,0.1)^3, z [n,3], Q [n,3,3]."""
    ids = np.arange(first, first + n)
    s = seed + 0x10000 * (cycle + 1)
    gyro = uniform(s, ids, [0, 1, 2], -0.2, 0.2)
    if mu_q is None:
        mu_q = _quat_exp(uniform(seed, ids, [0, 1, 2], -1.0, 1.0))
    # rotate (0,0,g) by q^-1
    qv = -mu_q[:, :3]
    w = mu_q[:, 3:4]
    gvec = np.broadcast_to(np.array([0.0, 0.0, ORIENT_G]), (n, 3))
    uv = 2.0 * np.cross(qv, gvec)
    grav_body = gvec + w * uv + np.cross(qv, uv)
    acc = grav_body + uniform(s, ids, [3, 4, 5], -0.1, 0.1)
    z = uniform(s, ids, [6, 7, 8], -0.05, 0.05)
    Q = np.broadcast_to(0.05 ** 2 * np.eye(3), (n, 3, 3)).copy()
    return gyro, acc, z, Q
