"""ukf_numpy.py -- second, independent CPU ORACLE in NumPy float64 (test infrastructure only).

Restates the same algorithm as oracle/ukf_oracle.hpp (MTK/ukfom UKF-on-manifolds as recalled in
SURVEY.md Appendix A + the reference's in-tree models), but written separately and vectorised over
a batch of filters, so that the two restatements check each other (tests/test_oracle_cross.py) and
so that the golden fixtures under tests/golden/ have a generator (tests/golden/make_golden.py).

PARITY UNPINNED: the reference (rock-slam/slam-pose_estimation) cannot be built in this image
(Eigen, boost, MTK, base-types absent) and ships no numerical test of this path, so neither
restatement is checked against real MTK output.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.

Reference lines followed (relative to /root/reference/src):
  pose_with_velocity/PoseUKF.cpp:7-69, 75-97, 112-173, 180-196
  orientation_estimator/OrientationUKF.cpp:12-39, 65-89
  UnscentedKalmanFilter.hpp:83-125
Array conventions are those of include/ukf_batch.h: quaternions (x, y, z, w).
"""
from __future__ import annotations

import numpy as np

ST_OK = 0
ST_SKIPPED_FIRST_TS = 1 << 0
ST_SKIPPED_SMALL_DT = 1 << 1
ST_ERR_NEG_DT = 1 << 2
ST_ERR_DT_TOO_LARGE = 1 << 3
ST_ERR_NONFINITE_MEAS = 1 << 4
ST_ERR_CHOLESKY = 1 << 5
ST_WARN_MEAN_NOCONV = 1 << 6
ST_UNINITIALISED = 1 << 7
ST_INACTIVE = 1 << 8
ST_REJECTED_GATE = 1 << 9

MEAS_POS3, MEAS_POS_XY, MEAS_POS_Z, MEAS_ORIENT_SO3, MEAS_VEL3 = 0, 1, 2, 3, 4
MEAS_VEL_XY, MEAS_VEL_Z, MEAS_XVEL_YAWVEL, MEAS_ANGVEL3, MEAS_ORIENT_BODYVEL3 = 5, 6, 7, 8, 9

EARTHW = (2.0 * np.pi) / 86164.0  # GravitationalModel.hpp:16

_EPS = np.finfo(np.float64).eps
_TAYLOR_N_BOUND = np.sqrt(np.sqrt(_EPS))
_MTK_TOL = 1e-11  # MTK::tolerance<double>()

MEAN_TOL = 1e-6
MEAN_MAX_IT = 10000


# --------------------------------------------------------------------------- SO(3) / quaternions
def cos_sinc_sqrt(x2):
    x2 = np.asarray(x2, dtype=np.float64)
    big = x2 >= _TAYLOR_N_BOUND
    x = np.sqrt(np.where(big, x2, 1.0))
    c_big = np.cos(x)
    s_big = np.sin(x) / x
    inv = (1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.)
    cosi = np.ones_like(x2)
    sinc = np.ones_like(x2)
    term = (-1 / 2.) * x2
    for i in range(3):
        cosi = cosi + term
        term = term * inv[2 * i]
        sinc = sinc + term
        term = term * (-inv[2 * i + 1] * x2)
    return np.where(big, c_big, cosi), np.where(big, s_big, sinc)


def so3_exp(v, scale=1.0):
    v = np.asarray(v, dtype=np.float64)
    s = np.asarray(scale, dtype=np.float64) / 2.0
    n2 = v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1] + v[..., 2] * v[..., 2]
    c, sc = cos_sinc_sqrt(s * s * n2)
    mult = sc * s
    return np.stack([mult * v[..., 0], mult * v[..., 1], mult * v[..., 2], c], axis=-1)


def so3_log(q):
    nv = np.sqrt(q[..., 0] * q[..., 0] + q[..., 1] * q[..., 1] + q[..., 2] * q[..., 2])
    nv = np.where(nv < _MTK_TOL, _MTK_TOL, nv)
    with np.errstate(divide="ignore"):
        s = 2.0 / nv * np.arctan(nv / q[..., 3])
    return q[..., :3] * s[..., None]


def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    w = aw * bw - ax * bx - ay * by - az * bz
    x = aw * bx + ax * bw + ay * bz - az * by
    y = aw * by + ay * bw + az * bx - ax * bz
    z = aw * bz + az * bw + ax * by - ay * bx
    return np.stack([x, y, z, w], axis=-1)


def quat_conj(a):
    return a * np.array([-1.0, -1.0, -1.0, 1.0])


def quat_inverse(a):
    n2 = np.sum(a * a, axis=-1, keepdims=True)
    return quat_conj(a) / n2


def quat_rotate(q, v):
    qv = q[..., :3]
    uv = np.cross(qv, v)
    uv = uv + uv
    return v + q[..., 3:4] * uv + np.cross(qv, uv)


def quat_to_matrix(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - (tyy + tzz); R[..., 0, 1] = txy - twz; R[..., 0, 2] = txz + twy
    R[..., 1, 0] = txy + twz; R[..., 1, 1] = 1 - (txx + tzz); R[..., 1, 2] = tyz - twx
    R[..., 2, 0] = txz - twy; R[..., 2, 1] = tyz + twx; R[..., 2, 2] = 1 - (txx + tyy)
    return R


def so3_boxplus(q, v, scale=1.0):
    return quat_mul(q, so3_exp(v, scale))


def so3_boxminus(q, other):
    return so3_log(quat_mul(quat_conj(other), q))


# --------------------------------------------------------------------------- manifolds
class _Compound:
    """MTK_BUILD_MANIFOLD compound: list of (kind, stored offset, tangent offset, dim)."""

    def __init__(self, fields):
        self.fields = fields
        self.S = sum(4 if k == "so3" else n for k, _, _, n in fields)
        self.D = sum(3 if k == "so3" else n for k, _, _, n in fields)

    def boxplus(self, x, d, scale=1.0):
        out = np.array(np.broadcast_to(x, np.broadcast_shapes(x.shape[:-1], d.shape[:-1]) + (self.S,)))
        for kind, so, to, n in self.fields:
            if kind == "so3":
                out[..., so:so + 4] = so3_boxplus(out[..., so:so + 4], d[..., to:to + 3], scale)
            else:
                out[..., so:so + n] = out[..., so:so + n] + scale * d[..., to:to + n]
        return out

    def boxminus(self, x, y):
        shape = np.broadcast_shapes(x.shape[:-1], y.shape[:-1])
        d = np.empty(shape + (self.D,))
        for kind, so, to, n in self.fields:
            if kind == "so3":
                d[..., to:to + 3] = so3_boxminus(x[..., so:so + 4], y[..., so:so + 4])
            else:
                d[..., to:to + n] = x[..., so:so + n] - y[..., so:so + n]
        return d


# PoseWithVelocity.hpp:18-23
POSE = _Compound([("vec", 0, 0, 3), ("so3", 3, 3, 3), ("vec", 7, 6, 3), ("vec", 10, 9, 3)])
# OrientationState.hpp:20-26
ORIENT = _Compound([("so3", 0, 0, 3), ("vec", 4, 3, 3), ("vec", 7, 6, 3), ("vec", 10, 9, 3), ("vec", 13, 12, 1)])
SO3 = _Compound([("so3", 0, 0, 3)])


def VECT(m):
    return _Compound([("vec", 0, 0, m)])


# --------------------------------------------------------------------------- ukfom::ukf
def cholesky_lower(A):
    """Eigen LLT (unblocked, left-looking).  Returns (L, ok[B])."""
    A = np.asarray(A, dtype=np.float64)
    B, D, _ = A.shape
    L = np.zeros_like(A)
    ok = np.ones(B, dtype=bool)
    for k in range(D):
        x = A[:, k, k].copy()
        for j in range(k):
            x = x - L[:, k, j] * L[:, k, j]
        ok &= x > 0.0
        x = np.sqrt(np.where(x > 0.0, x, 1.0))
        L[:, k, k] = x
        for i in range(k + 1, D):
            a = A[:, i, k].copy()
            for j in range(k):
                a = a - L[:, i, j] * L[:, k, j]
            L[:, i, k] = a / x
    return L, ok


def sigma_points(man, mu, sigma, delta=None):
    """X[B, 2D+1, S]: mu+delta, mu+(delta+L col j), mu+(delta-L col j) interleaved."""
    B, D = mu.shape[0], man.D
    L, ok = cholesky_lower(sigma)
    if delta is None:
        delta = np.zeros((B, D))
    dl = np.empty((B, 2 * D + 1, D))
    dl[:, 0] = delta
    for j in range(D):
        dl[:, 1 + 2 * j] = delta + L[:, :, j]
        dl[:, 2 + 2 * j] = delta - L[:, :, j]
    X = man.boxplus(mu[:, None, :], dl)
    return X, ok


def mean_sigma_points(man, X, tol=MEAN_TOL, max_it=MEAN_MAX_IT):
    B, N, _ = X.shape
    ref = X[:, 0].copy()
    active = np.ones(B, dtype=bool)
    it = np.zeros(B, dtype=np.int64)
    converged = np.ones(B, dtype=bool)
    while active.any():
        d = man.boxminus(X, ref[:, None, :])
        md = np.zeros((B, man.D))
        for i in range(N):
            md = md + d[:, i]
        md = md / float(N)
        norm = np.sqrt(np.sum(md * md, axis=-1))
        new_ref = man.boxplus(ref, md)
        ref = np.where(active[:, None], new_ref, ref)
        big = norm > tol
        it = np.where(active & big, it + 1, it)
        hit_cap = active & big & (it >= max_it)
        converged &= ~hit_cap
        active = active & big & (it < max_it)
    return ref, converged


def cov_sigma_points(man, mean, V):
    d = man.boxminus(V, mean[:, None, :])
    C = np.zeros((V.shape[0], man.D, man.D))
    for i in range(V.shape[1]):
        C = C + d[:, i, :, None] * d[:, i, None, :]
    return 0.5 * C


def cross_cov_sigma_points(manx, manz, meanx, meanz, X, Z):
    dx = manx.boxminus(X, meanx[:, None, :])
    dz = manz.boxminus(Z, meanz[:, None, :])
    C = np.zeros((X.shape[0], manx.D, manz.D))
    for i in range(X.shape[1]):
        C = C + dx[:, i, :, None] * dz[:, i, None, :]
    return 0.5 * C


def ukf_predict(man, mu, sigma, g, R, tol=MEAN_TOL, max_it=MEAN_MAX_IT):
    X, ok = sigma_points(man, mu, sigma)
    XX = g(X)
    m, conv = mean_sigma_points(man, XX, tol, max_it)
    C = cov_sigma_points(man, m, XX) + R
    status = np.where(ok, ST_OK, ST_ERR_CHOLESKY) | np.where(conv, 0, ST_WARN_MEAN_NOCONV)
    mu_out = np.where(ok[:, None], m, mu)
    sig_out = np.where(ok[:, None, None], C, sigma)
    return mu_out, sig_out, status.astype(np.uint32)


def apply_delta(man, mu, sigma, delta):
    X, ok = sigma_points(man, mu, sigma, delta)
    m = X[:, 0]
    return m, cov_sigma_points(man, m, X), ok


def ukf_update(man, manz, mu, sigma, z, h, Q, tol=MEAN_TOL, max_it=MEAN_MAX_IT, gate_chi2=-1.0):
    X, ok = sigma_points(man, mu, sigma)
    Z = h(X)
    mz, conv = mean_sigma_points(manz, Z, tol, max_it)
    S = cov_sigma_points(manz, mz, Z) + Q
    Cxz = cross_cov_sigma_points(man, manz, mu, mz, X, Z)
    Si = np.linalg.inv(S)
    K = Cxz @ Si
    innov = manz.boxminus(z, mz)
    maha = np.einsum("bi,bij,bj->b", innov, Si, innov)
    accept = np.ones_like(ok) if gate_chi2 < 0 else (maha <= gate_chi2)
    sig2 = sigma - (K @ S) @ np.swapaxes(K, 1, 2)
    delta = np.einsum("bij,bj->bi", K, innov)
    sig2s = np.where((ok & accept)[:, None, None], sig2, np.eye(man.D)[None])
    m2, C2, ok2 = apply_delta(man, mu, sig2s, delta)
    good = ok & accept & ok2
    status = np.where(ok & (ok2 | ~accept), ST_OK, ST_ERR_CHOLESKY) | np.where(conv, 0, ST_WARN_MEAN_NOCONV) \
        | np.where(ok & ~accept, ST_REJECTED_GATE, 0)
    return (np.where(good[:, None], m2, mu), np.where(good[:, None, None], C2, sigma),
            status.astype(np.uint32))


# --------------------------------------------------------------------------- time gate
def gate_dt(dt, min_dt=1e-9, max_dt=np.finfo(np.float64).max):
    dt = np.asarray(dt, dtype=np.float64)
    st = np.zeros(dt.shape, dtype=np.uint32)
    st = np.where(dt > max_dt, ST_ERR_DT_TOO_LARGE, st)
    st = np.where(dt <= min_dt, ST_SKIPPED_SMALL_DT, st)
    st = np.where(dt < 0.0, ST_ERR_NEG_DT, st)
    return st.astype(np.uint32)


def gate_timestamps(ts_us, last_us, min_dt=1e-9, max_dt=np.finfo(np.float64).max):
    """UnscentedKalmanFilter.hpp:83-100. Returns (new_last, dt, status)."""
    ts_us = np.asarray(ts_us, dtype=np.int64)
    last_us = np.asarray(last_us, dtype=np.int64)
    first = last_us == 0
    dt = np.where(first, 0.0, (ts_us - last_us).astype(np.float64) / 1000000.0)
    new_last = np.where(first | (dt > min_dt), ts_us, last_us)
    st = np.where(first, ST_SKIPPED_FIRST_TS, gate_dt(dt, min_dt, max_dt))
    return new_last, dt, st.astype(np.uint32)


# --------------------------------------------------------------------------- PoseUKF
def pose_process(X, acc, dt):
    """processModel / processModelWithAcceleration (PoseUKF.cpp:75-97); acc None -> former."""
    X = X.copy()
    dt = np.asarray(dt, dtype=np.float64)  # scalar or [B,1] (one value per filter)
    dtv = dt[..., None] if dt.ndim else dt
    if acc is not None:
        X[..., 7:10] = X[..., 7:10] + dtv * acc
    X[..., 0:3] = X[..., 0:3] + dtv * quat_rotate(X[..., 3:7], X[..., 7:10])
    X[..., 3:7] = so3_boxplus(X[..., 3:7], quat_rotate(X[..., 3:7], X[..., 10:13]), dt)
    return X


def _rotate_block(rot, cov, o):
    return (rot @ cov[:, o:o + 3, o:o + 3]) @ np.swapaxes(rot, 1, 2)


def pose_predict(mu, sigma, process_noise_cov, acc_mu, acc_cov, dt, **kw):
    """PoseUKF::predictionStepImpl (PoseUKF.cpp:180-196) for a batch; dt scalar or [B].
    acc_mu None or [B,3]; rows with a non-finite entry take the constant-velocity branch."""
    B = mu.shape[0]
    dt = np.broadcast_to(np.asarray(dt, dtype=np.float64), (B,))
    Rn = np.broadcast_to(process_noise_cov, (B, 12, 12))
    use_acc = np.zeros(B, dtype=bool) if acc_mu is None else np.all(np.isfinite(acc_mu), axis=-1)
    mu_o, sig_o = mu.copy(), sigma.copy()
    st_o = np.zeros(B, dtype=np.uint32)
    if use_acc.any():
        i = np.nonzero(use_acc)[0]
        R = Rn[i].copy()
        R[:, 6:9, 6:9] = 2.0 * np.broadcast_to(acc_cov, (B, 3, 3))[i]
        a, d = acc_mu[i], dt[i]
        m, s, st = ukf_predict(POSE, mu[i], sigma[i],
                               lambda X: pose_process(X, a[:, None, :], d[:, None]), R, **kw)
        mu_o[i], sig_o[i], st_o[i] = m, s, st
    if (~use_acc).any():
        i = np.nonzero(~use_acc)[0]
        rot = quat_to_matrix(mu[i, 3:7])
        R = Rn[i].copy()
        R[:, 0:3, 0:3] = _rotate_block(rot, Rn[i], 0)
        R[:, 3:6, 3:6] = _rotate_block(rot, Rn[i], 3)
        d = dt[i]
        R = d[:, None, None] * R
        m, s, st = ukf_predict(POSE, mu[i], sigma[i], lambda X: pose_process(X, None, d[:, None]), R, **kw)
        mu_o[i], sig_o[i], st_o[i] = m, s, st
    return mu_o, sig_o, st_o


_POSE_SELECT = {
    MEAS_POS3: [0, 1, 2], MEAS_POS_XY: [0, 1], MEAS_POS_Z: [2],
    MEAS_VEL3: [7, 8, 9], MEAS_VEL_XY: [7, 8], MEAS_VEL_Z: [9],
    MEAS_XVEL_YAWVEL: [7, 12], MEAS_ANGVEL3: [10, 11, 12],
}


def pose_update(mu, sigma, model, z, Q3, **kw):
    """PoseUKF::integrateMeasurement (PoseUKF.cpp:112-173) for a batch sharing one model id.
    z [B,3] (first m used; axis-angle for MEAS_ORIENT_SO3), Q3 [B,3,3] (leading m x m used)."""
    if model == MEAS_ORIENT_SO3:
        zq = so3_exp(z, 1.0)
        return ukf_update(POSE, SO3, mu, sigma, zq, lambda X: X[..., 3:7], Q3, **kw)
    idx = _POSE_SELECT[model]
    m = len(idx)
    return ukf_update(POSE, VECT(m), mu, sigma, z[:, :m], lambda X: X[..., idx], Q3[:, :m, :m], **kw)


def pose_update_mixed(mu, sigma, models, z, Q3, **kw):
    """Per-filter model ids (config 5); negative id = inactive this call."""
    mu_o, sig_o = mu.copy(), sigma.copy()
    st = np.full(mu.shape[0], ST_INACTIVE, dtype=np.uint32)
    for mid in np.unique(models):
        if mid < 0:
            continue
        i = np.nonzero(models == mid)[0]
        mu_o[i], sig_o[i], st[i] = pose_update(mu[i], sigma[i], int(mid), z[i], Q3[i], **kw)
    return mu_o, sig_o, st


# --------------------------------------------------------------------------- OrientationUKF
def orient_process(X, acc, omega, tau_g, tau_a, earth, dt):
    """processModel (OrientationUKF.cpp:12-32)."""
    X = X.copy()
    dt = np.asarray(dt, dtype=np.float64)  # scalar or [B,1]
    dtv = dt[..., None] if dt.ndim else dt
    av = quat_rotate(X[..., 0:4], omega - X[..., 7:10]) - earth
    X[..., 0:4] = so3_boxplus(X[..., 0:4], av, dt)
    a = quat_rotate(X[..., 0:4], acc - X[..., 10:13])
    a[..., 2] = a[..., 2] - X[..., 13]
    X[..., 4:7] = X[..., 4:7] + dtv * a
    X[..., 7:10] = X[..., 7:10] + dtv * ((-1.0 / tau_g) * X[..., 7:10])
    X[..., 10:13] = X[..., 10:13] + dtv * ((-1.0 / tau_a) * X[..., 10:13])
    return X


def earth_rotation(latitude):
    return np.array([EARTHW * np.cos(latitude), 0.0, EARTHW * np.sin(latitude)])  # OrientationUKF.cpp:47


def orient_predict(mu, sigma, process_noise_cov, acc, omega, tau_g, tau_a, earth, dt, **kw):
    """OrientationUKF::predictionStepImpl (OrientationUKF.cpp:79-89)."""
    B = mu.shape[0]
    dt = np.broadcast_to(np.asarray(dt, dtype=np.float64), (B,))
    Rn = np.broadcast_to(process_noise_cov, (B, 13, 13))
    rot = quat_to_matrix(mu[:, 0:4])
    R = Rn.copy()
    R[:, 0:3, 0:3] = _rotate_block(rot, Rn, 0)
    R[:, 3:6, 3:6] = _rotate_block(rot, Rn, 3)
    R = np.power(dt, 2.0)[:, None, None] * R
    g = lambda X: orient_process(X, acc[:, None, :], omega[:, None, :], tau_g, tau_a, earth, dt[:, None])
    return ukf_predict(ORIENT, mu, sigma, g, R, **kw)


def orient_update(mu, sigma, z, Q3, **kw):
    """integrateMeasurement(VelocityMeasurement) (OrientationUKF.cpp:65-72), h = q^-1 v (:34-39)."""
    h = lambda X: quat_rotate(quat_inverse(X[..., 0:4]), X[..., 4:7])
    return ukf_update(ORIENT, VECT(3), mu, sigma, z, h, Q3, **kw)


def orient_rotation_rate(mu, omega, earth):
    """getRotationRate (OrientationUKF.cpp:74-77)."""
    return omega - mu[:, 7:10] - quat_rotate(quat_inverse(mu[:, 0:4]), earth)


# --------------------------------------------------------------------------- BodyStateMeasurement
def body_state_export(mu, cov):
    """toRigidBodyState (BodyStateMeasurement.hpp:28-39) -> [B, 49] records (layout: include/ukf_batch.h)."""
    B = mu.shape[0]
    out = np.empty((B, 49))
    out[:, 0:7] = mu[:, 0:7]
    out[:, 7:10] = quat_rotate(mu[:, 3:7], mu[:, 7:10])     # velocity = orientation * velocity (:32)
    out[:, 10:13] = mu[:, 10:13]
    for b in range(4):
        out[:, 13 + 9 * b:22 + 9 * b] = cov[:, 3 * b:3 * b + 3, 3 * b:3 * b + 3].reshape(B, 9)
    return out


def body_state_import(rec):
    """fromRigidBodyState (BodyStateMeasurement.hpp:14-26) -> (mu [B,13], cov [B,12,12])."""
    B = rec.shape[0]
    mu = rec[:, 0:13].copy()
    cov = np.zeros((B, 12, 12))
    for b in range(4):
        cov[:, 3 * b:3 * b + 3, 3 * b:3 * b + 3] = rec[:, 13 + 9 * b:22 + 9 * b].reshape(B, 3, 3)
    return mu, cov
