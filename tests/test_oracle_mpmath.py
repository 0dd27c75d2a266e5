"""The CPU oracle against an ARBITRARY-PRECISION evaluation of the same algorithm (mpmath, 40 digits).

This pins the oracle's floating-point arithmetic: closed-form sin / cos / atan2-free log, the Cholesky, the
iterated manifold mean, the 1/2-weighted covariance, gain and downdate must reproduce a 40-digit evaluation of
SURVEY.md Appendix A to ~1e-13.  It cannot pin the SEMANTICS against real MTK (parity unpinned, DESIGN.md
section 3): the statements evaluated here are the same recalled ones.  Reference call sites restated:
PoseUKF.cpp:75-97 (process models), :112-117 (position update), :180-196 (noise shaping);
OrientationUKF.cpp:12-32 (process model), :34-39,65-72 (body-velocity update), :79-89 (noise shaping).
UnscentedKalmanFilter.hpp:107-125 is not exercised (plain dt)."""
import numpy as np
import pytest

mp = pytest.importorskip("mpmath")

from conftest import max_abs

mp.mp.dps = 40

# manifold = list of (kind, stored offset, tangent offset, length)
POSE = [("vec", 0, 0, 3), ("so3", 3, 3, 3), ("vec", 7, 6, 3), ("vec", 10, 9, 3)]        # PoseWithVelocity.hpp:18-23
ORIENT = [("so3", 0, 0, 3), ("vec", 4, 3, 3), ("vec", 7, 6, 3), ("vec", 10, 9, 3), ("vec", 13, 12, 1)]  # OrientationState.hpp:20-26


def dof(man):
    return sum(f[3] for f in man)


def qmul(a, b):  # Eigen quaternion product, storage x y z w
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by,
            aw * by + ay * bw + az * bx - ax * bz,
            aw * bz + az * bw + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz]


def qrot(q, v):  # rotation of v by the (unit) quaternion q
    x, y, z, w = q
    r = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    return [sum(r[i][k] * v[k] for k in range(3)) for i in range(3)], r


def so3_exp(v, scale=mp.mpf(1)):  # MTK::SO3::exp: w = cos(|v| s/2), vec = sinc(|v| s/2) (s/2) v
    n = mp.sqrt(sum(c * c for c in v))
    h = scale / 2
    a = n * h
    sinc = mp.sin(a) / a if a != 0 else mp.mpf(1)
    return [sinc * h * c for c in v] + [mp.cos(a)]


def so3_log(q):  # 2 atan(|vec| / w) / |vec| * vec, plain atan (plus-minus periodicity)
    nv = mp.sqrt(q[0] ** 2 + q[1] ** 2 + q[2] ** 2)
    if nv == 0:
        return [mp.mpf(0)] * 3
    s = 2 * mp.atan(nv / q[3]) / nv
    return [s * q[k] for k in range(3)]


def boxplus(man, x, d):
    out = list(x)
    for kind, so, to, n in man:
        if kind == "so3":
            out[so:so + 4] = qmul(x[so:so + 4], so3_exp(d[to:to + 3]))   # right multiplication
        else:
            for k in range(n):
                out[so + k] = x[so + k] + d[to + k]
    return out


def boxminus(man, x, y):
    out = [mp.mpf(0)] * dof(man)
    for kind, so, to, n in man:
        if kind == "so3":
            yc = [-y[so], -y[so + 1], -y[so + 2], y[so + 3]]
            out[to:to + 3] = so3_log(qmul(yc, x[so:so + 4]))
        else:
            for k in range(n):
                out[to + k] = x[so + k] - y[so + k]
    return out


def pose_process(x, acc, dt):  # processModelWithAcceleration / processModel, PoseUKF.cpp:75-97
    x = list(x)
    if acc is not None:
        for k in range(3):
            x[7 + k] += acc[k] * dt
    q = x[3:7]
    rv, _ = qrot(q, x[7:10])
    for k in range(3):
        x[k] += rv[k] * dt
    rw, _ = qrot(q, x[10:13])
    x[3:7] = qmul(q, so3_exp(rw, dt))
    return x


def orient_process(x, acc, omega, tau_g, tau_a, earth, dt):  # OrientationUKF.cpp:12-32
    x = list(x)
    av, _ = qrot(x[0:4], [omega[k] - x[7 + k] for k in range(3)])
    av = [av[k] - earth[k] for k in range(3)]
    x[0:4] = qmul(x[0:4], so3_exp(av, dt))
    a, _ = qrot(x[0:4], [acc[k] - x[10 + k] for k in range(3)])      # the UPDATED orientation (:22)
    a[2] -= x[13]
    for k in range(3):
        x[4 + k] += a[k] * dt
    for k in range(3):
        x[7 + k] += (-1 / tau_g) * x[7 + k] * dt
        x[10 + k] += (-1 / tau_a) * x[10 + k] * dt
    return x


def chol(A):
    n = A.rows
    L = mp.matrix(n, n)
    for j in range(n):
        s = A[j, j] - sum(L[j, k] ** 2 for k in range(j))
        L[j, j] = mp.sqrt(s)
        for i in range(j + 1, n):
            L[i, j] = (A[i, j] - sum(L[i, k] * L[j, k] for k in range(j))) / L[j, j]
    return L


def sigma_points(man, mu, Sig, delta=None):
    D = dof(man)
    L = chol(Sig)
    d0 = [mp.mpf(0)] * D if delta is None else delta
    X = [boxplus(man, mu, d0)]
    for j in range(D):
        X.append(boxplus(man, mu, [d0[i] + L[i, j] for i in range(D)]))
        X.append(boxplus(man, mu, [d0[i] - L[i, j] for i in range(D)]))
    return X


def manifold_mean(man, X, tol):
    D = dof(man)
    ref = list(X[0])
    for _ in range(100):
        d = [sum(boxminus(man, x, ref)[k] for x in X) / len(X) for k in range(D)]
        ref = boxplus(man, ref, d)
        if mp.sqrt(sum(c * c for c in d)) <= tol:
            break
    return ref


def half_cov(man, X, ref):
    D = dof(man)
    C = mp.matrix(D, D)
    for x in X:
        d = boxminus(man, x, ref)
        for a in range(D):
            for b in range(D):
                C[a, b] += d[a] * d[b] / 2
    return C


def rotate_blocks(R, rot, offsets):  # R with the 3x3 diagonal blocks at `offsets` replaced by rot * block * rot^T
    Rm = mp.matrix(3, 3)
    for i in range(3):
        for k in range(3):
            Rm[i, k] = rot[i][k]
    out = R.copy()
    for o in offsets:
        blk = mp.matrix(3, 3)
        for i in range(3):
            for k in range(3):
                blk[i, k] = R[o + i, o + k]
        blk = Rm * blk * Rm.T
        for i in range(3):
            for k in range(3):
                out[o + i, o + k] = blk[i, k]
    return out


def mp_predict(man, mu, Sig, g, Rn, tol):  # ukfom::ukf::predict, SURVEY.md Appendix A.3
    X = [g(x) for x in sigma_points(man, mu, Sig)]
    ref = manifold_mean(man, X, tol)
    return ref, half_cov(man, X, ref) + Rn


def mp_update(man, mu, Sig, z, Q, h, tol):  # ukfom::ukf::update with a vector measurement, Appendix A.4
    D = dof(man)
    X = sigma_points(man, mu, Sig)
    Z = [h(x) for x in X]
    zbar = [sum(zz[k] for zz in Z) / len(Z) for k in range(3)]
    Sm = mp.matrix(3, 3)
    Cxz = mp.matrix(D, 3)
    for x, zz in zip(X, Z):
        dz = [zz[k] - zbar[k] for k in range(3)]
        dx = boxminus(man, x, mu)
        for a in range(3):
            for b in range(3):
                Sm[a, b] += dz[a] * dz[b] / 2
        for a in range(D):
            for b in range(3):
                Cxz[a, b] += dx[a] * dz[b] / 2
    Sm = Sm + Q
    K = Cxz * (Sm ** -1)
    delta = K * mp.matrix([z[k] - zbar[k] for k in range(3)])
    Sig2 = Sig - K * Sm * K.T
    Xn = sigma_points(man, mu, Sig2, [delta[k] for k in range(D)])   # applyDelta
    return Xn[0], half_cov(man, Xn, Xn[0])


def to_mp(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return [mp.mpf(float(v)) for v in a]
    m = mp.matrix(a.shape[0], a.shape[1])
    for i in range(a.shape[0]):
        for j in range(a.shape[1]):
            m[i, j] = mp.mpf(float(a[i, j]))
    return m


def to_np(m):
    if isinstance(m, list):
        return np.array([float(v) for v in m])
    return np.array([[float(m[i, j]) for j in range(m.cols)] for i in range(m.rows)])


@pytest.mark.parametrize("use_acc", [True, False])
def test_pose_predict_and_position_update_against_40_digits(spe, oracle, use_acc):
    n = 3
    mu, cov = spe.synth.pose_initial(n)
    acc, z, Q = spe.synth.pose_cycle_inputs(n, 0, mu[:, :3])
    R = spe.synth.pose_default_process_noise()
    acc_cov = 0.01 * np.eye(3)
    dt = 0.01
    tol = mp.mpf(float(oracle.default_config().mean_tol))
    m_p, c_p, st = oracle.pose_predict(mu, cov, R, acc if use_acc else None, acc_cov if use_acc else None, dt)
    assert (st == 0).all()
    m_u, c_u, st = oracle.pose_update(m_p, c_p, spe.MEAS_POS3, z, Q)
    assert (st == 0).all()
    for i in range(n):
        mui = to_mp(mu[i])
        if use_acc:   # PoseUKF.cpp:188-192: raw noise, velocity block = 2 acc.cov, NOT scaled by dt
            Rn = to_mp(R)
            for a in range(3):
                for b in range(3):
                    Rn[6 + a, 6 + b] = 2 * mp.mpf(float(acc_cov[a, b]))
            acci = to_mp(acc[i])
        else:         # PoseUKF.cpp:182-187,194: position / orientation blocks rotated by the mean, times dt
            _, rot = qrot(mui[3:7], [mp.mpf(0)] * 3)
            Rn = rotate_blocks(to_mp(R), rot, (0, 3)) * mp.mpf(dt)
            acci = None
        mm, cc = mp_predict(POSE, mui, to_mp(cov[i]), lambda x: pose_process(x, acci, mp.mpf(dt)), Rn, tol)
        assert max_abs(to_np(mm), m_p[i]) < 1e-13 and max_abs(to_np(cc), c_p[i]) < 1e-13
        # the update is evaluated from the ORACLE's predicted state, so only the update's rounding is measured
        mu2, cu2 = mp_update(POSE, to_mp(m_p[i]), to_mp(0.5 * (c_p[i] + c_p[i].T)), to_mp(z[i]), to_mp(Q[i]),
                             lambda x: x[0:3], tol)   # measurementPosition, PoseUKF.cpp:7-11
        assert max_abs(to_np(mu2), m_u[i]) < 1e-13 and max_abs(to_np(cu2), c_u[i]) < 1e-13


def test_orientation_predict_and_velocity_update_against_40_digits(spe, oracle):
    s = spe.synth
    n = 2
    mu, cov = s.orient_initial(n)
    gyro, acc, z, Q = s.orient_cycle_inputs(n, 0, mu)
    R = s.orient_process_noise()
    dt = 0.01
    earth = np.array([oracle.earthw() * np.cos(s.ORIENT_LATITUDE), 0.0, oracle.earthw() * np.sin(s.ORIENT_LATITUDE)])
    tol = mp.mpf(float(oracle.default_config().mean_tol))
    m_p, c_p, st = oracle.orient_predict(mu, cov, R, acc, gyro, s.ORIENT_TAU, s.ORIENT_TAU, earth, dt)
    assert (st == 0).all()
    m_u, c_u, st = oracle.orient_update(m_p, c_p, z, Q)
    assert (st == 0).all()
    for i in range(n):
        mui = to_mp(mu[i])
        _, rot = qrot(mui[0:4], [mp.mpf(0)] * 3)
        Rn = rotate_blocks(to_mp(R), rot, (0, 3)) * mp.mpf(dt) ** 2     # OrientationUKF.cpp:81-86: times delta^2
        g = lambda x: orient_process(x, to_mp(acc[i]), to_mp(gyro[i]), mp.mpf(s.ORIENT_TAU), mp.mpf(s.ORIENT_TAU),
                                     to_mp(earth), mp.mpf(dt))
        mm, cc = mp_predict(ORIENT, mui, to_mp(cov[i]), g, Rn, tol)
        assert max_abs(to_np(mm), m_p[i]) < 1e-13 and max_abs(to_np(cc), c_p[i]) < 1e-13

        def h(x):   # velocityMeasurementModel: orientation.inverse() * velocity, OrientationUKF.cpp:34-39
            qi = [-x[0], -x[1], -x[2], x[3]]
            return qrot(qi, x[4:7])[0]
        mu2, cu2 = mp_update(ORIENT, to_mp(m_p[i]), to_mp(0.5 * (c_p[i] + c_p[i].T)), to_mp(z[i]), to_mp(Q[i]), h, tol)
        assert max_abs(to_np(mu2), m_u[i]) < 1e-13 and max_abs(to_np(cu2), c_u[i]) < 1e-13


def mp_update_so3(man, mu, Sig, zq, Q, h, tol):
    """ukfom::ukf::update with an SO(3)-valued measurement (iterated mean of Z, deltas through log)."""
    D = dof(man)
    so3 = [("so3", 0, 0, 3)]
    X = sigma_points(man, mu, Sig)
    Z = [h(x) for x in X]
    zbar = manifold_mean(so3, Z, tol)
    Sm = mp.matrix(3, 3)
    Cxz = mp.matrix(D, 3)
    for x, zz in zip(X, Z):
        dz = boxminus(so3, zz, zbar)
        dx = boxminus(man, x, mu)
        for a in range(3):
            for b in range(3):
                Sm[a, b] += dz[a] * dz[b] / 2
        for a in range(D):
            for b in range(3):
                Cxz[a, b] += dx[a] * dz[b] / 2
    Sm = Sm + Q
    K = Cxz * (Sm ** -1)
    delta = K * mp.matrix(boxminus(so3, zq, zbar))
    Sig2 = Sig - K * Sm * K.T
    Xn = sigma_points(man, mu, Sig2, [delta[k] for k in range(D)])
    return Xn[0], half_cov(man, Xn, Xn[0])


def test_pose_orientation_update_on_so3_against_40_digits(spe, oracle):
    """integrateMeasurement(OrientationMeasurement): z = SO3::exp(mu) (PoseUKF.cpp:133-138), h = orientation
    (PoseUKF.cpp:28-33); the measurement lives on SO(3)."""
    n = 3
    rng = np.random.default_rng(5)
    mu, cov = spe.synth.pose_initial(n)
    Q = np.stack([np.eye(3) * 0.01] * n)
    tol = mp.mpf(float(oracle.default_config().mean_tol))
    # axis-angle measurements near the state's orientation
    z = np.stack([oracle.so3_log(mu[i, 3:7]) + rng.uniform(-0.05, 0.05, 3) for i in range(n)])
    m_u, c_u, st = oracle.pose_update(mu, cov, spe.MEAS_ORIENT_SO3, z, Q)
    assert (st == 0).all()
    for i in range(n):
        zq = so3_exp(to_mp(z[i]))
        mu2, cu2 = mp_update_so3(POSE, to_mp(mu[i]), to_mp(cov[i]), zq, to_mp(Q[i]), lambda x: x[3:7], tol)
        assert max_abs(to_np(mu2), m_u[i]) < 1e-13 and max_abs(to_np(cu2), c_u[i]) < 1e-13


def test_rotation_delta_rebase_series_against_40_digits():
    """The engine's final rotation deltas (ukf_device.hpp so3_rebase_small): log(exp(-a) exp(d)) by one Euler step of the
    inverse left Jacobian (exact in the delta d) plus the leading second-order term in the last, sub-tolerance move a of the
    mean.  Bounds the kernel guarantees before it uses the formula: |d|^2 <= 2.25, |a|^2 <= 1e-12.  Storage order x y z w as
    everywhere in this file."""
    rng = np.random.default_rng(11)
    coef = [1 / 12., 1 / 720., 1 / 30240., 1 / 1209600., 1 / 47900160., 691 / 1307674368000.]
    worst = 0.0
    for trial in range(400):
        d = rng.uniform(-1, 1, 3)
        d *= rng.uniform(0.0, 1.5) / np.linalg.norm(d)
        a = rng.uniform(-1, 1, 3)
        a *= 10.0 ** rng.uniform(-9, -6) / np.linalg.norm(a)
        exact = so3_log(qmul(so3_exp([mp.mpf(-x) for x in a]), so3_exp([mp.mpf(x) for x in d])))
        t, da, a2 = float(d @ d), float(d @ a), float(a @ a)
        c = 0.0
        for k in reversed(coef):
            c = c * t + k
        shift = -d * (c * da + a2 / 12) - a * (1 - c * t - da / 12) - 0.5 * np.cross(a, d)   # approx - d
        worst = max(worst, max(abs(float(exact[k] - mp.mpf(d[k])) - shift[k]) for k in range(3)))
    assert worst < 4e-14, worst
