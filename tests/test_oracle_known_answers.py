"""Known-answer tests that pin the CPU oracle.  The reference holds no test or
fixture for the UKF path, so the oracle is pinned by closed-form answers: exp/log round trips, the
boxplus/boxminus inverse property, identity process model => Sigma + R, linear measurement => exact
Kalman update, and the reference's own quirks (PoseUKF.cpp:188-192, OrientationUKF.cpp:86)."""
import numpy as np
import pytest

from conftest import max_abs


def rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def spd(rng, d, scale):
    g = rng.uniform(-1, 1, (d, d))
    a = np.eye(d) + 0.2 * g @ g.T / d
    s = np.asarray(scale)
    return 0.5 * ((s[:, None] * a * s[None, :]) + (s[:, None] * a * s[None, :]).T)


def cov_static(rng):
    """Covariance whose velocity / angular-velocity spread is negligible (1e-24) and uncorrelated, so that
    the sigma points of a zero-rate state stay put under the process model (an exact identity model)."""
    c = np.zeros((12, 12))
    c[:6, :6] = spd(rng, 6, [0.1] * 3 + [0.05] * 3)
    c[6:, 6:] = 1e-24 * np.eye(6)
    return c


def test_so3_exp_log_round_trip(oracle):
    rng = np.random.default_rng(1)
    for scale in (1e-9, 1e-5, 1e-3, 0.02, 0.3, 1.5, 3.0):
        v = rng.normal(size=3)
        v = v / np.linalg.norm(v) * scale
        q = oracle.so3_exp(v)
        assert abs(np.linalg.norm(q) - 1.0) < 1e-14
        assert max_abs(oracle.so3_log(q), v) < 1e-13
    # Taylor / closed-form switch of MTK's cos_sinc_sqrt is continuous at eps^(1/4)
    b = np.sqrt(np.sqrt(np.finfo(float).eps))
    lo = oracle.so3_exp(np.array([2 * np.sqrt(b) * (1 - 1e-9), 0, 0]))
    hi = oracle.so3_exp(np.array([2 * np.sqrt(b) * (1 + 1e-9), 0, 0]))
    assert max_abs(lo, hi) < 1e-10


def test_so3_exp_matches_axis_angle_formula(oracle):
    v = np.array([0.3, -0.2, 0.5])
    a = np.linalg.norm(v)
    q = oracle.so3_exp(v)
    assert max_abs(q, np.concatenate([np.sin(a / 2) * v / a, [np.cos(a / 2)]])) < 1e-15
    # scale argument: exp(v, s) == exp(s v)   (boxplus(vec, scale) as used in PoseUKF.cpp:80-81)
    assert max_abs(oracle.so3_exp(v, 0.25), oracle.so3_exp(0.25 * v)) < 1e-15


def test_log_plus_minus_periodicity(oracle):
    """MTK::log with plus_minus_periodicity: q and -q give the same vector (plain atan)."""
    q = oracle.so3_exp(np.array([0.4, 0.1, -0.3]))
    assert max_abs(oracle.so3_log(q), oracle.so3_log(-q)) < 1e-15


def test_quaternion_rotation_matches_matrix(oracle):
    rng = np.random.default_rng(2)
    q, v = rand_quat(rng), rng.normal(size=3)
    assert max_abs(oracle.quat_rotate(q, v), oracle.quat_to_matrix(q) @ v) < 1e-14
    R = oracle.quat_to_matrix(q)
    assert max_abs(R @ R.T, np.eye(3)) < 1e-14 and abs(np.linalg.det(R) - 1) < 1e-14


def test_boxplus_boxminus_inverse(oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.normal(size=3), rand_quat(rng), rng.normal(size=6)])
    d = rng.normal(size=12) * 0.3
    assert max_abs(oracle.pose_boxminus(oracle.pose_boxplus(x, d), x), d) < 1e-13
    xo = np.concatenate([rand_quat(rng), rng.normal(size=10)])
    do = rng.normal(size=13) * 0.3
    assert max_abs(oracle.orient_boxminus(oracle.orient_boxplus(xo, do), xo), do) < 1e-13
    # right multiplication: x [+] d rotates about the BODY axis
    q = rand_quat(rng)
    x2 = np.concatenate([np.zeros(3), q, np.zeros(6)])
    d2 = np.zeros(12); d2[3:6] = [0.2, 0, 0]
    e = oracle.so3_exp(d2[3:6])
    from oracle import ukf_numpy as un
    assert max_abs(oracle.pose_boxplus(x2, d2)[3:7], un.quat_mul(q, e)) < 1e-15


def test_cholesky_reconstructs(oracle):
    rng = np.random.default_rng(4)
    A = spd(rng, 12, np.full(12, 0.3))
    L, ok = oracle.cholesky12(A)
    assert ok and max_abs(L @ L.T, A) < 1e-15 and np.allclose(L, np.tril(L))
    bad = A.copy(); bad[5, 5] = -1.0
    assert not oracle.cholesky12(bad)[1]


def test_identity_process_model_adds_process_noise_exactly(oracle):
    """dt -> tiny and zero rates: processModel is the identity, so Sigma' = Sigma + dt R_rot, mu' = mu.
    This is the consistency check of the 1/2 weight with un-scaled Cholesky sigma points (Appendix A.3)."""
    rng = np.random.default_rng(5)
    n = 6
    mu = np.stack([np.concatenate([rng.normal(size=3), rand_quat(rng), np.zeros(6)]) for _ in range(n)])
    cov = np.stack([cov_static(rng) for _ in range(n)])
    R = np.diag([0.01] * 3 + [0.001] * 3 + [1e-5] * 6)   # isotropic blocks: rotation leaves them unchanged
    dt = 0.5
    m, c, st = oracle.pose_predict(mu, cov, R, None, None, dt)
    assert (st == 0).all()
    assert max_abs(m, mu) < 1e-12
    assert max_abs(c, cov + dt * R) < 1e-12


def test_acceleration_branch_noise_is_neither_rotated_nor_scaled(oracle):
    """PoseUKF.cpp:188-192: with a finite acceleration the shadowing process_noise is the raw
    process_noise_cov with block(6,6,3,3) = 2 acc.cov -- no rot * R * rot^T, no delta scaling."""
    rng = np.random.default_rng(6)
    n = 4
    mu = np.stack([np.concatenate([rng.normal(size=3), rand_quat(rng), np.zeros(6)]) for _ in range(n)])
    cov = np.stack([cov_static(rng) for _ in range(n)])
    R = spd(rng, 12, np.full(12, 0.05))          # anisotropic: a rotation would change the blocks
    acc_cov = spd(rng, 3, np.full(3, 0.2))
    dt = 1e-3
    m, c, st = oracle.pose_predict(mu, cov, R, np.zeros((n, 3)), acc_cov, dt)
    expect = R.copy(); expect[6:9, 6:9] = 2.0 * acc_cov
    # zero velocity / rates and zero acceleration: the model is the identity, only the noise differs
    assert max_abs(c, cov + expect) < 1e-10
    m2, c2, _ = oracle.pose_predict(mu, cov, R, None, None, dt)   # constant-velocity branch
    rot = np.stack([oracle.quat_to_matrix(q) for q in mu[:, 3:7]])
    Rr = np.broadcast_to(R, (n, 12, 12)).copy()
    Rr[:, 0:3, 0:3] = rot @ R[0:3, 0:3] @ np.swapaxes(rot, 1, 2)
    Rr[:, 3:6, 3:6] = rot @ R[3:6, 3:6] @ np.swapaxes(rot, 1, 2)
    assert max_abs(c2, cov + dt * Rr) < 1e-10


def test_orientation_filter_noise_scales_with_dt_squared(oracle):
    """OrientationUKF.cpp:86: process_noise = pow(delta, 2.) * process_noise."""
    rng = np.random.default_rng(7)
    n = 3
    mu = np.stack([np.concatenate([rand_quat(rng), np.zeros(3), np.zeros(3), np.zeros(3), [0.0]]) for _ in range(n)])
    def cov_o():
        c = np.zeros((13, 13))
        c[:6, :6] = spd(rng, 6, np.full(6, 0.05))   # orientation, velocity
        c[6:, 6:] = 1e-24 * np.eye(7)               # biases and gravity: negligible spread
        return c
    cov = np.stack([cov_o() for _ in range(n)])
    R = np.diag([1e-2] * 3 + [1e-2] * 3 + [1e-3] * 7)
    dt = 0.1
    z3 = np.zeros((n, 3))
    # zero inputs, zero biases, zero gravity, no earth rotation, huge taus: identity model
    m, c, st = oracle.orient_predict(mu, cov, R, z3, z3, 1e30, 1e30, np.zeros(3), dt)
    assert (st == 0).all() and max_abs(m, mu) < 1e-12
    assert max_abs(c, cov + dt ** 2 * R) < 1e-12


def test_position_update_equals_linear_kalman_on_euclidean_block(oracle):
    """A PositionMeasurement is linear in the state, so the unscented update must reproduce the exact
    Kalman update for the mean of the Euclidean components and for the whole covariance except the
    re-sampling of the SO(3) rows in applyDelta (second order in the orientation correction)."""
    rng = np.random.default_rng(8)
    n = 5
    mu = np.stack([np.concatenate([rng.normal(size=3), rand_quat(rng), rng.normal(size=6)]) for _ in range(n)])
    cov = np.stack([spd(rng, 12, [0.1] * 3 + [0.01] * 3 + [0.1] * 3 + [0.02] * 3) for _ in range(n)])
    Q = np.stack([spd(rng, 3, np.full(3, 0.05)) for _ in range(n)])
    z = mu[:, :3] + rng.normal(size=(n, 3)) * 0.05
    m, c, st = oracle.pose_update(mu, cov, 0, z, Q)
    assert (st == 0).all()
    H = np.zeros((3, 12)); H[:, :3] = np.eye(3)
    for i in range(n):
        S = H @ cov[i] @ H.T + Q[i]
        K = cov[i] @ H.T @ np.linalg.inv(S)
        delta = K @ (z[i] - mu[i, :3])
        Pk = cov[i] - K @ S @ K.T
        eu_s = [0, 1, 2, 7, 8, 9, 10, 11, 12]; eu_t = [0, 1, 2, 6, 7, 8, 9, 10, 11]
        assert max_abs(m[i, eu_s], mu[i, eu_s] + delta[eu_t]) < 1e-12
        assert max_abs(c[i][np.ix_(eu_t, eu_t)], Pk[np.ix_(eu_t, eu_t)]) < 1e-12
        # orientation: mu [+] delta
        from oracle import ukf_numpy as un
        assert max_abs(m[i, 3:7], un.quat_mul(mu[i, 3:7], oracle.so3_exp(delta[3:6]))) < 1e-12
        # SO(3) rows differ from the linear answer only at second order
        assert max_abs(c[i], Pk) < 5e-2 * np.abs(Pk).max()


def test_update_is_idempotent_in_the_limit_of_huge_measurement_noise(oracle):
    rng = np.random.default_rng(9)
    mu = np.concatenate([rng.normal(size=3), rand_quat(rng), rng.normal(size=6)])[None]
    cov = spd(rng, 12, [0.1] * 3 + [0.05] * 3 + [0.1] * 3 + [0.02] * 3)[None]
    for model in range(9):
        z = np.zeros((1, 3))
        m, c, st = oracle.pose_update(mu, cov, model, z, 1e12 * np.eye(3)[None])
        assert (st == 0).all() and max_abs(m, mu) < 1e-8 and max_abs(c, cov) < 1e-8


def test_process_models_statement_order(oracle):
    """processModelWithAcceleration: v += a dt FIRST, then p += (q v) dt with the updated v, then q
    (PoseUKF.cpp:93-95).  Orientation model: acceleration uses the already updated orientation
    (OrientationUKF.cpp:20-22)."""
    rng = np.random.default_rng(10)
    from oracle import ukf_numpy as un
    q = rand_quat(rng)
    x = np.concatenate([rng.normal(size=3), q, rng.normal(size=6)])
    acc, dt = rng.normal(size=3), 0.1
    y = oracle.pose_process(x, acc, dt)
    v1 = x[7:10] + dt * acc
    assert max_abs(y[7:10], v1) < 1e-15
    assert max_abs(y[0:3], x[0:3] + dt * oracle.quat_rotate(q, v1)) < 1e-15
    assert max_abs(y[3:7], un.quat_mul(q, oracle.so3_exp(oracle.quat_rotate(q, x[10:13]), dt))) < 1e-15
    xo = np.concatenate([q, rng.normal(size=3), 1e-3 * rng.normal(size=3), 1e-2 * rng.normal(size=3), [9.81]])
    a_in, w_in, earth = rng.normal(size=3), rng.normal(size=3) * 0.1, np.array([4e-5, 0, 6e-5])
    yo = oracle.orient_process(xo, a_in, w_in, 100.0, 200.0, earth, dt)
    q1 = un.quat_mul(q, oracle.so3_exp(oracle.quat_rotate(q, w_in - xo[7:10]) - earth, dt))
    assert max_abs(yo[0:4], q1) < 1e-15
    a_nav = oracle.quat_rotate(q1, a_in - xo[10:13]) - np.array([0, 0, 9.81])
    assert max_abs(yo[4:7], xo[4:7] + dt * a_nav) < 1e-14
    assert max_abs(yo[7:10], xo[7:10] * (1 - dt / 100.0)) < 1e-16
    assert max_abs(yo[10:13], xo[10:13] * (1 - dt / 200.0)) < 1e-16
    assert yo[13] == 9.81


def test_time_gate_matches_reference_semantics(oracle, onp):
    """UnscentedKalmanFilter.hpp:83-125: first timestamp latches without predicting; dt <= min is a silent
    skip; negative / too large dt are errors; last time advances only when dt > min."""
    ts = np.array([1_000_000, 2_000_000, 2_000_000, 1_500_000, 2_000_000 + 10 ** 9], dtype=np.int64)
    last = np.array([0, 1_000_000, 2_000_000, 2_000_000, 2_000_000], dtype=np.int64)
    new_last, dt, st = oracle.gate_timestamps(ts, last, 1e-9, 100.0)
    assert list(st) == [onp.ST_SKIPPED_FIRST_TS, 0, onp.ST_SKIPPED_SMALL_DT, onp.ST_ERR_NEG_DT, onp.ST_ERR_DT_TOO_LARGE]
    assert list(new_last) == [1_000_000, 2_000_000, 2_000_000, 2_000_000, 2_000_000 + 10 ** 9]
    assert dt[1] == 1.0 and dt[3] == -0.5
    nl2, dt2, st2 = onp.gate_timestamps(ts, last, 1e-9, 100.0)
    assert (nl2 == new_last).all() and (st2 == st).all() and max_abs(dt2, dt) == 0.0


def test_earth_rate_constant(oracle):
    assert oracle.earthw() == (2.0 * np.pi) / 86164.0   # GravitationalModel.hpp:16


def test_so3_conventions_against_scipy_rotation(oracle, onp):
    """An implementation nobody here wrote: scipy.spatial.transform.Rotation (Hamilton quaternions, scalar LAST = Eigen's
    coefficient order x, y, z, w that the reference's BodyStateMeasurement.hpp:17 and the C-ABI use).  Pins the conventions the
    recalled MTK semantics stand on -- exp = rotation vector -> quaternion, log its inverse on the principal branch, q * v the
    active rotation of a vector (Eigen's operator*), matrix(), and boxplus as RIGHT multiplication q <- q * exp(delta)
     -- in both restatements of the oracle."""
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(11)
    for _ in range(200):
        v = rng.uniform(-1.0, 1.0, 3) * rng.choice([1e-9, 1e-3, 0.3, 1.5, 3.0])
        q_ref = R.from_rotvec(v).as_quat()                       # x, y, z, w
        q = oracle.so3_exp(v)
        assert max_abs(q, q_ref) <= 1e-14 and max_abs(onp.so3_exp(v[None])[0], q_ref) <= 1e-14
        if np.linalg.norm(v) < np.pi - 1e-3:                      # principal branch: log(exp(v)) = v
            assert max_abs(oracle.so3_log(q_ref), R.from_quat(q_ref).as_rotvec()) <= 1e-12
        qa = rand_quat(rng)
        x = rng.uniform(-2, 2, 3)
        assert max_abs(oracle.quat_rotate(qa, x), R.from_quat(qa).apply(x)) <= 1e-13
        assert max_abs(oracle.quat_to_matrix(qa), R.from_quat(qa).as_matrix()) <= 1e-13
        # boxplus on the Pose manifold: position / velocity / angular velocity add, orientation right-multiplies
        state = np.concatenate([rng.uniform(-1, 1, 3), qa, rng.uniform(-1, 1, 6)])
        d = rng.uniform(-0.5, 0.5, 12)
        out = oracle.pose_boxplus(state, d)
        q_expect = (R.from_quat(qa) * R.from_rotvec(d[3:6])).as_quat()
        q_expect = q_expect if np.dot(q_expect, out[3:7]) > 0 else -q_expect   # q and -q are the same rotation
        assert max_abs(out[3:7], q_expect) <= 1e-13
        assert max_abs(out[:3], state[:3] + d[:3]) <= 1e-15 and max_abs(out[7:], state[7:] + d[6:]) <= 1e-15
        # ... and boxminus undoes it: (x [+] d) [-] x = d, the tangent of the relative rotation x^-1 (x [+] d)
        back = oracle.pose_boxminus(out, state)
        rel = (R.from_quat(qa).inv() * R.from_quat(out[3:7])).as_rotvec()
        assert max_abs(back[3:6], rel) <= 1e-12 and max_abs(back, d) <= 1e-12
