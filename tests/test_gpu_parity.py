"""GPU parity: the HIP engine (through the C-ABI, via ctypes) against the CPU oracle on identical
seeded inputs.  Tolerances are BASELINE.json's north_star: 1e-9 for fp64, 1e-4 for fp32 (absolute,
max over every state and covariance entry).  PARITY UNPINNED w.r.t. real MTK: the oracle is a
restatement (see oracle/ukf_oracle.hpp)."""
import numpy as np
import pytest

from conftest import max_abs

pytestmark = pytest.mark.gpu

TOL = {0: 1e-9, 1: 1e-4}  # F64, F32
N_SMALL = 203  # not a multiple of 4/2/1 filters per wavefront: exercises the ragged tail


def _pose_setup(spe, n, prec, G):
    mu, cov = spe.synth.pose_initial(n)
    eng = spe.BatchPoseUKF(n, precision=prec, lanes_per_filter=G)
    eng.initialize(mu, cov)
    return eng, mu, cov


@pytest.mark.parametrize("G", [16, 32, 64])
@pytest.mark.parametrize("prec", [0, 1])
def test_pose_predict_acc_branch(spe, oracle, prec, G):
    n = N_SMALL
    eng, mu, cov = _pose_setup(spe, n, prec, G)
    acc, z, Q = spe.synth.pose_cycle_inputs(n, 0, mu[:, :3])
    acc_cov = 0.01 * np.eye(3)
    eng.set_acceleration(acc, acc_cov)
    eng.predict(0.01)
    m_g, c_g, init = eng.state()
    R = spe.synth.pose_default_process_noise()
    m_o, c_o, st_o = oracle.pose_predict(mu, cov, R, acc, acc_cov, 0.01)
    assert init.all() and (eng.status() == 0).all() and (st_o == 0).all()
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]
    assert max_abs(m_g, mu) > 1e-4  # something happened


@pytest.mark.parametrize("prec", [0, 1])
def test_pose_predict_constant_velocity_branch(spe, oracle, prec):
    n = N_SMALL
    eng, mu, cov = _pose_setup(spe, n, prec, 16)
    # half the batch has an acceleration latched, the other half keeps the NaN default (PoseUKF.cpp:109)
    acc, _, _ = spe.synth.pose_cycle_inputs(n, 0)
    acc[::2] = np.nan
    eng.set_acceleration(acc, 0.02 * np.eye(3))
    eng.predict(0.05)
    m_g, c_g, _ = eng.state()
    R = spe.synth.pose_default_process_noise()
    m_o, c_o, _ = oracle.pose_predict(mu, cov, R, acc, 0.02 * np.eye(3), 0.05)
    assert (eng.status() == 0).all()
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]


@pytest.mark.parametrize("model", list(range(9)))
@pytest.mark.parametrize("prec", [0, 1])
def test_pose_update_each_model(spe, oracle, prec, model):
    n = N_SMALL
    eng, mu, cov = _pose_setup(spe, n, prec, 16)
    _, zpos, Q = spe.synth.pose_cycle_inputs(n, 1, mu[:, :3], random_q=True)
    z = spe.synth.pose_measurement_for_model(mu, np.full(n, model), zpos - mu[:, :3])
    eng.update(model, z, Q)
    m_g, c_g, _ = eng.state()
    m_o, c_o, st_o = oracle.pose_update(mu, cov, model, z, Q)
    assert (eng.status() == 0).all() and (st_o == 0).all()
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]
    assert max_abs(m_g, mu) > 1e-5


@pytest.mark.parametrize("G", [16, 64])
@pytest.mark.parametrize("prec", [0, 1])
def test_pose_fused_cycle_equals_predict_then_update(spe, oracle, prec, G):
    n = N_SMALL
    eng, mu, cov = _pose_setup(spe, n, prec, G)
    acc, z, Q = spe.synth.pose_cycle_inputs(n, 0, mu[:, :3])
    acc_cov = 0.01 * np.eye(3)
    eng.set_acceleration(acc, acc_cov)
    eng.cycle(0.01, spe.MEAS_POS3, z, Q)
    m_g, c_g, _ = eng.state()
    R = spe.synth.pose_default_process_noise()
    m_o, c_o, _ = oracle.pose_predict(mu, cov, R, acc, acc_cov, 0.01)
    m_o, c_o, st = oracle.pose_update(m_o, c_o, 0, z, Q)
    assert (eng.status() == 0).all() and (st == 0).all()
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]


@pytest.mark.parametrize("stream", ["torch", "private"])
@pytest.mark.parametrize("prec", [0, 1])
def test_core_parity_on_both_stream_kinds(spe, oracle, prec, stream):
    """The binding's default (stream=None) puts an engine on torch's current stream; a C caller's ukfb_create gives it a private
    stream, where batches of 16 384 ... 262 143 filters run as split launches on two internal streams (ukfb_config.split_streams).
    The same parity holds on both: separate predict / update launches and the fused cycle, Pose and OrientationState, on a batch
    that splits."""
    n = 16_390
    s = spe.synth
    mu, cov = s.pose_initial(n)
    acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3])
    acc_cov = 0.01 * np.eye(3)
    R = s.pose_default_process_noise()
    eng = spe.BatchPoseUKF(n, precision=prec, stream=stream)
    assert eng.stream_kind == stream and eng.config().split_streams == 1
    eng.initialize(mu, cov)
    eng.set_acceleration(acc, acc_cov)
    eng.predict(0.01)
    eng.update(spe.MEAS_POS3, z, Q)
    eng.cycle(0.01, spe.MEAS_VEL3, mu[:, 7:10] + 0.01, Q)
    m_g, c_g, _ = eng.state()
    m_o, c_o, _ = oracle.pose_predict(mu, cov, R, acc, acc_cov, 0.01)
    m_o, c_o, _ = oracle.pose_update(m_o, c_o, spe.MEAS_POS3, z, Q)
    m_o, c_o, _ = oracle.pose_predict(m_o, c_o, R, acc, acc_cov, 0.01)
    m_o, c_o, _ = oracle.pose_update(m_o, c_o, spe.MEAS_VEL3, mu[:, 7:10] + 0.01, Q)
    assert eng.status_summary() == 0
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]
    eng.close()
    mu, cov = s.orient_initial(n)
    gyro, acc, z, Q = s.orient_cycle_inputs(n, 0, mu[:, :4])
    eng = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec, stream=stream)
    eng.initialize(mu, cov)
    eng.set_process_noise(s.orient_process_noise())
    eng.set_orient_inputs(gyro, acc)
    eng.cycle(0.01, spe.MEAS_ORIENT_BODYVEL3, z, Q)
    m_g, c_g, _ = eng.state()
    m_o, c_o, _ = oracle.orient_predict(mu, cov, s.orient_process_noise(), acc, gyro, s.ORIENT_TAU, s.ORIENT_TAU, eng.earth_rotation, 0.01)
    m_o, c_o, _ = oracle.orient_update(m_o, c_o, z, Q)
    assert eng.status_summary() == 0 and max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]
    eng.close()


def test_pose_mixed_models_and_inactive_filters(spe, oracle):
    """BASELINE config 5 shape: per-filter model ids, 25 % inactive."""
    n = 1021
    eng, mu, cov = _pose_setup(spe, n, 0, 16)
    models = spe.synth.pose_mixed_models(n, 0)
    _, zpos, Q = spe.synth.pose_cycle_inputs(n, 2, mu[:, :3], random_q=True)
    z = spe.synth.pose_measurement_for_model(mu, models, zpos - mu[:, :3])
    eng.update(models, z, Q)
    m_g, c_g, _ = eng.state()
    st_g = eng.status()
    m_o, c_o, st_o = oracle.pose_update(mu, cov, models, z, Q)
    assert (st_g == st_o).all()
    assert ((st_g & spe.ST_INACTIVE) != 0).sum() == (models < 0).sum() > 0
    assert max_abs(m_g, m_o) <= 1e-9 and max_abs(c_g, c_o) <= 1e-9
    off = models < 0
    assert max_abs(m_g[off], mu[off]) == 0.0 and max_abs(c_g[off], cov[off]) == 0.0


@pytest.mark.parametrize("G", [16, 32, 64])
@pytest.mark.parametrize("prec", [0, 1])
def test_orient_predict_update(spe, oracle, prec, G):
    n = N_SMALL
    s = spe.synth
    mu, cov = s.orient_initial(n)
    gyro, acc, z, Q = s.orient_cycle_inputs(n, 0, mu[:, :4])
    eng = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec,
                                  lanes_per_filter=G)
    eng.initialize(mu, cov)
    eng.set_process_noise(s.orient_process_noise())
    eng.set_orient_inputs(gyro, acc)
    eng.predict(0.01)
    m_g, c_g, _ = eng.state()
    m_o, c_o, st = oracle.orient_predict(mu, cov, s.orient_process_noise(), acc, gyro, s.ORIENT_TAU, s.ORIENT_TAU,
                                         eng.earth_rotation, 0.01)
    assert (eng.status() == 0).all() and (st == 0).all()
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]
    eng.update(spe.MEAS_ORIENT_BODYVEL3, z, Q)
    m_g2, c_g2, _ = eng.state()
    m_o2, c_o2, st2 = oracle.orient_update(m_o, c_o, z, Q)
    assert (eng.status() == 0).all() and (st2 == 0).all()
    assert max_abs(m_g2, m_o2) <= TOL[prec] and max_abs(c_g2, c_o2) <= TOL[prec]
    rr = eng.rotation_rate()
    # read-out of latched input and mean (the fp32 engine stores the gyro sample in float)
    assert max_abs(rr, oracle.orient_rotation_rate(m_g2, gyro, eng.earth_rotation)) <= (1e-12 if prec == 0 else 1e-6)


def test_pose_trajectory_100_cycles_fp64(spe, oracle):
    """Config 1/2 shape on a small batch: 100 IMU-rate predict + position update cycles."""
    n = 64
    eng, mu, cov = _pose_setup(spe, n, 0, 16)
    R = spe.synth.pose_default_process_noise()
    acc_cov = 0.01 * np.eye(3)
    m_o, c_o = mu.copy(), cov.copy()
    for k in range(100):
        acc, z, Q = spe.synth.pose_cycle_inputs(n, k, m_o[:, :3])
        eng.set_acceleration(acc, acc_cov)
        eng.cycle(0.01, spe.MEAS_POS3, z, Q)
        m_o, c_o, s1 = oracle.pose_predict(m_o, c_o, R, acc, acc_cov, 0.01)
        m_o, c_o, s2 = oracle.pose_update(m_o, c_o, 0, z, Q)
        assert (s1 == 0).all() and (s2 == 0).all()
    m_g, c_g, _ = eng.state()
    assert eng.status_summary() == 0
    assert max_abs(m_g, m_o) <= 1e-9 and max_abs(c_g, c_o) <= 1e-9


def test_large_rotations_take_the_fallback_paths(spe, oracle):
    """Angles far outside the polynomial ranges of the fast exp / log: fast spinning filters (|omega| dt up to
    ~7 rad, beyond the 2 pi reduction), wide orientation covariances (sigma-point spread ~1 rad, half-angle
    steps of the log) and orientation measurements ~2.5 rad away from the state.  A quarter of the filters keeps
    small angles so that wavefronts mix both paths.  Spreads of this size make the UKF itself ill-conditioned
    (the iterated mean may hit its cap), so the bar here is agreement with the oracle in values AND status."""
    n = 64
    rng = np.random.default_rng(12)
    mu, cov = spe.synth.pose_initial(n)
    big = np.arange(n) % 4 != 0
    mu[big, 10:13] = rng.uniform(-40.0, 40.0, (int(big.sum()), 3))          # angular velocity, rad/s
    for i in np.nonzero(big)[0]:
        cov[i, 3:6, :] *= 10.0
        cov[i, :, 3:6] *= 10.0                                              # orientation sigma 0.05 -> 0.5 rad
    e = spe.BatchPoseUKF(n); e.initialize(mu, cov)
    R = spe.synth.pose_default_process_noise()
    e.predict(0.1)
    m_g, c_g, _ = e.state()
    m_o, c_o, st_o = oracle.pose_predict(mu, cov, R, None, None, 0.1)
    assert (e.status() == st_o).all()
    scale = max(1.0, float(np.abs(c_o).max()))
    assert max_abs(m_g, m_o) <= 1e-8 and max_abs(c_g, c_o) <= 1e-8 * scale
    # orientation measurement far from the state: innovation of ~2.5 rad through log
    z = rng.uniform(-1.0, 1.0, (n, 3)); z *= (2.5 / np.linalg.norm(z, axis=1))[:, None]
    z[~big] *= 0.01
    Q = np.stack([np.eye(3) * 0.04] * n)
    e.update(spe.MEAS_ORIENT_SO3, z, Q)
    m_g2, c_g2, _ = e.state()
    m_o2, c_o2, st_o2 = oracle.pose_update(m_o, c_o, spe.MEAS_ORIENT_SO3, z, Q)
    assert (e.status() == st_o2).all()
    scale = max(1.0, float(np.abs(c_o2).max()))
    assert max_abs(m_g2, m_o2) <= 1e-7 and max_abs(c_g2, c_o2) <= 1e-7 * scale


@pytest.mark.parametrize("prec", [0, 1])
def test_predict_mean_iteration_exit_paths(spe, oracle, prec):
    """The three ways the prediction obtains its final rotation deltas (ukf_kernel16.hpp, p_delta_r), alone and mixed inside
    one wavefront (4 consecutive filters): (a) tiny covariance, the first mean step is already below the tolerance and the
    iteration loop never runs -> logarithms against the final mean; (b) ordinary spread -> the loop's last logarithms
    re-based by so3_rebase_small; (c) orientation sigma 1.2 rad, deltas beyond the 1.5 rad bound of the series ->
    logarithms again.  ukfom's iteration and the final boxminus are what is being matched."""
    n = 96
    mu, cov = spe.synth.pose_initial(n)
    kind = np.arange(n) % 3                      # neighbours differ: every wavefront holds a mixture ...
    kind[:32] = np.repeat([0, 1, 2], 12)[:32]    # ... except the first eight, which are (mostly) uniform
    for i in range(n):
        if kind[i] == 0:
            cov[i] *= 1e-10
        elif kind[i] == 2:
            cov[i, 3:6, :] *= 24.0
            cov[i, :, 3:6] *= 24.0               # orientation sigma 0.05 -> 1.2 rad
    eng = spe.BatchPoseUKF(n, precision=prec)
    eng.initialize(mu, cov)
    acc, _, _ = spe.synth.pose_cycle_inputs(n, 0, mu[:, :3])
    acc_cov = 0.01 * np.eye(3)
    eng.set_acceleration(acc, acc_cov)
    eng.predict(0.01)
    m_g, c_g, _ = eng.state()
    R = spe.synth.pose_default_process_noise()
    m_o, c_o, st_o = oracle.pose_predict(mu, cov, R, acc, acc_cov, 0.01)
    assert (eng.status() == st_o).all()
    wide = kind == 2
    assert max_abs(m_g[~wide], m_o[~wide]) <= TOL[prec] and max_abs(c_g[~wide], c_o[~wide]) <= TOL[prec]
    scale = max(1.0, float(np.abs(c_o[wide]).max()))     # ill-conditioned by construction: relative to the covariance's size
    assert max_abs(m_g[wide], m_o[wide]) <= 10 * TOL[prec] and max_abs(c_g[wide], c_o[wide]) <= 10 * TOL[prec] * scale


def _aniso(R, seed):
    """Process noise whose rotated 3x3 diagonal blocks are full SPD matrices (the rotation of PoseUKF.cpp:184-185 /
    OrientationUKF.cpp:84-85 matters), the rest as given."""
    rng = np.random.default_rng(seed)
    R = np.array(R, dtype=np.float64)
    for b in (0, 3):
        g = rng.uniform(-1, 1, (3, 3))
        s = max(R[b, b], 1e-6)
        R[b:b + 3, b:b + 3] = s * (np.diag([1.0, 0.3, 2.5]) + 0.4 * g @ g.T)
    return R


@pytest.mark.parametrize("prec", [0, 1])
def test_rotated_process_noise_and_the_isotropic_shortcut(spe, oracle, prec):
    """Batch-uniform noise with isotropic rotated blocks (the reference default, PoseUKF.cpp:104-107) lets the kernel skip the
    rotation (R s I R^T = s I); anisotropic blocks must take the rotated path.  Both against the oracle, which always
    rotates; Pose constant-velocity branch and OrientationState, fused and separate launches."""
    n = N_SMALL
    mu, cov = spe.synth.pose_initial(n)
    for R in (spe.synth.pose_default_process_noise(), _aniso(spe.synth.pose_default_process_noise(), 3)):
        eng = spe.BatchPoseUKF(n, precision=prec)
        eng.initialize(mu, cov); eng.set_process_noise(R)
        eng.predict(0.05)                                   # no acceleration latched: processModel + rotated noise * dt
        m_g, c_g, _ = eng.state()
        m_o, c_o, st = oracle.pose_predict(mu, cov, R, None, None, 0.05)
        assert (eng.status() == 0).all() and (st == 0).all()
        assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]
    s = spe.synth
    mu, cov = s.orient_initial(n)
    gyro, acc, z, Q = s.orient_cycle_inputs(n, 0, mu[:, :4])
    for R in (s.orient_process_noise(), _aniso(s.orient_process_noise() * 1e3, 4)):
        eng = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec)
        eng.initialize(mu, cov); eng.set_process_noise(R); eng.set_orient_inputs(gyro, acc)
        eng.cycle(0.02, spe.MEAS_ORIENT_BODYVEL3, z, Q)
        m_g, c_g, _ = eng.state()
        m_o, c_o, st = oracle.orient_predict(mu, cov, R, acc, gyro, s.ORIENT_TAU, s.ORIENT_TAU, eng.earth_rotation, 0.02)
        m_o, c_o, st2 = oracle.orient_update(m_o, c_o, z, Q)
        assert (eng.status() == 0).all() and (st == 0).all() and (st2 == 0).all()
        assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]
    # a quaternion that is not of unit length switches the shortcut off for its wavefront (R R^T != I): still the oracle's result
    mu, cov = spe.synth.pose_initial(8)
    mu[2, 3:7] *= 1.01
    eng = spe.BatchPoseUKF(8, precision=prec)
    eng.initialize(mu, cov)
    eng.predict(0.05)
    m_g, c_g, _ = eng.state()
    m_o, c_o, st = oracle.pose_predict(mu, cov, spe.synth.pose_default_process_noise(), None, None, 0.05)
    assert (eng.status() == st).all()
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]
