"""The plain-launch instantiation (`ukf_kernel16<..., PLAINL>`, kernel name "...,cycle-plain>"): a fused cycle with ONE time step
and ONE measurement model for the launch, no per-filter streams and the reference's accept-any gate runs a kernel that has those
facts as its type.  Its arithmetic must be the general kernel's, bit for bit: every case below runs twice -- with the launcher's
choice, and with UKFB_NO_PLAIN_KERNEL=1 (read per launch), which keeps the general kernel -- incl. filters that fail their
factorisation, uninitialised filters, ragged ends, missing accelerations, split launches; and a launch that does not qualify
(a chi-square gate, per-filter models, an activity mask, per-filter time steps) must not take it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _twice(fn):
    out = []
    for off in ("0", "1"):
        os.environ["UKFB_NO_PLAIN_KERNEL"] = off
        try:
            out.append(fn())
        finally:
            os.environ.pop("UKFB_NO_PLAIN_KERNEL", None)
    return out


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("n", [4099, 40_003])           # one launch / two half launches on two streams
def test_pose_plain_launch_equals_the_general_kernel(spe, prec, n):
    import torch
    s = spe.synth
    tdt = torch.float64 if prec == 0 else torch.float32
    mu, cov = s.pose_initial(n)
    cov[5] = -cov[5]                                     # not factorisable: status bit, state untouched
    acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3], random_q=True)
    acc[7] = np.nan                                      # this filter takes the constant-velocity branch, its wavefront the general noise path
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", tdt)   # noqa: E731
    a_t, Q_t = dev(acc), dev(Q)
    z_by_model = {spe.MEAS_POS3: dev(z), spe.MEAS_VEL3: dev(mu[:, 7:10] + 0.01), spe.MEAS_ANGVEL3: dev(mu[:, 10:13] - 0.01)}
    torch.cuda.synchronize()

    def run():
        e = spe.BatchPoseUKF(n, precision=prec, stream="private")
        e.initialize(mu[:n - 2], cov[:n - 2])            # the last two filters stay uninitialised
        e.set_acceleration(None, 0.01 * np.eye(3))
        e.bind_acceleration_dev(a_t)
        names = []
        for model in (spe.MEAS_POS3, spe.MEAS_VEL3, spe.MEAS_ANGVEL3, spe.MEAS_POS3):
            e.cycle_dev(0.01, model, z_by_model[model], Q_t)
            names.append(e.last_launch_info()["kernel"])
        e.cycle_dev(0.01, spe.MEAS_POS_XY, z_by_model[spe.MEAS_POS3], Q_t)      # not a full 3-vector: the general kernel
        names.append(e.last_launch_info()["kernel"])
        # three cycles in one launch (no schedule): the plain multi-cycle instantiation
        z3 = torch.stack([z_by_model[spe.MEAS_VEL3]] * 2).contiguous()
        Q3 = torch.stack([Q_t] * 2).contiguous()
        torch.cuda.synchronize()
        e.cycle_multi_dev(3, 0.01, spe.MEAS_VEL3, z3, Q3, 2, 1)
        names.append(e.last_launch_info()["kernel"])
        m, c, _ = e.state()
        st = e.status()
        e.close()
        return m, c, st, names

    (m1, c1, st1, names1), (m0, c0, st0, names0) = _twice(run)
    # (POS_XY is not a full 3-vector: the streams-only level -- no per-filter streams, the general model tables)
    assert all(k.endswith(",cycle-plain>") for k in names1[:4]) and names1[4].endswith(",cycle-streams>") and names1[5].endswith(",multicycle-plain>")
    assert all(k.endswith(",cycle>") for k in names0[:5]) and names0[5].endswith(",multicycle>")
    assert np.array_equal(m1, m0, equal_nan=True) and np.array_equal(c1, c0, equal_nan=True) and (st1 == st0).all()
    assert (st1[5] & spe.ST_ERR_CHOLESKY) and (st1[n - 1] & spe.ST_UNINITIALISED) and st1[6] == 0
    assert np.isfinite(m1[:n - 2]).all() and not np.array_equal(m1[:5], mu[:5])


@pytest.mark.parametrize("prec", [0, 1])
def test_orientation_plain_launch_equals_the_general_kernel(spe, prec):
    import torch
    s = spe.synth
    n = 4099
    tdt = torch.float64 if prec == 0 else torch.float32
    mu, cov = s.orient_initial(n)
    gyro, acc, z, Q = s.orient_cycle_inputs(n, 0, mu[:, :4])
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", tdt)   # noqa: E731
    z_t, Q_t = dev(z), dev(Q)
    torch.cuda.synchronize()

    def run():
        e = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec)
        e.set_process_noise(s.orient_process_noise())
        e.initialize(mu, cov)
        e.set_orient_inputs(gyro, acc)
        for _ in range(3):
            e.cycle_dev(0.01, spe.MEAS_ORIENT_BODYVEL3, z_t, Q_t)
        name = e.last_launch_info()["kernel"]
        z2, Q2 = torch.stack([z_t] * 2).contiguous(), torch.stack([Q_t] * 2).contiguous()
        torch.cuda.synchronize()
        e.cycle_multi_dev(4, 0.01, spe.MEAS_ORIENT_BODYVEL3, z2, Q2, 2, 0)
        assert e.last_launch_info()["kernel"].endswith("multicycle-plain>" if name.endswith("plain>") else "multicycle>")
        m, c, _ = e.state()
        st = e.status()
        e.close()
        return m, c, st, name

    (m1, c1, st1, k1), (m0, c0, st0, k0) = _twice(run)
    assert k1.endswith("orient,cycle-plain>") and k0.endswith("orient,cycle>")
    assert np.array_equal(m1, m0) and np.array_equal(c1, c0) and (st1 == st0).all() and (st1 == 0).all()


@pytest.mark.parametrize("prec", [0, 1])
def test_plain_levels_of_separate_predict_and_update_launches(spe, prec):
    """Callers that keep the reference's two calls (predictionStep, then integrateMeasurement: UnscentedKalmanFilter.hpp:107-125,
    PoseUKF.cpp:112-173) get plain instantiations as well: prediction-only, and update-only at level 2 (one full-3-vector model)
    or level 1 (any uniform model / per-filter models) -- bit-identical with the general kernels, incl. an activity mask or a
    gate, which keep the general kernel."""
    import torch
    s = spe.synth
    n = 4099
    tdt = torch.float64 if prec == 0 else torch.float32
    mu, cov = s.pose_initial(n)
    cov[5] = -cov[5]
    acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3], random_q=True)
    models = s.pose_mixed_models(n, 0)
    zm = s.pose_measurement_for_model(mu, models, z - mu[:, :3])
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", tdt)   # noqa: E731
    z_t, Q_t, zm_t = dev(z), dev(Q), dev(zm)
    m_t = torch.from_numpy(models).to("cuda")
    torch.cuda.synchronize()

    def run():
        e = spe.BatchPoseUKF(n, precision=prec, stream="private")
        e.initialize(mu[:n - 2], cov[:n - 2])
        e.set_acceleration(acc, 0.01 * np.eye(3))
        names = []
        e.predict(0.01); names.append(e.last_launch_info()["kernel"])
        e.update_dev(spe.MEAS_POS3, z_t, Q_t); names.append(e.last_launch_info()["kernel"])
        e.predict(0.02); e.update_dev(spe.MEAS_POS_XY, z_t, Q_t); names.append(e.last_launch_info()["kernel"])
        e.predict(0.01); e.update_dev(0, zm_t, Q_t, meas_model_dev=m_t); names.append(e.last_launch_info()["kernel"])
        act = np.ones(n, dtype=np.uint8); act[::3] = 0
        e.update(spe.MEAS_VEL3, mu[:, 7:10], Q, active=act); names.append(e.last_launch_info()["kernel"])
        m, c, _ = e.state()
        st = e.status()
        e.close()
        return m, c, st, names

    (m1, c1, st1, names1), (m0, c0, st0, names0) = _twice(run)
    assert [k.split(",")[-1] for k in names1] == ["predict-plain>", "update-plain>", "update-streams>", "update-streams>", "update>"]
    assert [k.split(",")[-1] for k in names0] == ["predict>", "update>", "update>", "update>", "update>"]
    assert np.array_equal(m1, m0, equal_nan=True) and np.array_equal(c1, c0, equal_nan=True) and (st1 == st0).all()
    assert (st1[5] & spe.ST_ERR_CHOLESKY) and (st1[n - 1] & spe.ST_UNINITIALISED)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("model", ["pose", "orient"])
def test_short_update_factorisation_is_the_complete_one_where_it_is_taken(spe, oracle, prec, model):
    """ukfb_config.full_update_check = 0 (default): a fused plain cycle factorises only the RT + 3 columns of the downdated covariance
    that applyDelta reads when that covariance is positive definite by construction (prediction committed in the same launch, process
    noise positive semidefinite -- host check --, measurement covariance positive definite -- per-filter check).  Same bits and same
    status words as full_update_check = 1 over several cycles, including a filter whose input covariance is indefinite (its prediction
    is refused: the wavefront takes the complete factorisation), a filter whose sample carries an INDEFINITE measurement covariance
    (likewise; the status is the oracle's) and an engine whose process noise is not positive semidefinite (the host switches the short
    form off).  ukfom's applyDelta always factorises completely."""
    import torch
    s = spe.synth
    n = 4099
    tdt = torch.float64 if prec == 0 else torch.float32
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", tdt)   # noqa: E731
    if model == "pose":
        mu, cov = s.pose_initial(n)
        acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3], random_q=True)
        noise = s.pose_default_process_noise()
    else:
        mu, cov = s.orient_initial(n)
        gyro, acc, z, Q = s.orient_cycle_inputs(n, 0, mu[:, :4])
        noise = s.orient_process_noise()
    cov[5] = -cov[5]                                      # prediction refused -> no update of a verified covariance
    Q = Q.copy()
    Q[9] = np.diag([0.01, -0.5, 0.01])                    # an indefinite measurement covariance
    z_t, Q_t, a_t = dev(z), dev(Q), dev(acc)
    g_t = dev(gyro) if model == "orient" else None
    torch.cuda.synchronize()

    def run(full, bad_noise=False):
        if model == "pose":
            e = spe.BatchPoseUKF(n, precision=prec, stream="private", full_update_check=full)
            e.set_acceleration(None, 0.01 * np.eye(3))
        else:
            e = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec, stream="private", full_update_check=full)
        R = noise.copy()
        if bad_noise:
            R[7, 7] = -1e-9                               # not positive semidefinite: the host keeps the complete factorisation
        e.set_process_noise(R)
        e.initialize(mu, cov)
        if model == "pose":
            e.bind_acceleration_dev(a_t)
        else:
            e.bind_orient_inputs_dev(g_t, a_t)
        sts = []
        for _ in range(3):
            e.cycle_dev(0.01, spe.MEAS_POS3 if model == "pose" else spe.MEAS_ORIENT_BODYVEL3, z_t, Q_t)
            sts.append(e.status().copy())
        name = e.last_launch_info()["kernel"]
        m, c, _ = e.state()
        e.close()
        return m, c, sts, name

    m0, c0, st0, k0 = run(0)
    m1, c1, st1, k1 = run(1)
    assert k0 == k1 and k0.endswith("cycle-plain>")
    assert np.array_equal(m0, m1, equal_nan=True) and np.array_equal(c0, c1, equal_nan=True)
    assert all((a == b).all() for a, b in zip(st0, st1))
    assert st0[0][5] & spe.ST_ERR_CHOLESKY and st0[0][6] == 0
    # the filter with the indefinite measurement covariance: whatever the complete factorisation says (the oracle agrees)
    if model == "pose":
        mo, co, s1 = oracle.pose_predict(mu[8:12], cov[8:12], noise, acc[8:12], 0.01 * np.eye(3), 0.01)
        mo, co, s2 = oracle.pose_update(mo, co, spe.MEAS_POS3, z[8:12], Q[8:12])
        assert (st0[0][8:12] == (s1 | s2)).all()
    mb0, cb0, stb0, _ = run(0, bad_noise=True)
    mb1, cb1, stb1, _ = run(1, bad_noise=True)
    assert np.array_equal(mb0, mb1, equal_nan=True) and np.array_equal(cb0, cb1, equal_nan=True) and all((a == b).all() for a, b in zip(stb0, stb1))


@pytest.mark.parametrize("n", [4099, 20_011])      # filter order (streams-only kernel) / bucketed by update class
def test_short_update_factorisation_with_per_filter_models(spe, oracle, n):
    """The same for the streams-only kernels: per-filter measurement models (nine models, a quarter of the filters idle, measurement
    covariances whose unused trailing dimensions are padding), in filter order and bucketed -- full_update_check 0 and 1 give the same
    bits and status words, and both agree with the oracle."""
    import torch
    s = spe.synth
    mu, cov = s.pose_initial(n)
    cov[7] = -cov[7]
    acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3], random_q=True)
    models = s.pose_mixed_models(n, 0)
    zm = s.pose_measurement_for_model(mu, models, z - mu[:, :3])
    for i in np.nonzero((models == spe.MEAS_POS_Z) | (models == spe.MEAS_VEL_XY))[0][:50]:
        Q[i, 2, 2] = -1.0                                   # padding of a 1- or 2-dimensional measurement: must not matter
    dev = lambda x, t=torch.float64: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", t)   # noqa: E731
    z_t, Q_t, a_t = dev(zm), dev(Q), dev(acc)
    m_t = torch.from_numpy(models).to("cuda")
    torch.cuda.synchronize()

    def run(full):
        e = spe.BatchPoseUKF(n, stream="private", full_update_check=full)
        e.set_acceleration(None, 0.01 * np.eye(3))
        e.initialize(mu, cov)
        e.bind_acceleration_dev(a_t)
        for _ in range(2):
            e.cycle_dev(0.01, 0, z_t, Q_t, meas_model_dev=m_t)
        name = e.last_launch_info()["kernel"]
        m, c, _ = e.state()
        st = e.status()
        e.close()
        return m, c, st, name

    m0, c0, st0, k0 = run(0)
    m1, c1, st1, k1 = run(1)
    assert k0 == k1 and k0.endswith("cycle-bucketed-streams>" if n >= 16384 else "cycle-streams>")
    assert np.array_equal(m0, m1, equal_nan=True) and np.array_equal(c0, c1, equal_nan=True) and (st0 == st1).all()
    R = s.pose_default_process_noise()
    mo, co = mu[:512], cov[:512]
    for _ in range(2):
        mo2, co2, s1 = oracle.pose_predict(mo, co, R, acc[:512], 0.01 * np.eye(3), 0.01)
        mo, co = np.where((s1 == 0)[:, None], mo2, mo), np.where((s1 == 0)[:, None, None], co2, co)
        mo, co, s2 = oracle.pose_update(mo, co, models[:512], zm[:512], Q[:512])
    ok = np.ones(512, dtype=bool); ok[7] = False
    assert np.abs(m0[:512][ok] - mo[ok]).max() <= 1e-9 and np.abs(c0[:512][ok] - co[ok]).max() <= 1e-9


def test_launches_that_do_not_qualify_keep_the_general_kernel(spe):
    import torch
    s = spe.synth
    n = 1024
    mu, cov = s.pose_initial(n)
    acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3])
    dev = lambda x, t=torch.float64: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", t)   # noqa: E731
    z_t, Q_t = dev(z), dev(Q)
    models_t = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e = spe.BatchPoseUKF(n)
    e.initialize(mu, cov)
    e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t)
    assert e.last_launch_info()["kernel"].endswith("cycle-plain>")
    e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t, meas_model_dev=models_t)          # per-filter model ids: streams only
    assert e.last_launch_info()["kernel"].endswith(",cycle-streams>")
    e.cycle(0.01, spe.MEAS_POS3, z, Q)                                            # host arrays, same launch shape: plain again
    assert e.last_launch_info()["kernel"].endswith("cycle-plain>")
    e.cycle_timestamps(np.full(n, 5_000_000, dtype=np.int64), np.zeros(n, dtype=np.int32), z, Q)   # per-filter sample times
    assert e.last_launch_info()["kernel"].endswith(",cycle>")
    e.configure(gate_chi2=11.3)                                                   # a chi-square gate: the predicate is per filter
    e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t)
    assert e.last_launch_info()["kernel"].endswith(",cycle>")
    e.close()
