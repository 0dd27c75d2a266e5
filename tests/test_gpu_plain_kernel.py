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
