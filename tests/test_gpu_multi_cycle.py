"""GPU tests of the multi-cycle launch (ukfb_cycle_multi_dev): `cycles` fused predict + update cycles in ONE kernel
with the filter resident in LDS in between.  Contract (include/ukf_batch.h): the same arithmetic as `cycles` calls of
ukfb_cycle_dev, so the state must agree BIT FOR BIT with the single-cycle launches, and therefore with the oracle to the
usual tolerance (PoseUKF.cpp:112-117,188-193; OrientationUKF.cpp:65-89)."""
import numpy as np
import pytest

from conftest import max_abs

pytestmark = pytest.mark.gpu
TOL = {0: 1e-9, 1: 1e-4}
N = 203   # not a multiple of the 4 filters of a wavefront


def _tdt(prec):
    import torch
    return torch.float64 if prec == 0 else torch.float32


def _rings(arrs, prec):
    """[slots][n][..] device rings from per-slot host arrays"""
    import torch
    t = torch.from_numpy(np.stack(arrs)).to("cuda", _tdt(prec)).contiguous()
    torch.cuda.synchronize()       # the narrowing to fp32 runs on torch's stream; the engine launches on its own
    return t


@pytest.mark.parametrize("G", [16, 64])
@pytest.mark.parametrize("prec", [0, 1])
def test_pose_multi_cycle_equals_single_launches(spe, oracle, prec, G):
    import torch
    s = spe.synth
    slots, cycles, first = 4, 7, 2          # wraps the ring twice, starts in the middle
    mu, cov = s.pose_initial(N)
    ins = [s.pose_cycle_inputs(N, k, mu[:, :3], random_q=True) for k in range(slots)]
    acc_r = _rings([i[0] for i in ins], prec)
    z_r = _rings([i[1] for i in ins], prec)
    Q_r = _rings([i[2].reshape(N, 9) for i in ins], prec)
    acc_cov = 0.01 * np.eye(3)

    def engine():
        e = spe.BatchPoseUKF(N, precision=prec, lanes_per_filter=G)
        e.initialize(mu, cov)
        e.set_acceleration(None, acc_cov)
        return e
    a = engine()
    for c in range(cycles):
        k = (first + c) % slots
        a.bind_acceleration_dev(acc_r[k])
        a.cycle_dev(0.01, spe.MEAS_POS3, z_r[k], Q_r[k])
    a.sync()
    b = engine()
    b.cycle_multi_dev(cycles, 0.01, spe.MEAS_POS3, z_r, Q_r, slots, first, in_a_dev=acc_r)
    b.sync()
    if G == 16:
        assert "multicycle" in b.last_launch_info()["kernel"]
    ma, ca, _ = a.state()
    mb, cb, _ = b.state()
    assert np.array_equal(ma, mb) and np.array_equal(ca, cb)
    assert (a.status() == 0).all() and (b.status() == 0).all()
    # and against the oracle
    cast = (lambda x: x) if prec == 0 else (lambda x: x.astype(np.float32).astype(np.float64))
    m_o, c_o = mu.copy(), cov.copy()
    R = s.pose_default_process_noise()
    for c in range(cycles):
        acc, z, Q = ins[(first + c) % slots]
        m_o, c_o, s1 = oracle.pose_predict(m_o, c_o, R, cast(acc), acc_cov, 0.01)
        m_o, c_o, s2 = oracle.pose_update(m_o, c_o, np.full(N, spe.MEAS_POS3, dtype=np.int32), cast(z), cast(Q))
        assert (s1 == 0).all() and (s2 == 0).all()
    assert max_abs(mb, m_o) <= TOL[prec] and max_abs(cb, c_o) <= TOL[prec]
    del torch


@pytest.mark.parametrize("prec", [0, 1])
def test_pose_multi_cycle_latched_acceleration_and_one_slot(spe, prec):
    """no input rings for the process model: the latched acceleration serves every cycle; one slot of z, Q"""
    s = spe.synth
    mu, cov = s.pose_initial(N)
    acc, z, Q = s.pose_cycle_inputs(N, 0, mu[:, :3])
    z_r, Q_r = _rings([z], prec), _rings([Q.reshape(N, 9)], prec)

    def engine():
        e = spe.BatchPoseUKF(N, precision=prec)
        e.initialize(mu, cov)
        e.set_acceleration(acc, 0.01 * np.eye(3))
        return e
    a, b = engine(), engine()
    for _ in range(3):
        a.cycle_dev(0.01, spe.MEAS_POS3, z_r[0], Q_r[0])
    b.cycle_multi_dev(3, 0.01, spe.MEAS_POS3, z_r, Q_r, 1)
    ma, ca, _ = a.state()
    mb, cb, _ = b.state()
    assert np.array_equal(ma, mb) and np.array_equal(ca, cb)
    # one cycle through the multi-cycle kernel = one single launch
    c1, d1 = engine(), engine()
    c1.cycle_dev(0.01, spe.MEAS_POS3, z_r[0], Q_r[0])
    d1.cycle_multi_dev(1, 0.01, spe.MEAS_POS3, z_r, Q_r, 1)
    assert np.array_equal(c1.state()[0], d1.state()[0]) and np.array_equal(c1.state()[1], d1.state()[1])
    # zero cycles: nothing happens
    d1.cycle_multi_dev(0, 0.01, spe.MEAS_POS3, z_r, Q_r, 1)
    assert np.array_equal(c1.state()[1], d1.state()[1])


@pytest.mark.parametrize("prec", [0, 1])
def test_orient_multi_cycle_equals_single_launches(spe, oracle, prec):
    s = spe.synth
    slots, cycles = 3, 5
    mu, cov = s.orient_initial(N)
    ins = [s.orient_cycle_inputs(N, k, mu[:, :4]) for k in range(slots)]
    g_r = _rings([i[0] for i in ins], prec)
    a_r = _rings([i[1] for i in ins], prec)
    z_r = _rings([i[2] for i in ins], prec)
    Q_r = _rings([i[3].reshape(N, 9) for i in ins], prec)

    def engine():
        e = spe.BatchOrientationUKF(N, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec)
        e.set_process_noise(s.orient_process_noise())
        e.initialize(mu, cov)
        return e
    a = engine()
    for c in range(cycles):
        k = c % slots
        a.bind_orient_inputs_dev(g_r[k], a_r[k])
        a.cycle_dev(0.01, spe.MEAS_ORIENT_BODYVEL3, z_r[k], Q_r[k])
    a.sync()
    b = engine()
    b.cycle_multi_dev(cycles, 0.01, spe.MEAS_ORIENT_BODYVEL3, z_r, Q_r, slots, 0, in_a_dev=a_r, in_b_dev=g_r)
    b.sync()
    ma, ca, _ = a.state()
    mb, cb, _ = b.state()
    assert np.array_equal(ma, mb) and np.array_equal(ca, cb)
    assert (b.status() == 0).all()
    cast = (lambda x: x) if prec == 0 else (lambda x: x.astype(np.float32).astype(np.float64))
    m_o, c_o = mu.copy(), cov.copy()
    for c in range(cycles):
        gyro, acc, z, Q = ins[c % slots]
        m_o, c_o, s1 = oracle.orient_predict(m_o, c_o, s.orient_process_noise(), cast(acc), cast(gyro), s.ORIENT_TAU,
                                             s.ORIENT_TAU, b.earth_rotation, 0.01)
        m_o, c_o, s2 = oracle.orient_update(m_o, c_o, cast(z), cast(Q))
    assert max_abs(mb, m_o) <= TOL[prec] and max_abs(cb, c_o) <= TOL[prec]


def test_multi_cycle_status_is_the_or_over_cycles_and_failures_skip(spe):
    """a filter whose covariance is indefinite fails its predictions (state untouched, ERR_CHOLESKY) in every cycle;
    an uninitialised filter reports UNINITIALISED; a time step beyond max_time_delta gates every cycle; the others run"""
    s = spe.synth
    n = 9
    mu, cov = s.pose_initial(n)
    cov[3] = -np.eye(12)
    acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3])
    z_r, Q_r = _rings([z], 0), _rings([Q.reshape(n, 9)], 0)
    e = spe.BatchPoseUKF(n)
    e.initialize(mu[:8], cov[:8])            # filter 8 stays uninitialised
    e.set_acceleration(acc, 0.01 * np.eye(3))
    e.cycle_multi_dev(3, 0.01, spe.MEAS_POS3, z_r, Q_r, 1)
    st = e.status()
    assert st[3] & spe.ST_ERR_CHOLESKY and st[8] == spe.ST_UNINITIALISED
    assert (np.delete(st, [3, 8]) == 0).all()
    m, c, _ = e.state()
    assert np.array_equal(m[3], mu[3]) and np.array_equal(c[3], cov[3])
    before = e.state()
    e.configure(max_time_delta=2.0)
    e.cycle_multi_dev(2, 5.0, spe.MEAS_POS3, z_r, Q_r, 1)       # dt > max_time_delta: every cycle is gated
    assert (e.status()[:3] & spe.ST_ERR_DT_TOO_LARGE).all()
    assert np.array_equal(e.state()[0][:3], before[0][:3])
    with pytest.raises(spe.UkfbError):
        e.cycle_multi_dev(2, 0.01, spe.MEAS_POS3, z_r, Q_r, 1, first_slot=1)
    with pytest.raises(spe.UkfbError):
        e.cycle_multi_dev(2, 0.01, spe.MEAS_ORIENT_BODYVEL3, z_r, Q_r, 1)


@pytest.mark.parametrize("prec", [0, 1])
def test_host_pointer_multi_cycle(spe, prec):
    """ukfb_cycle_multi: host doubles, one input set per cycle, uploaded to an engine-owned ring; equals single cycles"""
    s = spe.synth
    cycles = 5
    mu, cov = s.pose_initial(N)
    ins = [s.pose_cycle_inputs(N, k, mu[:, :3]) for k in range(cycles)]
    acc = np.stack([i[0] for i in ins]); z = np.stack([i[1] for i in ins]); Q = np.stack([i[2] for i in ins])

    def engine():
        e = spe.BatchPoseUKF(N, precision=prec)
        e.initialize(mu, cov)
        e.set_acceleration(None, 0.01 * np.eye(3))
        return e
    a, b = engine(), engine()
    for c in range(cycles):
        a.set_acceleration(acc[c], None)
        a.cycle(0.01, spe.MEAS_POS3, z[c], Q[c])
    b.cycle_multi(0.01, spe.MEAS_POS3, z, Q, in_a=acc)
    assert np.array_equal(a.state()[0], b.state()[0]) and np.array_equal(a.state()[1], b.state()[1])
    assert (b.status() == 0).all()
    b.cycle_multi(0.01, spe.MEAS_POS3, z[:2], Q[:2])      # shorter call, latched acceleration (the ring is reused)
    assert (b.status() == 0).all()


@pytest.mark.parametrize("G", [16, 64])
@pytest.mark.parametrize("prec", [0, 1])
def test_scheduled_multi_cycle(spe, oracle, prec, G):
    """ukfb_cycle_schedule_dev: every cycle its own dt and measurement model, negative = prediction only -- an IMU-rate
    filter with slower aiding sensors replayed from a buffer (40 cycles = two launches of the tuned kernel).  Against the
    same sequence as single launches (predict / fused cycle) and against the oracle."""
    s = spe.synth
    rng = np.random.default_rng(5)
    slots, cycles = 8, 40
    mu, cov = s.pose_initial(N)
    ins = [s.pose_cycle_inputs(N, k, mu[:, :3], random_q=True) for k in range(slots)]
    acc_r = _rings([i[0] for i in ins], prec)
    Q_r = _rings([i[2].reshape(N, 9) for i in ins], prec)
    # measurement values per slot: position-like numbers are fine for every 3-vector model (the filter just follows them)
    z_host = [0.1 * i[1] for i in ins]
    z_r = _rings(z_host, prec)
    dts = rng.uniform(0.005, 0.02, cycles)
    models = np.full(cycles, -1, dtype=np.int32)
    models[4::5] = spe.MEAS_VEL3            # every 5th cycle a velocity sample
    models[9::10] = spe.MEAS_POS3           # every 10th a position fix instead
    models[17] = spe.MEAS_ANGVEL3
    acc_cov = 0.01 * np.eye(3)

    def engine():
        e = spe.BatchPoseUKF(N, precision=prec, lanes_per_filter=G)
        e.initialize(mu, cov)
        e.set_acceleration(None, acc_cov)
        return e
    a = engine()
    for c in range(cycles):
        k = c % slots
        a.bind_acceleration_dev(acc_r[k])
        if models[c] < 0:
            a.predict(float(dts[c]))
        else:
            a.cycle_dev(float(dts[c]), int(models[c]), z_r[k], Q_r[k])
    a.sync()
    b = engine()
    b.cycle_schedule_dev(dts, models, z_r, Q_r, slots, 0, in_a_dev=acc_r)
    b.sync()
    ma, ca, _ = a.state()
    mb, cb, _ = b.state()
    # the prediction-only launches of `a` are another kernel instantiation than the fused one: equal up to rounding
    tight = 1e-12 if prec == 0 else 2e-5
    assert max_abs(ma, mb) <= tight and max_abs(ca, cb) <= tight
    assert (b.status() == 0).all()          # prediction-only cycles are plain predictionSteps: no INACTIVE mark
    cast = (lambda x: x) if prec == 0 else (lambda x: x.astype(np.float32).astype(np.float64))
    m_o, c_o = mu.copy(), cov.copy()
    R = s.pose_default_process_noise()
    for c in range(cycles):
        k = c % slots
        m_o, c_o, _ = oracle.pose_predict(m_o, c_o, R, cast(ins[k][0]), acc_cov, float(dts[c]))
        if models[c] >= 0:
            m_o, c_o, _ = oracle.pose_update(m_o, c_o, np.full(N, models[c], dtype=np.int32), cast(z_host[k]), cast(ins[k][2]))
    assert max_abs(mb, m_o) <= TOL[prec] and max_abs(cb, c_o) <= TOL[prec]
    with pytest.raises(spe.UkfbError):
        b.cycle_schedule_dev([0.01], [spe.MEAS_ORIENT_BODYVEL3], z_r, Q_r, slots)


@pytest.mark.parametrize("G", [16, 32])
@pytest.mark.parametrize("prec", [0, 1])
def test_mixed_model_multi_cycle(spe, oracle, prec, G):
    """ukfb_cycle_multi_mixed_dev: per-filter model ids per cycle (BASELINE config 5's stream, buffered), negative = none"""
    import torch
    s = spe.synth
    slots, cycles = 4, 6
    mu, cov = s.pose_initial(N)
    acc_cov = 0.01 * np.eye(3)
    ins, mods = [], []
    for k in range(slots):
        acc, z, Q = s.pose_cycle_inputs(N, k, mu[:, :3], random_q=True)
        models = s.pose_mixed_models(N, k)
        z = s.pose_measurement_for_model(mu, models, z - mu[:, :3])
        ins.append((acc, z, Q)); mods.append(models)
    acc_r = _rings([i[0] for i in ins], prec)
    z_r = _rings([i[1] for i in ins], prec)
    Q_r = _rings([i[2].reshape(N, 9) for i in ins], prec)
    m_r = torch.from_numpy(np.stack(mods).astype(np.int32)).cuda().contiguous()

    def engine():
        e = spe.BatchPoseUKF(N, precision=prec, lanes_per_filter=G)
        e.initialize(mu, cov)
        e.set_acceleration(None, acc_cov)
        return e
    a = engine()
    st_or = np.zeros(N, dtype=np.uint32)
    for c in range(cycles):
        k = c % slots
        a.bind_acceleration_dev(acc_r[k])
        a.cycle_dev(0.01, spe.MEAS_POS3, z_r[k], Q_r[k], meas_model_dev=m_r[k])
        st_or |= a.status()
    b = engine()
    b.cycle_multi_mixed_dev(cycles, 0.01, m_r, z_r, Q_r, slots, 0, in_a_dev=acc_r)
    ma, ca, _ = a.state()
    mb, cb, _ = b.state()
    assert np.array_equal(ma, mb) and np.array_equal(ca, cb)
    assert (b.status() == st_or).all()                 # OR over the cycles, INACTIVE where a filter had no sample
    cast = (lambda x: x) if prec == 0 else (lambda x: x.astype(np.float32).astype(np.float64))
    m_o, c_o = mu.copy(), cov.copy()
    R = s.pose_default_process_noise()
    for c in range(cycles):
        acc, z, Q = ins[c % slots]
        m_o, c_o, _ = oracle.pose_predict(m_o, c_o, R, cast(acc), acc_cov, 0.01)
        m_o, c_o, _ = oracle.pose_update(m_o, c_o, mods[c % slots], cast(z), cast(Q))
    assert max_abs(mb, m_o) <= TOL[prec] and max_abs(cb, c_o) <= TOL[prec]


@pytest.mark.parametrize("prec", [0, 1])
def test_full_size_multi_cycle_equals_single_launches(spe, prec):
    """The metric's 1 048 576 PoseWithVelocity filters: one launch of 6 cycles (ring of 4 input sets, starting at slot 3) and
    the same 6 cycles as single launches leave the SAME bits in HBM -- compared on the device, every filter."""
    import torch
    from test_gpu_parity_configs import _dev_state
    s = spe.synth
    n, CH, slots, cycles, first = 1_048_576, 131_072, 4, 6, 3
    tdt = _tdt(prec)
    eng = spe.BatchPoseUKF(n, precision=prec)
    acc_r = torch.empty((slots, n, 3), dtype=tdt, device="cuda")
    z_r = torch.empty((slots, n, 3), dtype=tdt, device="cuda")
    Q_r = torch.empty((slots, n, 9), dtype=tdt, device="cuda")
    for lo in range(0, n, CH):
        mu, cov = s.pose_initial(CH, first=lo)
        eng.initialize(mu, cov, first=lo)
        for k in range(slots):
            acc, z, Q = s.pose_cycle_inputs(CH, k, mu[:, :3], first=lo)
            acc_r[k, lo:lo + CH] = torch.from_numpy(acc).to("cuda", tdt)
            z_r[k, lo:lo + CH] = torch.from_numpy(z).to("cuda", tdt)
            Q_r[k, lo:lo + CH] = torch.from_numpy(Q.reshape(-1, 9)).to("cuda", tdt)
    eng.set_acceleration(None, 0.01 * np.eye(3))
    torch.cuda.synchronize()       # the rings were filled on torch's stream, the engine launches on its own
    mu_v, cov_v = _dev_state(eng)
    mu0, cov0 = mu_v.clone(), cov_v.clone()
    torch.cuda.synchronize()
    for c in range(cycles):
        k = (first + c) % slots
        eng.bind_acceleration_dev(acc_r[k])
        eng.cycle_dev(0.01, spe.MEAS_POS3, z_r[k], Q_r[k])
    eng.sync()
    assert eng.status_summary() == 0
    mu1, cov1 = mu_v.clone(), cov_v.clone()
    mu_v.copy_(mu0); cov_v.copy_(cov0)
    torch.cuda.synchronize()       # (and the restored state before the engine's stream reads it)
    eng.cycle_multi_dev(cycles, 0.01, spe.MEAS_POS3, z_r, Q_r, slots, first, in_a_dev=acc_r)
    eng.sync()
    assert eng.status_summary() == 0 and "multicycle" in eng.last_launch_info()["kernel"]
    assert torch.equal(mu_v, mu1) and torch.equal(cov_v, cov1)
    assert not torch.equal(mu_v, mu0)
