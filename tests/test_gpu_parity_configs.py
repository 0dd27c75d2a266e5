"""GPU parity of the kernel instantiations that BASELINE configs 4 and 5 actually time (and the fp32 /
timestamp variants of the fused launch), plus full-size properties of those two configurations.

  * fused predict+update launch on OrientationState (OrientationUKF.cpp:12-39,65-89) -- `ukf_kernel16<T, orient, cycle>`
  * fused per-filter-model launch on PoseWithVelocity in fp32 (PoseUKF.cpp:112-173)
  * fused predictionStepFromSampleTime + integrateMeasurement in fp32, incl. the `ts < 0` contract of
    include/ukf_batch.h (filter untouched, status INACTIVE)

Tolerances are north_star's: 1e-9 (fp64 engine) / 1e-4 (fp32 engine) against the fp64 CPU oracle."""
import numpy as np
import pytest

from conftest import max_abs

pytestmark = pytest.mark.gpu
TOL = {0: 1e-9, 1: 1e-4}
N_SMALL = 203   # not a multiple of 4: the last wavefront carries a ragged tail


class _DevArray:
    """Zero-copy torch view of engine-owned device memory (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _dev_state(eng):
    """(mean, packed covariance) of an engine as torch tensors over its own HBM buffers."""
    import torch
    eng.sync()
    mu_p, cov_p, _ = eng.device_views()
    ts = "<f8" if eng.precision == 0 else "<f4"
    return (torch.as_tensor(_DevArray(mu_p, (eng.capacity, eng.S), ts), device="cuda"),
            torch.as_tensor(_DevArray(cov_p, (eng.capacity, eng.PK), ts), device="cuda"))


def _orient_engine(spe, n, prec, G):
    s = spe.synth
    eng = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec, lanes_per_filter=G)
    eng.set_process_noise(s.orient_process_noise())
    return eng


@pytest.mark.parametrize("G", [16, 64])
@pytest.mark.parametrize("prec", [0, 1])
def test_orient_fused_cycle(spe, oracle, prec, G):
    """Config 4's launch: predict (gyro + acc driven) and body-velocity update in ONE kernel, three cycles."""
    import torch
    n = N_SMALL
    s = spe.synth
    mu, cov = s.orient_initial(n)
    eng = _orient_engine(spe, n, prec, G)
    eng.initialize(mu, cov)
    tdt = torch.float64 if prec == 0 else torch.float32
    m_o, c_o = mu.copy(), cov.copy()
    for k in range(3):
        gyro, acc, z, Q = s.orient_cycle_inputs(n, k, mu[:, :4])
        if k == 1:   # host-pointer entry point
            eng.bind_orient_inputs_dev(None, None)
            eng.set_orient_inputs(gyro, acc)
            eng.cycle(0.01, spe.MEAS_ORIENT_BODYVEL3, z, Q)
        else:        # device-pointer entry point (what bench.py times)
            g_t, a_t = torch.from_numpy(gyro).to("cuda", tdt), torch.from_numpy(acc).to("cuda", tdt)
            z_t, Q_t = torch.from_numpy(z).to("cuda", tdt), torch.from_numpy(Q.reshape(n, 9)).to("cuda", tdt)
            torch.cuda.synchronize()
            eng.bind_orient_inputs_dev(g_t, a_t)
            eng.cycle_dev(0.01, spe.MEAS_ORIENT_BODYVEL3, z_t, Q_t)
            eng.sync()
        assert "cycle" in eng.last_launch_info()["kernel"]
        # the fp32 engine sees float inputs
        cast = (lambda x: x) if prec == 0 else (lambda x: x.astype(np.float32).astype(np.float64))
        m_o, c_o, s1 = oracle.orient_predict(m_o, c_o, s.orient_process_noise(), cast(acc), cast(gyro), s.ORIENT_TAU,
                                             s.ORIENT_TAU, eng.earth_rotation, 0.01)
        m_o, c_o, s2 = oracle.orient_update(m_o, c_o, cast(z), cast(Q))
        assert (s1 == 0).all() and (s2 == 0).all()
        assert (eng.status() == 0).all()
    m_g, c_g, _ = eng.state()
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]


@pytest.mark.parametrize("prec", [0, 1])
def test_pose_mixed_fused_cycle_dev(spe, oracle, prec):
    """Config 5's launch: fused predict(acc) + per-filter measurement model from a device array, 25 % inactive."""
    import torch
    n = N_SMALL
    s = spe.synth
    tdt = torch.float64 if prec == 0 else torch.float32
    mu, cov = s.pose_initial(n)
    R = s.pose_default_process_noise()
    acc_cov = 0.01 * np.eye(3)
    eng = spe.BatchPoseUKF(n, precision=prec)
    eng.initialize(mu, cov)
    eng.set_acceleration(None, acc_cov)
    m_o, c_o = mu.copy(), cov.copy()
    seen = set()
    for k in range(3):
        acc, z, Q = s.pose_cycle_inputs(n, k, mu[:, :3], random_q=True)
        models = s.pose_mixed_models(n, k)
        zz = s.pose_measurement_for_model(mu, models, z - mu[:, :3])
        seen.update(int(m) for m in models)
        a_t = torch.from_numpy(acc).to("cuda", tdt)
        z_t, Q_t = torch.from_numpy(zz).to("cuda", tdt), torch.from_numpy(Q.reshape(n, 9)).to("cuda", tdt)
        m_t = torch.from_numpy(models).cuda()
        torch.cuda.synchronize()
        eng.bind_acceleration_dev(a_t)
        eng.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t, meas_model_dev=m_t)
        eng.sync()
        st = eng.status()
        m_o, c_o, s1 = oracle.pose_predict(m_o, c_o, R, acc, acc_cov, 0.01)
        m_p, c_p = m_o.copy(), c_o.copy()
        m_o, c_o, s2 = oracle.pose_update(m_o, c_o, models, zz, Q)
        assert (s1 == 0).all() and (st == s2).all()
        off = models < 0
        assert off.any() and ((st & spe.ST_INACTIVE) != 0)[off].all() and (st[~off] == 0).all()
        # an inactive filter is predicted but not updated
        assert max_abs(m_o[off], m_p[off]) == 0 and max_abs(c_o[off], c_p[off]) == 0
    assert seen >= set(range(9))
    m_g, c_g, _ = eng.state()
    assert max_abs(m_g, m_o) <= TOL[prec] and max_abs(c_g, c_o) <= TOL[prec]


@pytest.mark.parametrize("prec", [0, 1])
def test_cycle_timestamps_contract(spe, oracle, onp, prec):
    """ukfb_cycle_timestamps: per-filter predictionStepFromSampleTime + integrateMeasurement in one launch.
    ts < 0 means 'no sample': the filter is untouched (mean, covariance, last time) and reports INACTIVE even
    if a valid model id is present (include/ukf_batch.h)."""
    n = 64
    s = spe.synth
    mu, cov = s.pose_initial(n)
    R = s.pose_default_process_noise()
    _, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3])
    models = s.pose_mixed_models(n, 1)
    models[:8] = spe.MEAS_POS3         # valid model ids on the filters that get no sample
    zz = s.pose_measurement_for_model(mu, models, z - mu[:, :3])
    eng = spe.BatchPoseUKF(n, precision=prec)
    eng.initialize(mu, cov)
    last = np.full(n, 5_000_000, dtype=np.int64)
    last[8:12] = 0                      # first sample: latch only, no prediction, update still runs
    eng.set_last_measurement_time(last)
    ts = np.full(n, 5_010_000, dtype=np.int64)
    ts[:8] = -1
    ts[12:16] = 4_000_000               # negative dt: the reference throws, neither step touches the filter
    eng.cycle_timestamps(ts, models, zz, Q)
    st = eng.status()
    m_g, c_g, _ = eng.state()
    new_last = eng.last_measurement_time()
    # --- no-sample filters
    assert (st[:8] == spe.ST_INACTIVE).all()
    assert (new_last[:8] == last[:8]).all()
    if prec == 0:
        assert max_abs(m_g[:8], mu[:8]) == 0 and max_abs(c_g[:8], cov[:8]) == 0
    else:
        assert max_abs(m_g[:8], mu[:8].astype(np.float32)) == 0 and max_abs(c_g[:8], cov[:8].astype(np.float32)) == 0
    # --- everything else against the oracle
    nl, dt, gs = oracle.gate_timestamps(ts[8:], last[8:])
    assert (new_last[8:] == nl).all()
    m_o, c_o = mu[8:].copy(), cov[8:].copy()
    run = gs == 0
    a, b, s1 = oracle.pose_predict(m_o[run], c_o[run], R, None, None, dt[run])
    m_o[run], c_o[run] = a, b
    err = (gs & (onp.ST_ERR_NEG_DT | onp.ST_ERR_DT_TOO_LARGE)) != 0
    mods = models[8:].copy()
    upd = ~err & (mods >= 0)
    a, b, s2 = oracle.pose_update(m_o[upd], c_o[upd], mods[upd], zz[8:][upd], Q[8:][upd])
    m_o[upd], c_o[upd] = a, b
    exp = gs.copy()
    exp[~upd] |= spe.ST_INACTIVE
    assert (st[8:] == exp).all(), (st[8:], exp)
    assert max_abs(m_g[8:], m_o) <= TOL[prec] and max_abs(c_g[8:], c_o) <= TOL[prec]
    assert err.any() and (gs & onp.ST_SKIPPED_FIRST_TS).any()


def test_full_size_config4_orient_fp32(spe, oracle):
    """BASELINE config 4: 4 194 304 OrientationState filters, fp32, fused launch.  Size-independent properties:
    a batch equals its two halves bit for bit, quaternions stay unit, covariances stay SPD, every status word
    is 0; plus the oracle on a 2 048-filter slice."""
    import torch
    n, CH = 4_194_304, 262_144
    s = spe.synth
    dev = torch.device("cuda")
    split = n // 2 + 1          # ragged: the halves do not start on a wavefront boundary
    full = _orient_engine(spe, n, 1, 16)
    halves = [(_orient_engine(spe, split, 1, 16), 0, split), (_orient_engine(spe, n - split, 1, 16), split, n)]
    g_t, a_t, z_t = (torch.empty((n, 3), dtype=torch.float32, device=dev) for _ in range(3))
    Q_t = torch.empty((n, 9), dtype=torch.float32, device=dev)
    keep = {}
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        mu, cov = s.orient_initial(hi - lo, first=lo)
        gyro, acc, z, Q = s.orient_cycle_inputs(hi - lo, 0, mu[:, :4], first=lo)
        if lo == 0:
            keep = dict(mu=mu[:2048], cov=cov[:2048], gyro=gyro[:2048], acc=acc[:2048], z=z[:2048], Q=Q[:2048])
        full.initialize(mu, cov, first=lo)
        for eng, a, b in halves:
            x, y = max(lo, a), min(hi, b)
            if x < y:
                eng.initialize(mu[x - lo:y - lo], cov[x - lo:y - lo], first=x - a)
        g_t[lo:hi] = torch.from_numpy(gyro).to(dev, torch.float32); a_t[lo:hi] = torch.from_numpy(acc).to(dev, torch.float32)
        z_t[lo:hi] = torch.from_numpy(z).to(dev, torch.float32); Q_t[lo:hi] = torch.from_numpy(Q.reshape(-1, 9)).to(dev, torch.float32)
    torch.cuda.synchronize()
    for eng, lo, hi in [(full, 0, n)] + halves:
        gs, as_, zs, qs = (t[lo:hi].contiguous() for t in (g_t, a_t, z_t, Q_t))
        torch.cuda.synchronize()
        eng.bind_orient_inputs_dev(gs, as_)
        for _ in range(2):
            eng.cycle_dev(0.01, spe.MEAS_ORIENT_BODYVEL3, zs, qs)
        assert eng.status_summary() == 0
        assert eng.last_launch_info()["kernel"] == "ukf_kernel16<f32,orient,cycle-plain>"   # (one dt and one model for the launch, accept-any gate)
    # whole batch == its halves, bit for bit (compared in HBM: 4 M x 105 floats)
    mu_f, cov_f = _dev_state(full)
    for eng, lo, hi in halves:
        mu_h, cov_h = _dev_state(eng)
        assert torch.equal(mu_f[lo:hi], mu_h) and torch.equal(cov_f[lo:hi], cov_h)
    assert bool(torch.isfinite(mu_f).all()) and bool(torch.isfinite(cov_f).all())
    assert float((torch.linalg.vector_norm(mu_f[:, :4], dim=1) - 1).abs().max()) < 1e-5
    for first in (0, split - 1024, n - 2048):
        _, c_s, _ = full.state(first, 2048)
        assert (np.linalg.eigvalsh(c_s) > 0).all()
    m_f, c_f, _ = full.state(0, 2048)
    # oracle on the first 2048 filters (float inputs, as the engine saw them)
    f32 = lambda x: x.astype(np.float32).astype(np.float64)   # noqa: E731
    m_o, c_o = f32(keep["mu"]), f32(keep["cov"])
    for _ in range(2):
        m_o, c_o, s1 = oracle.orient_predict(m_o, c_o, s.orient_process_noise(), f32(keep["acc"]), f32(keep["gyro"]),
                                             s.ORIENT_TAU, s.ORIENT_TAU, full.earth_rotation, 0.01, threads=8)
        m_o, c_o, s2 = oracle.orient_update(m_o, c_o, f32(keep["z"]), f32(keep["Q"]), threads=8)
        assert (s1 == 0).all() and (s2 == 0).all()
    assert max_abs(m_f[:2048], m_o) <= 1e-4 and max_abs(c_f[:2048], c_o) <= 1e-4


def test_full_size_config5_mixed_fp64(spe, oracle):
    """BASELINE config 5: 262 144 PoseWithVelocity filters, fp64, per-filter model id over the 9 models of
    PoseUKF.cpp:112-173 and 25 % of the filters without a measurement.  Properties: filters without a measurement
    are predicted only (equal to a predict-only engine), a batch equals its halves bit for bit, and a
    2 048-filter slice matches the oracle to 1e-9."""
    import torch
    n, CH = 262_144, 131_072
    s = spe.synth
    dev = torch.device("cuda")
    R = s.pose_default_process_noise()
    acc_cov = 0.01 * np.eye(3)
    split = n // 2 + 2
    full = spe.BatchPoseUKF(n)
    ponly = spe.BatchPoseUKF(n)
    halves = [(spe.BatchPoseUKF(split), 0, split), (spe.BatchPoseUKF(n - split), split, n)]
    a_t, z_t = (torch.empty((n, 3), dtype=torch.float64, device=dev) for _ in range(2))
    Q_t = torch.empty((n, 9), dtype=torch.float64, device=dev)
    m_t = torch.empty((n,), dtype=torch.int32, device=dev)
    models_all = np.empty(n, dtype=np.int32)
    keep = {}
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        mu, cov = s.pose_initial(hi - lo, first=lo)
        acc, z, Q = s.pose_cycle_inputs(hi - lo, 0, mu[:, :3], first=lo, random_q=True)
        models = s.pose_mixed_models(hi - lo, 0, first=lo)
        zz = s.pose_measurement_for_model(mu, models, z - mu[:, :3])
        models_all[lo:hi] = models
        if lo == 0:
            keep = dict(mu=mu[:2048], cov=cov[:2048], acc=acc[:2048], z=zz[:2048], Q=Q[:2048], models=models[:2048])
        for eng in (full, ponly):
            eng.initialize(mu, cov, first=lo)
        for eng, a, b in halves:
            x, y = max(lo, a), min(hi, b)
            if x < y:
                eng.initialize(mu[x - lo:y - lo], cov[x - lo:y - lo], first=x - a)
        a_t[lo:hi] = torch.from_numpy(acc).to(dev); z_t[lo:hi] = torch.from_numpy(zz).to(dev)
        Q_t[lo:hi] = torch.from_numpy(Q.reshape(-1, 9)).to(dev); m_t[lo:hi] = torch.from_numpy(models).to(dev)
    torch.cuda.synchronize()
    for eng, lo, hi in [(full, 0, n)] + halves:
        eng.set_acceleration(None, acc_cov)
        xs = [t[lo:hi].contiguous() for t in (a_t, z_t, Q_t, m_t)]
        torch.cuda.synchronize()
        eng.bind_acceleration_dev(xs[0])
        eng.cycle_dev(0.01, spe.MEAS_POS3, xs[1], xs[2], meas_model_dev=xs[3])
        assert (eng.status_summary() & ~spe.ST_INACTIVE) == 0
        # (per-filter model ids on >= 16 384 filters: grouped by update class first, one indirect launch over the groups)
        assert eng.last_launch_info()["kernel"] == "ukf_kernel16<f64,pose,cycle-bucketed-streams>"
    ponly.set_acceleration(None, acc_cov); ponly.bind_acceleration_dev(a_t); ponly.predict(0.01)
    m_f, c_f, _ = full.state(); st = full.status()
    off = models_all < 0
    assert 0.2 < off.mean() < 0.3
    assert ((st & spe.ST_INACTIVE) != 0)[off].all() and (st[~off] == 0).all()
    m_p, c_p, _ = ponly.state()
    # (the predict-only launch is another instantiation of the kernel: same arithmetic, not the same rounding)
    assert max_abs(m_f[off], m_p[off]) <= 1e-11 and max_abs(c_f[off], c_p[off]) <= 1e-11
    m_h = np.concatenate([e.state()[0] for e, _, _ in halves]); c_h = np.concatenate([e.state()[1] for e, _, _ in halves])
    assert np.array_equal(m_f, m_h) and np.array_equal(c_f, c_h)
    assert np.isfinite(m_f).all() and np.isfinite(c_f).all()
    assert np.abs(np.linalg.norm(m_f[:, 3:7], axis=1) - 1).max() < 1e-12
    m_o, c_o, s1 = oracle.pose_predict(keep["mu"], keep["cov"], R, keep["acc"], acc_cov, 0.01, threads=8)
    m_o, c_o, s2 = oracle.pose_update(m_o, c_o, keep["models"], keep["z"], keep["Q"], threads=8)
    assert (s1 == 0).all() and (st[:2048] == s2).all()
    assert max_abs(m_f[:2048], m_o) <= 1e-9 and max_abs(c_f[:2048], c_o) <= 1e-9


def test_randomised_scenarios_against_the_oracle():
    """tests/fuzz_parity.py: batch sizes 1..256 (ragged wavefronts), both precisions, covariance scales over six decades, spins,
    time steps, missing accelerations, per-filter models with inactive filters, random SPD measurement covariances, optional
    Mahalanobis gate; separate launches and the fused cycle, Pose and Orient; state, covariance and status after every launch."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fails = fz.run(120, 7)
    assert not fails, fails[:3]
    assert fz.LAUNCHES[0] > 400 and 0 < fz.WORST[0] < 1 and 0 < fz.WORST[1] < 1


def test_model_groups_list_is_a_stable_partition(spe):
    """The list a grouped launch runs over (ukfb_last_model_groups): every filter exactly once, classes in the order sigma-point /
    linear selection / no sample, every class starting at a multiple of 4 and in ascending filter order (a stable partition), -1
    in the gaps and nowhere else -- for the path where every scatter block sums the per-block counts itself (up to 2 048 blocks of
    1 024 filters), for the one with the scan kernel in between (more blocks), and on the SECOND cycle of an engine, whose list
    still holds the first cycle's entries (the gaps are rewritten, nothing is memset)."""
    import torch
    s = spe.synth
    for n in (16_384, 20_011, 262_144, 2_097_152 + 5_000):
        eng = spe.BatchPoseUKF(n, precision=spe.F32)
        CH = 262_144
        for lo in range(0, n, CH):
            hi = min(n, lo + CH)
            mu, cov = s.pose_initial(hi - lo, first=lo)
            eng.initialize(mu, cov, first=lo)
        eng.set_acceleration(None, 0.01 * np.eye(3))
        z_t = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        Q_t = torch.eye(3, dtype=torch.float32, device="cuda").reshape(1, 9).repeat(n, 1).contiguous()
        a_t = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        eng.bind_acceleration_dev(a_t)
        for cyc in range(2):
            models = np.concatenate([s.pose_mixed_models(min(CH, n - lo), cyc, first=lo) for lo in range(0, n, CH)])
            if cyc == 1:
                models[models == spe.MEAS_ORIENT_SO3] = 0          # a class that shrinks to nothing: its old entries must not survive
                models[:5] = spe.MEAS_ORIENT_SO3                      # ... except five filters (a gap of three behind them)
            m_t = torch.from_numpy(models).cuda()
            torch.cuda.synchronize()
            eng.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t, meas_model_dev=m_t)
            eng.sync()
            assert "bucketed" in eng.last_launch_info()["kernel"]
            lst = eng.last_model_groups()
            cls = np.where(models < 0, 0, np.where(models == spe.MEAS_ORIENT_SO3, 2, 1))
            want = []
            for c in (2, 1, 0):
                idx = np.nonzero(cls == c)[0].astype(np.int32)
                pad = (-len(idx)) % 4
                want.append(np.concatenate([idx, np.full(pad, -1, dtype=np.int32)]))
            want = np.concatenate(want)
            assert len(lst) == (n + 9) // 4 * 4 and len(want) <= len(lst)
            assert (lst[:len(want)] == want).all(), (n, cyc, np.nonzero(lst[:len(want)] != want)[0][:5])
            assert (lst[len(want):] == -1).all()
        eng.close()


@pytest.mark.parametrize("prec", [0, 1])
def test_model_buckets_equal_filter_order(spe, oracle, prec):
    """ukfb_cycle_dev with per-filter model ids groups the filters by the class of their update (none / linear selection /
    OrientationMeasurement's sigma-point path, PoseUKF.cpp:7-69,112-173) before ONE indirect launch, so that no wavefront
    mixes classes.  A filter never reads another filter's data: the grouped launch must give what the launch in filter
    order gives (bucket_models = 0) -- to rounding, the wave-uniform shortcuts may differ -- and what the oracle gives.
    Sizes: ragged (not a multiple of 4 or of the 1024-filter partition blocks), and the degenerate mixes: one class only,
    a class with fewer than four filters."""
    import torch
    s = spe.synth
    tdt = torch.float64 if prec == 0 else torch.float32
    tol = TOL[prec]
    R = s.pose_default_process_noise()
    acc_cov = 0.01 * np.eye(3)
    rng = np.random.default_rng(5)
    for n, mix in ((20_011, "nine"), (16_384, "all-so3"), (17_409, "none"), (16_390, "few")):
        mu, cov = s.pose_initial(n)
        acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3], random_q=True)
        if mix == "nine":
            models = s.pose_mixed_models(n, 0)
        elif mix == "all-so3":
            models = np.full(n, spe.MEAS_ORIENT_SO3, dtype=np.int32)
        elif mix == "none":
            models = np.full(n, -1, dtype=np.int32)
        else:   # three filters in the sigma-point class, two without a sample, the rest position fixes
            models = np.zeros(n, dtype=np.int32)
            models[rng.choice(n, 3, replace=False)] = spe.MEAS_ORIENT_SO3
            models[[7, n - 1]] = -1
        zz = s.pose_measurement_for_model(mu, np.maximum(models, 0), z - mu[:, :3])
        a_t = torch.from_numpy(acc).to("cuda", tdt)
        z_t, Q_t = torch.from_numpy(zz).to("cuda", tdt), torch.from_numpy(Q.reshape(n, 9)).to("cuda", tdt)
        m_t = torch.from_numpy(models).cuda()
        torch.cuda.synchronize()
        out = []
        for buckets in (1, 0):
            eng = spe.BatchPoseUKF(n, precision=prec, bucket_models=buckets)
            eng.initialize(mu, cov)
            eng.set_acceleration(None, acc_cov)
            eng.bind_acceleration_dev(a_t)
            eng.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t, meas_model_dev=m_t)
            eng.sync()
            assert ("bucketed" in eng.last_launch_info()["kernel"]) == (buckets == 1)
            out.append((eng.state(), eng.status()))
            eng.close()
        (m_b, c_b, _), st_b = out[0]
        (m_f, c_f, _), st_f = out[1]
        assert (st_b == st_f).all()
        assert max_abs(m_b, m_f) <= tol * 1e-3 and max_abs(c_b, c_f) <= tol * 1e-3
        k = min(n, 4096)
        cast = (lambda x: x) if prec == 0 else (lambda x: x.astype(np.float32).astype(np.float64))
        m_o, c_o, s1 = oracle.pose_predict(mu[:k], cov[:k], R, cast(acc[:k]), acc_cov, 0.01, threads=8)
        m_o, c_o, s2 = oracle.pose_update(m_o, c_o, models[:k], cast(zz[:k]), cast(Q[:k]), threads=8)
        assert (s1 == 0).all() and (st_b[:k] == s2).all()
        assert max_abs(m_b[:k], m_o) <= tol and max_abs(c_b[:k], c_o) <= tol


def test_randomised_scenarios_of_the_round3_launch_paths():
    """tests/fuzz_round3.py: model-class buckets against filter order and the oracle (random class proportions, ragged sizes),
    split launches + overlapped host uploads against single launches (random operation sequences, bit for bit)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_round3", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_round3.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert not fz.run(12, 3)
