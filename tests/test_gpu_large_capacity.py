"""Maximum sizes: an engine of 40 000 003 PoseWithVelocity filters (36 GB of the 288 GB; the packed covariance alone has
3.1e9 elements, so element and byte offsets pass 2^31 and 2^32).  Only three islands of filters are initialised -- the first
4 099, 4 099 around the middle, the last 4 099 (ragged end: the capacity is not a multiple of the 4 filters of a workgroup) --
every other filter reports UKFB_ST_UNINITIALISED and is left alone.  The islands must come out as the same filters do in small
engines: BIT FOR BIT through direct launches (fused cycle, prediction, update), to rounding through the model-class buckets
(per-filter model ids: an indirect launch whose list holds 64-bit filter indices; the small engines run them in filter order)
and event rounds addressed to the far end of the batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 40_000_003
ISLAND = 4099


@pytest.mark.parametrize("prec", [0, 1])
def test_islands_in_a_40_million_filter_engine(spe, prec):
    import torch
    s = spe.synth
    tdt = torch.float64 if prec == 0 else torch.float32
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2**30:
        pytest.skip("needs 60 GB of free HBM")
    starts = [0, N // 2 - 2000, N - ISLAND]
    big = spe.BatchPoseUKF(N, precision=prec)
    acc_t = torch.zeros((N, 3), dtype=tdt, device="cuda")
    z_t = torch.zeros((N, 3), dtype=tdt, device="cuda")
    Q_t = torch.zeros((N, 9), dtype=tdt, device="cuda")
    models_t = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    isl = []
    for lo in starts:
        mu, cov = s.pose_initial(ISLAND, first=lo)
        acc, z, Q = s.pose_cycle_inputs(ISLAND, 0, mu[:, :3], first=lo, random_q=True)
        models = s.pose_mixed_models(ISLAND, 3)
        zz = s.pose_measurement_for_model(mu, models, z - mu[:, :3])
        big.initialize(mu, cov, first=lo)
        acc_t[lo:lo + ISLAND] = torch.from_numpy(acc).to("cuda", tdt)
        z_t[lo:lo + ISLAND] = torch.from_numpy(z).to("cuda", tdt)
        Q_t[lo:lo + ISLAND] = torch.from_numpy(Q.reshape(-1, 9)).to("cuda", tdt)
        models_t[lo:lo + ISLAND] = torch.from_numpy(models).cuda()
        isl.append((lo, mu, cov, acc, z, Q, models, zz))
    zz_t = z_t.clone()
    for lo, *_, zz in isl:
        zz_t[lo:lo + ISLAND] = torch.from_numpy(zz).to("cuda", tdt)
    torch.cuda.synchronize()

    ev_f = np.arange(ISLAND - 1, -1, -5)        # event rounds: samples for every fifth filter of an island, far end first
    ev_t = np.full(ev_f.size, 2_000_000, dtype=np.int64)

    def direct_launches(e, a_dev, z_dev, Q_dev):
        e.set_acceleration(None, 0.01 * np.eye(3))
        e.bind_acceleration_dev(a_dev)
        e.cycle_dev(0.01, spe.MEAS_POS3, z_dev, Q_dev)
        e.predict(0.02)
        e.update_dev(spe.MEAS_POS3, z_dev, Q_dev)

    def indirect_launches(e, zz_dev, Q_dev, m_dev, ev_first, z_host, Q_host):
        e.cycle_dev(0.01, spe.MEAS_POS3, zz_dev, Q_dev, meas_model_dev=m_dev)
        kernel = e.last_launch_info()["kernel"]
        if e is big:     # filters nobody initialised: reported, not touched (the event call below starts a new status word)
            assert (e.status(ISLAND, 64) & spe.ST_UNINITIALISED).all() and (e.status(N // 2 + 5000, 64) & spe.ST_UNINITIALISED).all()
        f = np.concatenate([lo + ev_f for lo in ev_first])
        e.process_events(f, np.tile(ev_t, len(ev_first)), np.full(f.size, spe.MEAS_VEL3, dtype=np.int32), z_host, Q_host)
        return kernel

    def snapshot(e, lo):
        m, c, init = e.state(lo, ISLAND)
        assert init.all() and np.isfinite(m).all() and np.isfinite(c).all()
        return m, c, e.status(lo, ISLAND)

    direct_launches(big, acc_t, z_t, Q_t)
    snap_a = {lo: snapshot(big, lo) for lo in starts}
    kernel = indirect_launches(big, zz_t, Q_t, models_t, starts, np.concatenate([i[4][ev_f] for i in isl]), np.concatenate([i[5][ev_f] for i in isl]))
    assert "bucketed" in kernel                  # 40 M per-filter model ids: grouped by class on the device first
    snap_b = {lo: snapshot(big, lo) for lo in starts}
    # Direct launches: bit for bit -- also for the filters that share a wavefront with an uninitialised neighbour (the islands'
    # edges; rows that commit nothing have no say in the wave-uniform shortcuts of their wave-mates).  The grouped launch
    # runs a filter in a class-uniform wavefront, the small engine in filter order (mixed wavefronts): equal to rounding.
    tol = 1e-13 if prec == 0 else 2e-5
    for k, (lo, mu, cov, acc, z, Q, models, zz) in enumerate(isl):
        small = spe.BatchPoseUKF(ISLAND, precision=prec)
        small.initialize(mu, cov)
        sl = slice(lo, lo + ISLAND)
        a_s, z_s, Q_s, zz_s, m_s = (x[sl].contiguous() for x in (acc_t, z_t, Q_t, zz_t, models_t))
        torch.cuda.synchronize()
        direct_launches(small, a_s, z_s, Q_s)
        m_s_, c_s_, st_s = snapshot(small, 0)
        m_b, c_b, st_b = snap_a[lo]
        assert np.array_equal(m_b, m_s_) and np.array_equal(c_b, c_s_) and (st_b == st_s).all(), (prec, lo, float(np.abs(m_b - m_s_).max()))
        assert not np.array_equal(m_b, mu)
        indirect_launches(small, zz_s, Q_s, m_s, [0], z[ev_f], Q[ev_f])
        m_s_, c_s_, st_s = snapshot(small, 0)
        m_b, c_b, st_b = snap_b[lo]
        assert np.abs(m_b - m_s_).max() <= tol and np.abs(c_b - c_s_).max() <= tol and (st_b == st_s).all(), (prec, lo, float(np.abs(m_b - m_s_).max()))
        assert not np.array_equal(m_b, snap_a[lo][0])
        small.close()
    big.close()
