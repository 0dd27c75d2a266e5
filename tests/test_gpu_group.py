"""Device groups through the C-ABI (ukfb_group_*, include/ukf_batch.h): one process, one engine per shard.

Independence basis: every filter of the reference owns its own ukf object (/root/reference/src/UnscentedKalmanFilter.hpp:150),
so contiguous shards need no collective on the data path; the one exchange is the RCCL all-gather of the means.  A gpurun
box has ONE GPU: the N > 1 data path is covered by 2, 3 and 8 shards on device 0 -- incl. the gather's staging and ragged
compaction, with peer copies standing in for the exchange RCCL refuses two ranks on one device -- and the RCCL collective
itself by a one-device group (a communicator of one rank)."""
import numpy as np
import pytest

from conftest import max_abs

pytestmark = pytest.mark.gpu


def _inputs(spe, n, prec):
    import torch
    s = spe.synth
    tdt = torch.float64 if prec == 0 else torch.float32
    mu, cov = s.pose_initial(n)
    ring = [s.pose_cycle_inputs(n, k, mu[:, :3]) for k in range(3)]
    return mu, cov, ring, tdt


@pytest.mark.parametrize("prec", [0, 1])
def test_two_shards_on_one_device_equal_the_unsharded_batch(spe, prec):
    """Host-array and device-pointer entry points of a 2-shard group against ONE engine over the same filters: bit for bit
    (same kernels, the same filters in other launches).  The shards are ragged (n odd)."""
    import torch
    n = 4099
    mu, cov, ring, tdt = _inputs(spe, n, prec)
    acc_cov = 0.01 * np.eye(3)
    one = spe.BatchPoseUKF(n, precision=prec)
    grp = spe.UKFGroup(spe.MODEL_POSE, prec, n, [0, 0])
    assert grp.n == 2 and [(s["first"], s["count"]) for s in grp.shards] == [spe.shard_range(n, 2, r) for r in range(2)]
    one.initialize(mu, cov); grp.initialize(mu, cov)
    one.set_acceleration(ring[0][0], acc_cov); grp.set_acceleration(ring[0][0], acc_cov)
    # host arrays over the whole batch
    one.cycle(0.01, spe.MEAS_POS3, ring[0][1], ring[0][2]); grp.cycle(0.01, spe.MEAS_POS3, ring[0][1], ring[0][2])
    one.predict(0.02); grp.predict(0.02)
    one.update(spe.MEAS_VEL3, ring[1][1], ring[1][2]); grp.update(spe.MEAS_VEL3, ring[1][1], ring[1][2])
    # device-resident samples: one pointer per shard
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", tdt)   # noqa: E731
    a_t, z_t, Q_t = (dev(x) for x in ring[2])
    cuts = [(s["first"], s["first"] + s["count"]) for s in grp.shards]
    a_s, z_s, Q_s = ([t[lo:hi].contiguous() for lo, hi in cuts] for t in (a_t, z_t, Q_t))
    torch.cuda.synchronize()
    one.bind_acceleration_dev(a_t); one.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t)
    grp.bind_acceleration_dev(a_s); grp.cycle_dev(0.01, spe.MEAS_POS3, z_s, Q_s)
    grp.sync(); one.sync()
    m1, c1, i1 = one.state(); mg, cg, ig = grp.state()
    assert np.array_equal(m1, mg) and np.array_equal(c1, cg) and i1.all() and ig.all()
    assert (one.status() == grp.status()).all() and grp.status_summary() == one.status_summary() == 0
    assert max_abs(m1, mu) > 1e-3
    # a range that straddles the shard boundary
    lo = grp.shards[1]["first"] - 5
    ms, cs, _ = grp.state(lo, 11)
    assert np.array_equal(ms, m1[lo:lo + 11]) and np.array_equal(cs, c1[lo:lo + 11])
    # shards that share a device gather by peer copies (RCCL wants one rank per device): same staging, same ragged compaction
    out = [torch.full((n, 13), float("nan"), dtype=tdt, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    grp.gather_means(out)
    grp.sync()
    assert grp.last_gather_exchange() == "copies"
    for o in out:
        assert np.array_equal(o.double().cpu().numpy(), m1)
    grp.close(); one.close()


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("shards,n", [(2, 101), (3, 1000), (8, 4099), (8, 9)])
def test_ragged_gather_of_n_shards_on_one_device(spe, prec, shards, n):
    """The N > 1 gather path of ukfb_group_gather_means -- padded staging, exchange, compaction of ragged shards (down to 9 filters
    over 8 shards: one shard of two, seven of one) -- on one device: every shard's output buffer holds the means of all filters in batch
    order, bit-equal to ukfb_group_get_state, also after further cycles (the staging buffers are reused)."""
    import torch
    mu, cov, ring, tdt = _inputs(spe, n, prec)
    grp = spe.UKFGroup(spe.MODEL_POSE, prec, n, [0] * shards)
    counts = [s["count"] for s in grp.shards]
    assert sum(counts) == n and max(counts) - min(counts) <= 1
    grp.initialize(mu, cov)
    grp.set_acceleration(ring[0][0], 0.01 * np.eye(3))
    out = [torch.full((n, 13), float("nan"), dtype=tdt, device="cuda") for _ in range(shards)]
    torch.cuda.synchronize()
    for k in range(2):
        grp.cycle(0.01, spe.MEAS_POS3, ring[k][1], ring[k][2])
        grp.gather_means(out)
        grp.sync()
        mg, _, _ = grp.state()
        assert grp.last_gather_exchange() == "copies" and np.isfinite(mg).all()
        for o in out:
            assert np.array_equal(o.double().cpu().numpy(), mg)
    grp.close()


@pytest.mark.parametrize("prec", [0, 1])
def test_one_device_group_gathers_over_rccl(spe, prec):
    """A group of one device: the launches are those of a plain engine (bit-equal state), and the gather runs through
    ncclCommInitAll + ncclAllGather on a communicator of one rank (the RCCL plumbing that N devices use)."""
    import torch
    n = 2051
    mu, cov, ring, tdt = _inputs(spe, n, prec)
    one = spe.BatchPoseUKF(n, precision=prec)
    grp = spe.UKFGroup(spe.MODEL_POSE, prec, n, [0])
    one.initialize(mu, cov); grp.initialize(mu, cov)
    for e in (one, grp):
        e.set_acceleration(ring[0][0], 0.01 * np.eye(3))
        e.cycle(0.01, spe.MEAS_POS3, ring[0][1], ring[0][2])
    out = [torch.full((n, 13), float("nan"), dtype=tdt, device="cuda")]
    torch.cuda.synchronize()
    grp.timer_begin()
    grp.gather_means(out)
    ms, per = grp.timer_end()
    grp.sync()
    torch.cuda.synchronize()
    m1, c1, _ = one.state(); mg, cg, _ = grp.state()
    assert np.array_equal(m1, mg) and np.array_equal(c1, cg)
    got = out[0].double().cpu().numpy()
    assert np.array_equal(got, mg) and len(per) == 1 and ms >= 0
    grp.gather_means(out)      # the communicator and the staging are reused
    grp.sync()
    assert np.array_equal(out[0].double().cpu().numpy(), mg)
    grp.close(); one.close()


def test_orientation_group_equals_the_unsharded_batch(spe):
    """OrientationState filters through the group entry points (parameters, latched IMU inputs, prediction, body-velocity
    update, fused cycle from host arrays) against one BatchOrientationUKF: bit for bit, three ragged shards on one device."""
    n = 1031
    s = spe.synth
    mu, cov = s.orient_initial(n)
    one = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE)
    grp = spe.UKFGroup(spe.MODEL_ORIENT, spe.F64, n, [0, 0, 0])
    grp.set_orient_params(s.ORIENT_TAU, s.ORIENT_TAU, one.earth_rotation)
    for e in (one, grp):
        e.set_process_noise(s.orient_process_noise())
        e.initialize(mu, cov)
        e.predict(0.02)                                   # before any IMU sample: the constructor's latches
    for k in range(2):
        gyro, acc, z, Q = s.orient_cycle_inputs(n, k, mu[:, :4])
        for e in (one, grp):
            e.set_orient_inputs(gyro, acc)
            if k == 0:
                e.cycle(0.01, spe.MEAS_ORIENT_BODYVEL3, z, Q)
            else:
                e.predict(0.01)
                e.update(spe.MEAS_ORIENT_BODYVEL3, z, Q)
    grp.sync()
    m1, c1, _ = one.state(); mg, cg, _ = grp.state()
    assert np.array_equal(m1, mg) and np.array_equal(c1, cg) and grp.status_summary() == one.status_summary() == 0
    assert max_abs(m1, mu) > 1e-4 and [sh["count"] for sh in grp.shards] == [344, 344, 343]
    grp.close(); one.close()


@pytest.mark.parametrize("prec", [0, 1])
def test_group_event_stream_timestamps_and_per_filter_models(spe, prec):
    """The asynchronous side of the boundary over a sharded batch: ukfb_group_process_events (events routed to the shard that
    owns their filter, shards run concurrently on host threads), ukfb_group_cycle_timestamps and ukfb_group_cycle_mixed_dev
    -- a 3-shard group on device 0 against ONE engine over the same filters, bit for bit, statuses included.  The third
    shard gets no event at all (its filters must read status 0 afterwards, as the filters without samples of one engine do)."""
    import torch
    s = spe.synth
    rng = np.random.default_rng(17)
    n = 1531
    mu, cov, ring, tdt = _inputs(spe, n, prec)
    one = spe.BatchPoseUKF(n, precision=prec)
    grp = spe.UKFGroup(spe.MODEL_POSE, prec, n, [0, 0, 0])
    one.initialize(mu, cov); grp.initialize(mu, cov)
    # leave a status behind everywhere (INACTIVE for the filters whose model is negative) that the event call must replace
    models = s.pose_mixed_models(n, 2)
    zz = s.pose_measurement_for_model(mu, models, ring[0][1] - mu[:, :3])
    dev = lambda x, t=tdt: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", t)   # noqa: E731
    m_t, z_t, Q_t = torch.from_numpy(models).cuda(), dev(zz), dev(ring[0][2])
    cuts = [(sh["first"], sh["first"] + sh["count"]) for sh in grp.shards]
    m_s, z_s, Q_s = ([t[lo:hi].contiguous() for lo, hi in cuts] for t in (m_t, z_t, Q_t))
    torch.cuda.synchronize()
    one.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t, meas_model_dev=m_t); grp.cycle_mixed_dev(0.01, m_s, z_s, Q_s)
    grp.sync(); one.sync()
    assert (one.status() == grp.status()).all() and (one.status() & spe.ST_INACTIVE).any()
    # events: 0..4 samples per filter of the first two shards, shuffled; none for the last shard
    last_first = grp.shards[2]["first"]
    f_ev, t_ev = [], []
    for f in range(last_first):
        k = int(rng.integers(0, 5))
        f_ev += [f] * k
        t_ev += list(1_000_000 + np.cumsum(rng.integers(1_000, 40_000, size=k)))
    f_ev, t_ev = np.array(f_ev), np.array(t_ev, dtype=np.int64)
    perm = rng.permutation(f_ev.size)
    f_ev, t_ev = f_ev[perm], t_ev[perm]
    mod_ev = rng.choice(np.array([-1, 0, 4, 3], dtype=np.int32), size=f_ev.size)
    z_ev = s.pose_measurement_for_model(mu[f_ev], np.maximum(mod_ev, 0), 0.05 * rng.normal(size=(f_ev.size, 3)))
    Q_ev = np.tile(0.01 * np.eye(3), (f_ev.size, 1, 1))
    r1 = one.process_events(f_ev, t_ev, mod_ev, z_ev, Q_ev)
    rg = grp.process_events(f_ev, t_ev, mod_ev, z_ev, Q_ev)
    assert r1 == rg and rg[1] == 4
    m1, c1, _ = one.state(); mg, cg, _ = grp.state()
    assert np.array_equal(m1, mg) and np.array_equal(c1, cg) and (one.status() == grp.status()).all()
    assert (grp.status(last_first) == 0).all() and max_abs(m1[:last_first], mu[:last_first]) > 1e-3
    # an index outside the batch is refused before anything is launched
    with pytest.raises(spe.UkfbError):
        grp.process_events([n], [1], [0], np.zeros((1, 3)), np.eye(3)[None])
    # per-filter sample times over the whole batch
    ts = np.where(rng.random(n) < 0.8, 2_000_000 + rng.integers(0, 30_000, size=n), -1).astype(np.int64)
    mod = rng.choice(np.array([-1, 0, 4], dtype=np.int32), size=n)
    zt = s.pose_measurement_for_model(mu, np.maximum(mod, 0), 0.05 * rng.normal(size=(n, 3)))
    Qt = np.tile(0.02 * np.eye(3), (n, 1, 1))
    one.cycle_timestamps(ts, mod, zt, Qt); grp.cycle_timestamps(ts, mod, zt, Qt)
    m1, c1, _ = one.state(); mg, cg, _ = grp.state()
    assert np.array_equal(m1, mg) and np.array_equal(c1, cg) and (one.status() == grp.status()).all()
    grp.close(); one.close()


def test_host_array_calls_of_a_large_group_fan_out_on_threads(spe):
    """From 32 768 filters on, the host-array calls of a group (initialize, get_state, cycle, update, cycle_timestamps) run
    one host thread per shard, so that every device's upload proceeds at once.  Same results as ONE engine, bit for bit."""
    s = spe.synth
    rng = np.random.default_rng(23)
    n = 40_001
    mu, cov = s.pose_initial(n)
    acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3], random_q=True)
    one = spe.BatchPoseUKF(n)
    grp = spe.UKFGroup(spe.MODEL_POSE, spe.F64, n, [0, 0, 0])
    one.initialize(mu, cov); grp.initialize(mu, cov)
    one.set_acceleration(acc, 0.01 * np.eye(3)); grp.set_acceleration(acc, 0.01 * np.eye(3))
    for k in range(3):
        one.cycle(0.01, spe.MEAS_POS3, z, Q); grp.cycle(0.01, spe.MEAS_POS3, z, Q)
    one.update(spe.MEAS_VEL3, mu[:, 7:10], Q); grp.update(spe.MEAS_VEL3, mu[:, 7:10], Q)
    ts = np.where(rng.random(n) < 0.7, 3_000_000 + rng.integers(0, 20_000, size=n), -1).astype(np.int64)
    mod = rng.choice(np.array([-1, 0, 4], dtype=np.int32), size=n)
    zt = s.pose_measurement_for_model(mu, np.maximum(mod, 0), 0.05 * rng.normal(size=(n, 3)))
    one.cycle_timestamps(ts, mod, zt, Q); grp.cycle_timestamps(ts, mod, zt, Q)
    m1, c1, i1 = one.state(); mg, cg, ig = grp.state()
    assert np.array_equal(m1, mg) and np.array_equal(c1, cg) and i1.all() and ig.all() and (one.status() == grp.status()).all()
    assert max_abs(m1, mu) > 1e-3 and np.isfinite(m1).all()
    # an error inside one shard's thread comes back through the group call with its text
    with pytest.raises(spe.UkfbError) as ei:
        grp.update(99, z, Q)
    assert "measurement model" in str(ei.value)
    grp.close(); one.close()
