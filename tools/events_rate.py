"""Throughput of the time-ordered asynchronous measurement stream (ukfb_process_events):
host-resident unordered events -> per-filter time order -> fused timestamp-predict + update launches."""
import torch  # noqa: F401
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slam_pose_estimation_amd as spe

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(3)
mu, cov = spe.synth.pose_initial(n)
e = spe.BatchPoseUKF(n); e.initialize(mu, cov)
e.set_last_measurement_time(np.full(n, 1_000_000, dtype=np.int64))
E = n * per
f = np.repeat(np.arange(n, dtype=np.int64), per)
t = 1_000_000 + (np.tile(np.arange(1, per + 1, dtype=np.int64), n) * 10_000) + rng.integers(0, 2000, E)
m = rng.integers(0, 9, E).astype(np.int32)
models_z = spe.synth.pose_measurement_for_model(np.repeat(mu, per, axis=0), m, rng.uniform(-0.05, 0.05, (E, 3)))
Q = np.tile(np.eye(3) * 0.0025, (E, 1, 1))
perm = rng.permutation(E)
f, t, m, z, Q = f[perm], t[perm], m[perm], models_z[perm], Q[perm]
for rep in range(3):
    e.initialize(mu, cov); e.set_last_measurement_time(np.full(n, 1_000_000, dtype=np.int64)); e.sync()
    t0 = time.perf_counter()
    st, rounds = e.process_events(f, t, m, z, Q)
    e.sync()
    dt = time.perf_counter() - t0
    print("filters %d, events %d, rounds %d, status_or %d: %.1f ms -> %.1f M events/s" % (n, E, rounds, st, dt * 1e3, E / dt / 1e6))

# the same host-resident stream over a sharded batch (ukfb_group_process_events: events routed to the owners on the host, one
# thread per shard).  On a one-GPU box every shard sits on device 0 -- the figure shows the cost of the routing, not a speed-up.
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 4
if shards > 0:
    g = spe.UKFGroup(spe.MODEL_POSE, spe.F64, n, [0] * shards)
    for rep in range(3):
        g.initialize(mu, cov)
        for sh in g.shards:
            sh["engine"].set_last_measurement_time(np.full(sh["count"], 1_000_000, dtype=np.int64))
        g.sync()
        t0 = time.perf_counter()
        st, rounds = g.process_events(f, t, m, z, Q)
        g.sync()
        dt = time.perf_counter() - t0
        print("group of %d shards on device 0: events %d, rounds %d, status_or %d: %.1f ms -> %.1f M events/s" % (shards, E, rounds, st, dt * 1e3, E / dt / 1e6))
    g.close()

# the same stream resident in HBM (engine precision): ordering, ranking and scatter on the device
td = torch.float64
d = [torch.from_numpy(f).cuda(), torch.from_numpy(t).cuda(), torch.from_numpy(m).cuda(),
     torch.from_numpy(z).to("cuda", td), torch.from_numpy(Q.reshape(-1, 9)).to("cuda", td)]
for rep in range(3):
    e.initialize(mu, cov); e.set_last_measurement_time(np.full(n, 1_000_000, dtype=np.int64)); e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st, rounds = e.process_events_dev(E, *d)
    e.sync()
    dt = time.perf_counter() - t0
    print("device-resident: events %d, rounds %d, status_or %d: %.2f ms -> %.1f M events/s" % (E, rounds, st, dt * 1e3, E / dt / 1e6))

# skewed stream: one filter with 256 samples, every 64th filter with one -> 256 rounds; the indirect per-round launches make
# the cost follow the events (a full-batch launch per round would be 256 x the fused kernel over all n filters)
k = 256
singles = np.arange(64, n, 64, dtype=np.int64)
fs = np.concatenate([np.full(k, 7, dtype=np.int64), singles])
ts_ = np.concatenate([1_000_000 + 10_000 * np.arange(1, k + 1, dtype=np.int64), 1_000_000 + rng.integers(1_000, 900_000, singles.size)])
ms = np.zeros(fs.size, dtype=np.int32)
zs = spe.synth.pose_measurement_for_model(np.concatenate([np.repeat(mu[7:8], k, axis=0), mu[singles]]), ms, rng.uniform(-0.02, 0.02, (fs.size, 3)))
Qs = np.tile(np.eye(3) * 0.0025, (fs.size, 1, 1))
ds = [torch.from_numpy(fs).cuda(), torch.from_numpy(ts_).cuda(), torch.from_numpy(ms).cuda(),
      torch.from_numpy(zs).to("cuda", td), torch.from_numpy(Qs.reshape(-1, 9)).to("cuda", td)]
for rep in range(3):
    e.initialize(mu, cov); e.set_last_measurement_time(np.full(n, 1_000_000, dtype=np.int64)); e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st, rounds = e.process_events_dev(fs.size, *ds)
    e.sync()
    dt = time.perf_counter() - t0
    print("skewed, device-resident: events %d, rounds %d, status_or %d: %.2f ms -> %.2f M events/s" % (fs.size, rounds, st, dt * 1e3, fs.size / dt / 1e6))
