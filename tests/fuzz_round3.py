#!/usr/bin/env python3
"""Runs ON THE GPU BOX: randomised scenarios for the launch paths round 3 added (test infrastructure; drives the oracle).

  * model-class buckets: a mixed per-filter-model launch grouped by update class against the same launch in filter order
    (ukfb_config.bucket_models 1 / 0) and against the oracle on a random subset -- random class proportions incl. the
    degenerate ones, ragged sizes above the 16 384-filter threshold;
  * split launches + overlapped host uploads: a random sequence of device-pointer cycles, host-array cycles / updates
    (caller's arrays overwritten at return), predictions, multi-cycle launches and downloads on an engine that owns its
    stream with split_streams 1 against the same sequence with split_streams 0 -- bit for bit (filters are independent).

usage: python3 tests/fuzz_round3.py [scenarios=40] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import slam_pose_estimation_amd as spe  # noqa: E402
from oracle import capi as oracle  # noqa: E402

TOL = {0: 1e-9, 1: 1e-4}


def mad(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0


def dev(x, tdt):
    return torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", tdt)


def bucket_scenario(rng, k):
    prec = int(rng.integers(0, 2))
    tdt = torch.float64 if prec == 0 else torch.float32
    n = int(rng.integers(16384, 30000))
    s = spe.synth
    mu, cov = s.pose_initial(n, first=int(rng.integers(0, 1 << 20)))
    acc, z, Q = s.pose_cycle_inputs(n, int(rng.integers(0, 50)), mu[:, :3], random_q=True)
    # class proportions: none / linear / SO(3), sometimes degenerate
    p = rng.dirichlet([0.7, 1.5, 0.5]) if rng.random() < 0.8 else np.eye(3)[rng.integers(0, 3)]
    cls = rng.choice(3, size=n, p=p)
    lin = rng.choice(np.array([0, 1, 2, 4, 5, 6, 7, 8], dtype=np.int32), size=n)
    models = np.where(cls == 0, -1, np.where(cls == 2, 3, lin)).astype(np.int32)
    zz = s.pose_measurement_for_model(mu, np.maximum(models, 0), z - mu[:, :3])
    use_acc = rng.random() < 0.7
    a_t, z_t, Q_t = dev(acc, tdt), dev(zz, tdt), dev(Q, tdt)
    m_t = torch.from_numpy(models).cuda()
    torch.cuda.synchronize()
    dt = float(rng.choice([0.005, 0.01, 0.05]))
    # (a third of the fp32 scenarios in the wide-arithmetic mode -- fp32 arrays, fp64 arithmetic; drawn from a stream of its own so
    #  that the scenarios of a seed are what they were before the mode existed)
    wide = {"wide_arithmetic": 1} if (prec == 1 and np.random.default_rng([k, 77]).random() < 0.34) else {}
    out = []
    for b in (1, 0):
        e = spe.BatchPoseUKF(n, precision=prec, bucket_models=b, **wide)
        e.initialize(mu, cov)
        e.set_acceleration(None, 0.01 * np.eye(3))
        if use_acc:
            e.bind_acceleration_dev(a_t)
        e.cycle_dev(dt, spe.MEAS_POS3, z_t, Q_t, meas_model_dev=m_t)
        e.sync()
        if ("bucketed" in e.last_launch_info()["kernel"]) != (b == 1):
            return f"bucket k={k}: wrong kernel {e.last_launch_info()['kernel']} for bucket_models={b}"
        out.append((e.state(), e.status()))
        e.close()
    (mb, cb, _), sb = out[0]
    (mf, cf, _), sf = out[1]
    if not (sb == sf).all():
        return f"bucket k={k} n={n} prec={prec}: status differs at {np.nonzero(sb != sf)[0][:5]}"
    if mad(mb, mf) > TOL[prec] * 1e-3 or mad(cb, cf) > TOL[prec] * 1e-3:
        return f"bucket k={k} n={n} prec={prec}: grouped vs filter order {mad(mb, mf):.3e} {mad(cb, cf):.3e}"
    idx = np.sort(rng.choice(n, 384, replace=False))
    cast = (lambda x: x) if prec == 0 else (lambda x: x.astype(np.float32).astype(np.float64))
    mo, co, s1 = oracle.pose_predict(mu[idx], cov[idx], s.pose_default_process_noise(), cast(acc[idx]) if use_acc else None,
                                     0.01 * np.eye(3), dt, threads=8)
    mo, co, s2 = oracle.pose_update(mo, co, models[idx], cast(zz[idx]), cast(Q[idx]), threads=8)
    if not ((s1 | s2) == sb[idx]).all():
        return f"bucket k={k} n={n} prec={prec}: status vs oracle"
    scale = max(1.0, float(np.abs(co).max()))
    if mad(mb[idx], mo) > TOL[prec] * scale or mad(cb[idx], co) > TOL[prec] * scale:
        return f"bucket k={k} n={n} prec={prec} p={np.round(p, 2)}: vs oracle {mad(mb[idx], mo):.3e} {mad(cb[idx], co):.3e}"
    return None


def split_scenario(rng, k):
    prec = int(rng.integers(0, 2))
    tdt = torch.float64 if prec == 0 else torch.float32
    orient = rng.random() < 0.35
    n = int(rng.integers(16384, 50000))
    s = spe.synth
    if orient:
        mu, cov = s.orient_initial(n)
        ring = [s.orient_cycle_inputs(n, c, mu[:, :4]) for c in range(3)]     # gyro, acc, z, Q
        zi, Qi, model = 2, 3, spe.MEAS_ORIENT_BODYVEL3
    else:
        mu, cov = s.pose_initial(n)
        ring = [s.pose_cycle_inputs(n, c, mu[:, :3]) for c in range(3)]       # acc, z, Q
        zi, Qi, model = 1, 2, spe.MEAS_POS3
    dring = [tuple(dev(x, tdt) for x in r) for r in ring]
    z_ring = torch.stack([d[zi] for d in dring]).contiguous()
    Q_ring = torch.stack([d[Qi] for d in dring]).contiguous()
    torch.cuda.synchronize()
    ops = [str(rng.choice(["cycle_dev", "cycle_host", "update_host", "predict", "multi", "download", "uniq"])) for _ in range(int(rng.integers(4, 9)))]
    out = []
    for split in (1, 0):
        if orient:
            e = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec, stream="private", split_streams=split)
            e.set_process_noise(s.orient_process_noise())
        else:
            e = spe.BatchPoseUKF(n, precision=prec, stream="private", split_streams=split)
        e.initialize(mu, cov)
        if orient:
            e.set_orient_inputs(ring[0][0], ring[0][1])
        else:
            e.set_acceleration(ring[0][0], 0.01 * np.eye(3))
        zb, Qb = np.empty((n, 3)), np.empty((n, 3, 3))
        for j, op in enumerate(ops):
            r = j % 3
            if op == "cycle_dev":
                e.cycle_dev(0.01, model, dring[r][zi], dring[r][Qi])
            elif op in ("cycle_host", "update_host", "uniq"):
                zb[...] = ring[r][zi]; Qb[...] = ring[r][Qi]
                if op == "cycle_host":
                    e.cycle(0.01, model, zb, Qb)
                elif op == "update_host":
                    e.update(model, zb, Qb)
                else:
                    e.cycle_uniform_q(0.01, model, zb, ring[r][Qi][0])
                zb[...] = np.nan; Qb[...] = np.nan       # the caller's arrays are free at return
            elif op == "predict":
                e.predict(0.02)
            elif op == "multi":
                e.cycle_multi_dev(2, 0.01, model, z_ring, Q_ring, 3, r)
            else:
                e.state(0, 8)
        m, c, _ = e.state()
        out.append((m, c, e.status()))
        e.close()
    if not (np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and (out[0][2] == out[1][2]).all()):
        return f"split k={k} n={n} prec={prec} orient={orient} ops={ops}: split and single launches differ ({mad(out[0][0], out[1][0]):.3e})"
    if not np.isfinite(out[0][0]).all():
        return f"split k={k}: non-finite state"
    return None


def run(count=40, seed=1):
    rng = np.random.default_rng(seed)
    fails = []
    for k in range(count):
        msg = bucket_scenario(rng, k) if k % 2 == 0 else split_scenario(rng, k)
        if msg:
            fails.append(msg)
            print("FAIL", msg, flush=True)
    print(f"fuzz_round3: {count} scenarios, {len(fails)} failing (seed {seed})", flush=True)
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
