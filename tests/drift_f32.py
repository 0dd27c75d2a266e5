"""Long-run drift of the fp32 engine: GPU fp32 beside BOTH CPU oracles (fp64 and the float instantiation).

Test infrastructure (it drives the oracle): used by tests/test_gpu_f32_horizon.py and runnable as a script on
the GPU box,

    python tests/drift_f32.py [pose|orient] [filters] [cycles] [checkpoint step] [1: also the fp64 engine]
                                                                        (UKFB_LIB=... selects a diagnostic build)

Workloads are the ones bench.py times for BASELINE configs 3 and 4 (same initial states, same ring of N_RING
input sets, dt = 0.01): Pose predict on the acceleration branch + PositionMeasurement update
(PoseUKF.cpp:88-97,112-117,180-196) and OrientationState predict + body-velocity update
(OrientationUKF.cpp:12-39,65-72,79-89).  All three legs see the SAME float-rounded inputs.

At every checkpoint three distances are recorded, for the mean and for the covariance (max abs over the batch):
    gpu_o64   GPU fp32 engine      <-> fp64 oracle          (what north_star's 1e-4 is about)
    o32_o64   float oracle         <-> fp64 oracle          (what plain fp32 arithmetic of the same algorithm does)
    gpu_o32   GPU fp32 engine      <-> float oracle         (two fp32 evaluations of the same chaotic-free recursion)
If gpu_o64 stays within ~2x of o32_o64 the drift is fp32 conditioning of the workload, not a kernel defect.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_RING = 4
DT = 0.01
CHECKPOINTS = (1, 10, 50, 100, 300, 600)


def _f32(x):
    return np.asarray(x).astype(np.float32).astype(np.float64)


def _dist(a, b):
    return float(np.max(np.abs(a - b)))


def run(spe, capi, workload="pose", n=2048, cycles=600, checkpoints=CHECKPOINTS, threads=8, with_f64_engine=False, wide=False):
    """wide: the fp32 engine runs with ukfb_config.wide_arithmetic = 1 (fp32 arrays, fp64 arithmetic).
    Returns a list of rows {cycle, gpu_o64: (mu, cov), o32_o64: (mu, cov), gpu_o32: (mu, cov)[, gpu64_o64: (mu, cov)]}."""
    import torch
    sy = spe.synth
    orient = workload == "orient"
    dev = torch.device("cuda", 0)
    if orient:
        mu, cov = sy.orient_initial(n)
        ring = [sy.orient_cycle_inputs(n, k, mu[:, :4]) for k in range(N_RING)]          # gyro, acc, z, Q
        ring = [tuple(_f32(x) for x in r) for r in ring]
        Rn = sy.orient_process_noise()

        def make(prec):
            e = spe.BatchOrientationUKF(n, sy.ORIENT_TAU, sy.ORIENT_TAU, sy.ORIENT_LATITUDE, precision=prec,
                                        **({"wide_arithmetic": 1} if (wide and prec == spe.F32) else {}))
            e.set_process_noise(Rn)
            return e
    else:
        mu, cov = sy.pose_initial(n)
        ring = [sy.pose_cycle_inputs(n, k, mu[:, :3]) for k in range(N_RING)]             # acc, z, Q
        ring = [tuple(_f32(x) for x in r) for r in ring]
        Rn = sy.pose_default_process_noise()
        acc_cov = 0.01 * np.eye(3)

        def make(prec):
            e = spe.BatchPoseUKF(n, precision=prec, **({"wide_arithmetic": 1} if (wide and prec == spe.F32) else {}))
            e.set_acceleration(None, acc_cov)
            return e
    mu, cov = _f32(mu), _f32(cov)
    engines = [(make(spe.F32), torch.float32)]
    if with_f64_engine:
        engines.append((make(spe.F64), torch.float64))
    dring = []
    for eng, tdt in engines:
        eng.initialize(mu, cov)
        dring.append([tuple(torch.from_numpy(np.ascontiguousarray(x.reshape(n, -1))).to(dev, tdt) for x in r) for r in ring])
    torch.cuda.synchronize()
    earth = engines[0][0].earth_rotation if orient else None

    def oracle_cycle(m, c, k, prec):
        if orient:
            gyro, acc, z, Q = ring[k % N_RING]
            m, c, s1 = capi.orient_predict(m, c, Rn, acc, gyro, sy.ORIENT_TAU, sy.ORIENT_TAU, earth, DT, prec=prec, threads=threads)
            m, c, s2 = capi.orient_update(m, c, z, Q, prec=prec, threads=threads)
        else:
            acc, z, Q = ring[k % N_RING]
            m, c, s1 = capi.pose_predict(m, c, Rn, acc, acc_cov, DT, prec=prec, threads=threads)
            m, c, s2 = capi.pose_update(m, c, 0, z, Q, prec=prec, threads=threads)
        return m, c, int(np.bitwise_or.reduce(s1 | s2))

    def gpu_cycle(i, k):
        eng = engines[i][0]
        r = dring[i][k % N_RING]
        if orient:
            eng.bind_orient_inputs_dev(r[0], r[1])
            eng.cycle_dev(DT, spe.MEAS_ORIENT_BODYVEL3, r[2], r[3])
        else:
            eng.bind_acceleration_dev(r[0])
            eng.cycle_dev(DT, spe.MEAS_POS3, r[1], r[2])

    m64, c64 = mu.copy(), cov.copy()
    m32, c32 = mu.copy(), cov.copy()
    rows = []
    st64 = st32 = 0
    marks = sorted(set(c for c in checkpoints if c <= cycles) | {cycles})
    for k in range(cycles):
        for i in range(len(engines)):
            gpu_cycle(i, k)
        m64, c64, s = oracle_cycle(m64, c64, k, 0)
        st64 |= s
        m32, c32, s = oracle_cycle(m32, c32, k, 1)
        st32 |= s
        if k + 1 in marks:
            mg, cg, _ = engines[0][0].state()
            row = {"cycle": k + 1,
                   "gpu_o64": (_dist(mg, m64), _dist(cg, c64)),
                   "o32_o64": (_dist(m32, m64), _dist(c32, c64)),
                   "gpu_o32": (_dist(mg, m32), _dist(cg, c32)),
                   "status": (engines[0][0].status_summary(), st64, st32)}
            if with_f64_engine:
                mg2, cg2, _ = engines[1][0].state()
                row["gpu64_o64"] = (_dist(mg2, m64), _dist(cg2, c64))
            rows.append(row)
    for eng, _ in engines:
        eng.close()
    return rows


def fmt(rows):
    out = ["cycle   gpu32-o64 (mean, cov)     o32-o64 (mean, cov)       gpu32-o32 (mean, cov)      ratio(mean, cov)   status(gpu,o64,o32)"]
    for r in rows:
        g, o, x = r["gpu_o64"], r["o32_o64"], r["gpu_o32"]
        rm = g[0] / o[0] if o[0] > 0 else float("inf")
        rc = g[1] / o[1] if o[1] > 0 else float("inf")
        extra = f"   gpu64-o64 {r['gpu64_o64'][0]:.2e} {r['gpu64_o64'][1]:.2e}" if "gpu64_o64" in r else ""
        out.append(f"{r['cycle']:5d}   {g[0]:.3e} {g[1]:.3e}     {o[0]:.3e} {o[1]:.3e}     {x[0]:.3e} {x[1]:.3e}     "
                   f"{rm:6.2f} {rc:6.2f}      {r['status']}{extra}")
    return "\n".join(out)


if __name__ == "__main__":
    import torch  # noqa: F401  (before the engine library: one HIP runtime per process)
    import slam_pose_estimation_amd as spe
    from oracle import capi
    wl = sys.argv[1] if len(sys.argv) > 1 else "pose"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    cyc = int(sys.argv[3]) if len(sys.argv) > 3 else 600
    step = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    both = len(sys.argv) > 5 and sys.argv[5] == "1"
    print(f"# {wl} fp32, {n} filters, {cyc} cycles, lib={os.environ.get('UKFB_LIB', 'product')}")
    marks = tuple(range(step, cyc + 1, step)) if step > 0 else CHECKPOINTS
    print(fmt(run(spe, capi, wl, n, cyc, checkpoints=marks, threads=max(1, min(16, capi.max_threads())),
                  with_f64_engine=both)), flush=True)
