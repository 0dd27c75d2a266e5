#!/usr/bin/env python3
"""bench.py -- UKF predict+update filter-cycles/s on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one fused launch of PoseUKF::predictionStep(dt) on the acceleration branch followed by
integrateMeasurement(PositionMeasurement) for EVERY filter of the batch.  The
workload is the configuration the metric is quoted on: 1 048 576 PoseWithVelocity filters per GPU, computed
in fp64 (the reference's own arithmetic type).  Filters are independent, so ranks own disjoint filter ranges
and there is no data-path collective; RCCL is used once, after the timed region, to gather the means.
Default scaling is WEAK (every rank runs 1 048 576 filters, value = all filters x steps / time);
--scaling strong shards 1 048 576 filters in total over the ranks instead.  Inputs (acceleration,
measurement, measurement covariance) are resident in HBM before the clock starts.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     -- HBM roofline of the fused kernel from ALGORITHMIC bytes
  cpu_baseline -- the CPU oracle (a from-scratch port, kind "port") timed on this host's cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOTAL_FILTERS = 1_048_576
DT = 0.01
ALG_SCALARS_POSE = 2 * 157 + 15  # read+write (13 + 144) + acc 3 + z 3 + Q 9
ALG_SCALARS_ORIENT = 2 * 183 + 18  # read+write (14 + 169) + gyro 3 + acc 3 + z 3 + Q 9
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


class _DevArray:
    """Zero-copy view of engine-owned device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--filters", type=int, default=TOTAL_FILTERS,
                    help="filters per GPU (weak scaling, default) or in total (--scaling strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank runs --filters filters (N x --filters in total); strong: --filters "
                         "filters are sharded evenly over the ranks (the literal '1 M filters on 1/2/4/8 GPUs')")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--lanes-per-filter", type=int, default=0, help="16/32/64 (0: engine default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["pose", "pose-mixed", "orient"], default="pose",
                    help="pose: the headline metric (default). pose-mixed: BASELINE config 5 (per-filter model id over "
                         "the 9 Pose models, 25 %% inactive). orient: config 4 (OrientationState predict + body-velocity update)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal of "
                    "the N>1 plumbing on a box with fewer GPUs than ranks)")
    ap.add_argument("--cpu-sample-filters", type=int, default=65536)
    ap.add_argument("--cpu-sample-seconds", type=float, default=10.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def cpu_baseline(args):
    """Time the oracle (reported baseline only; never the product path)."""
    from oracle import capi
    import slam_pose_estimation_amd as spe
    n = args.cpu_sample_filters
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, capi.max_threads(), 16))
    mu, cov = spe.synth.pose_initial(n)
    R = spe.synth.pose_default_process_noise()
    acc_cov = 0.01 * np.eye(3)
    prec = 0 if args.precision == "f64" else 1
    inputs = [spe.synth.pose_cycle_inputs(n, k, mu[:, :3]) for k in range(4)]
    # untimed touch
    capi.pose_predict(mu[:64], cov[:64], R, inputs[0][0][:64], acc_cov, DT, prec=prec, threads=1)
    cycles, el = 0, 0.0
    t0 = time.perf_counter()
    while el < args.cpu_sample_seconds and cycles < 2000:   # bounded sample: about 10 s of all-core CPU work
        acc, z, Q = inputs[cycles % 4]
        mu, cov, _ = capi.pose_predict(mu, cov, R, acc, acc_cov, DT, prec=prec, threads=threads)
        mu, cov, _ = capi.pose_update(mu, cov, 0, z, Q, prec=prec, threads=threads)
        cycles += 1
        el = time.perf_counter() - t0
    return {"value": n * cycles / el, "unit": "filter-cycles/s", "cores": threads, "kind": "port",
            "sample": f"{n} PoseWithVelocity filters x {cycles} predict(acc)+position-update cycles, "
                      f"{args.precision}, oracle/ukf_oracle.hpp with OpenMP over filters, {el:.2f} s"}


def load_traffic(kernel_name, filters_per_launch):
    """HBM bytes per launch from a committed rocprofv3 --pmc pass, if one matches this workload."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        for t in doc.get("entries", []):
            if t.get("kernel") == kernel_name and int(t.get("filters_per_launch", -1)) == int(filters_per_launch):
                return float(t["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def main():
    args = parse()
    import torch
    import slam_pose_estimation_amd as spe

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 through torch.distributed.run (one process per GPU)")
    dist = None
    # UKFB_BENCH_FORCE_DIST=1 under torchrun: rehearsal of the N>1 plumbing (RCCL init, barrier, all_reduce,
    # gather) with a single rank on a one-GPU box
    if world > 1 or ("RANK" in os.environ and os.environ.get("UKFB_BENCH_FORCE_DIST")):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dev_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    prec = spe.F64 if args.precision == "f64" else spe.F32
    tdtype = torch.float64 if prec == spe.F64 else torch.float32
    total = args.filters * world if args.scaling == "weak" else args.filters
    first, per = spe.shard_range(total, world, rank)

    # ---- build the shard (host generation in chunks, then resident in HBM)
    CH = 131072
    n_ring = 4
    orient = args.workload == "orient"
    S = 14 if orient else 13
    if orient:
        sy = spe.synth
        eng = spe.BatchOrientationUKF(per, sy.ORIENT_TAU, sy.ORIENT_TAU, sy.ORIENT_LATITUDE, precision=prec,
                                      device=dev.index, lanes_per_filter=args.lanes_per_filter)
        eng.set_process_noise(sy.orient_process_noise())
    else:
        eng = spe.BatchPoseUKF(per, precision=prec, device=dev.index, lanes_per_filter=args.lanes_per_filter)
    acc_d = [torch.empty((per, 3), dtype=tdtype, device=dev) for _ in range(n_ring)]
    gyr_d = [torch.empty((per, 3), dtype=tdtype, device=dev) for _ in range(n_ring)] if orient else None
    z_d = [torch.empty((per, 3), dtype=tdtype, device=dev) for _ in range(n_ring)]
    Q_d = [torch.empty((per, 9), dtype=tdtype, device=dev) for _ in range(n_ring)]
    m_d = [torch.empty((per,), dtype=torch.int32, device=dev) for _ in range(n_ring)] if args.workload == "pose-mixed" else None
    for lo in range(0, per, CH):
        hi = min(per, lo + CH)
        if orient:
            mu, cov = spe.synth.orient_initial(hi - lo, first=first + lo)
        else:
            mu, cov = spe.synth.pose_initial(hi - lo, first=first + lo)
        eng.initialize(mu, cov, first=lo)
        for k in range(n_ring):
            if orient:
                gyro, acc, z, Q = spe.synth.orient_cycle_inputs(hi - lo, k, mu[:, :4], first=first + lo)
                gyr_d[k][lo:hi] = torch.from_numpy(gyro).to(dev, tdtype)
            else:
                acc, z, Q = spe.synth.pose_cycle_inputs(hi - lo, k, mu[:, :3], first=first + lo,
                                                        random_q=args.workload == "pose-mixed")
                if args.workload == "pose-mixed":
                    models = spe.synth.pose_mixed_models(hi - lo, k, first=first + lo)
                    z = spe.synth.pose_measurement_for_model(mu, models, z - mu[:, :3])
                    m_d[k][lo:hi] = torch.from_numpy(models).to(dev)
            acc_d[k][lo:hi] = torch.from_numpy(acc).to(dev, tdtype)
            z_d[k][lo:hi] = torch.from_numpy(z).to(dev, tdtype)
            Q_d[k][lo:hi] = torch.from_numpy(Q.reshape(-1, 9)).to(dev, tdtype)
    if not orient:
        eng.set_acceleration(None, 0.01 * np.eye(3))
    torch.cuda.synchronize()

    def step(k):
        r = k % n_ring
        if orient:
            eng.bind_orient_inputs_dev(gyr_d[r], acc_d[r])
            eng.cycle_dev(DT, spe.MEAS_ORIENT_BODYVEL3, z_d[r], Q_d[r])
        else:
            eng.bind_acceleration_dev(acc_d[r])
            eng.cycle_dev(DT, spe.MEAS_POS3, z_d[r], Q_d[r], meas_model_dev=m_d[r] if m_d else None)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed pre-roll: the FIRST burst of queued launches in a process is reported complete ~50 ms late by the
    # runtime in about one process out of three (GPU timestamps show the kernels back to back; later bursts
    # never, tools/sync_latency.py and DESIGN.md section 5).  A short burst (up to 64 launches)
    # absorbs that one-time event before the W warm-up steps and the K timed steps.
    for k in range(min(args.steps, 64)):
        step(k)
    fence()
    for k in range(args.warmup):
        step(k)
    fence()
    eng.timer_begin()           # HIP events on the stream the kernel is launched on
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    t_launch = time.perf_counter()
    kernel_ms_total = eng.timer_end()
    t_event = time.perf_counter()
    fence()
    elapsed = time.perf_counter() - t0
    if os.environ.get("UKFB_BENCH_DEBUG"):
        print("launch %.2f ms, event sync %.2f ms, fence %.2f ms, gpu %.2f ms" % (
            (t_launch - t0) * 1e3, (t_event - t_launch) * 1e3, (t0 + elapsed - t_event) * 1e3, kernel_ms_total), file=sys.stderr)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status_or = eng.status_summary()

    # ---- result gather over RCCL/xGMI (outside the timed region; means only)
    gather_ms = None
    if dist is not None:
        mu_ptr, _, _ = eng.device_views()
        mu_local = torch.as_tensor(_DevArray(mu_ptr, (per, S), "<f8" if prec == spe.F64 else "<f4"), device=dev)
        if args.backend != "nccl":
            mu_local = mu_local.cpu()
        fence()
        g0 = time.perf_counter()
        gathered = spe.gather_means(mu_local, total, world, dist)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert gathered.shape == (total, S)

    info = eng.last_launch_info()
    if rank == 0:
        tsize = 8 if prec == spe.F64 else 4
        value = total * args.steps / elapsed
        alg_bytes_launch = (ALG_SCALARS_ORIENT if orient else ALG_SCALARS_POSE) * tsize * per
        kernel_ms = kernel_ms_total / args.steps
        achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "UKF predict+update filter-cycles/s, " + ("OrientationState" if orient else "PoseWithVelocity") + " filters",
            "value": value,
            "unit": "filter-cycles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": (f"{total} OrientationState UKF filters, fused predict(gyro+acc, dt=0.01)"
                                    f"+body-velocity update per step, {args.precision}, {per} filters per GPU" if orient else
                                    f"{total} PoseWithVelocity UKF filters, fused predict(acc branch, dt=0.01)+"
                                    + ("per-filter measurement model (9 models, 25 % inactive)" if args.workload == "pose-mixed"
                                       else "PositionMeasurement") + f" update per step, {args.precision}, {per} filters per GPU"),
                       "filters": total, "filters_per_gpu": per,
                       "lanes_per_filter": 64 // max(1, info["filters_per_workgroup"]),
                       "parallelism": f"filter-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": load_traffic(info["kernel"], per),
                         "kernel": info["kernel"], "kernel_ms_per_launch": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "lds_bytes_per_workgroup": info["lds_bytes"]},
            "status_or": status_or,
            "gather_ms": gather_ms,
        }
        if not args.no_cpu_baseline and world == 1 and args.workload == "pose":
            out["cpu_baseline"] = cpu_baseline(args)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
