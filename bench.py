#!/usr/bin/env python3
"""bench.py -- UKF predict+update filter-cycles/s on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

Self-launching: with N > 1 and no RANK in the environment this process starts N children (one per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous) BEFORE anything touches the GPU, waits for them
and relays rank 0's JSON line.  Under `python -m torch.distributed.run ... bench.py --gpus N` (RANK already set)
it is one of the ranks.  N = 1 runs the same rank function in-process.

One "step" = one fused launch of PoseUKF::predictionStep(dt) on the acceleration branch followed by
integrateMeasurement(PositionMeasurement) for EVERY filter of this rank's shard.
The workload is the configuration the metric is quoted on: 1 048 576 PoseWithVelocity filters, fp64 (the
reference's own arithmetic type).  Default scaling is STRONG, the metric's "1 M filters on 1/2/4/8 GPUs":
the filters are sharded over the ranks (131 072 per GPU at N = 8, BASELINE config 3's layout);
`--scaling weak` gives every rank --filters filters instead.  Filters are independent
(UnscentedKalmanFilter.hpp:150), so there is no data-path collective; RCCL is used once, after the timed
region, to gather the means.  Inputs are resident in HBM before the clock starts.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     -- HBM roofline of the fused kernel from ALGORITHMIC bytes, the measured
                  PMC traffic of this workload, and a VALU-issue roofline (`valu`) from the committed PMC pass
  cpu_baseline -- the CPU oracle (a from-scratch port, kind "port") built -O3 -march=native on this host and
                  timed on all hardware threads and on one core
  parity       -- max |GPU - oracle| on a sample of filters after all launches of this run (same inputs)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOTAL_FILTERS = 1_048_576
DT = 0.01
ALG_SCALARS_POSE = 2 * 157 + 15    # read+write (13 + 144) + acc 3 + z 3 + Q 9
ALG_SCALARS_ORIENT = 2 * 183 + 18  # read+write (14 + 169) + gyro 3 + acc 3 + z 3 + Q 9
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
N_SIMD = 1024                      # 256 CUs x 4 SIMDs
N_RING = 4                         # input sets the steps cycle through
SUSTAINED_STEPS = 500
BURST_STEPS = 20
RECENT_CYCLES = 16                 # window of the second in-run parity check
MULTI_CYCLES = 8                   # cycles per launch of the extra multi-cycle region (ukfb_cycle_multi_dev)
TOL = {"f64": 1e-9, "f32": 1e-4}   # north_star
F32_HORIZON = {"pose": 450, "orient": 150}   # cycles the fp32 engines stay within 1e-4 of the fp64 oracle (tests/test_gpu_f32_horizon.py)


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
                    help="filters in total (--scaling strong, default) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --filters filters are sharded evenly over the ranks (the metric's '1 M filters on "
                         "1/2/4/8 GPUs'); weak: every rank runs --filters filters (N x --filters in total)")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--wide-arithmetic", type=int, choices=[0, 1], default=0,
                    help="with --precision f32: 1 = fp32 arrays in HBM, every instruction of the cycle in fp64 "
                         "(ukfb_config.wide_arithmetic); the line then reports dtype f64 and config.hbm_format f32")
    ap.add_argument("--lanes-per-filter", type=int, default=0, help="16/32/64 (0: engine default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extra-regions", action="store_true", help="skip the sustained / burst kernel timings")
    ap.add_argument("--clock-warmup-seconds", type=float, default=0.4,
                    help="untimed launches before the W warm-up steps (state restored afterwards) so that the GPU clocks have settled")
    ap.add_argument("--clock-warmup-cycles", type=int, default=0,
                    help="> 0: the untimed pre-roll is exactly this many cycles instead of a time span (counter passes: the set of "
                         "launches a profiler sees must not depend on how fast the box is -- the workload is not stationary)")
    ap.add_argument("--workload", choices=["pose", "pose-cv", "pose-mixed", "orient"], default="pose",
                    help="pose: the headline metric (default). pose-cv: the same without a latched acceleration -- the "
                         "constant-velocity branch of predictionStepImpl (PoseUKF.cpp:195: rotated noise, scaled by dt), "
                         "SURVEY 8(d)'s secondary run. pose-mixed: BASELINE config 5 (per-filter model id over "
                         "the 9 Pose models, 25 %% inactive). orient: config 4 (OrientationState predict + body-velocity update)")
    ap.add_argument("--cycles-per-launch", type=int, default=1,
                    help="C > 1: the K timed cycles run as launches of C cycles each (ukfb_cycle_multi_dev: the filters stay "
                         "in LDS between the cycles of a launch); a step is still one predict+update cycle")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal of "
                    "the N>1 plumbing on a box with fewer GPUs than ranks)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="rehearse launch / rendezvous / sharding / barrier / gather with NO engine and NO timing "
                         "(CPU boxes; prints value null)")
    ap.add_argument("--cpu-sample-filters", type=int, default=65536)
    ap.add_argument("--cpu-sample-seconds", type=float, default=9.0,
                    help="CPU time of the three all-core samples together (the thread-ladder probes and the one-core sample come on top)")
    ap.add_argument("--parity-sample", type=int, default=4096)
    ap.add_argument("--inputs", choices=["ring", "tracking"], default="ring",
                    help="ring (default, the figures of every round): the steps cycle through 4 fixed input sets drawn around the "
                         "INITIAL position.  tracking (pose / pose-cv): every step's position fix is regenerated on the device as "
                         "current mean position + that step's noise (one elementwise kernel per step on the engine's stream, inside "
                         "the timed region), i.e. the innovation stays at noise level however long the run is (SURVEY 8(d))")
    ap.add_argument("--launcher", choices=["procs", "group"], default="procs",
                    help="procs (default, the measured contract): one process per GPU, torch.distributed over RCCL. group: ONE "
                         "process drives all N GPUs through the C-ABI device group (ukfb_group_*: one engine and stream per "
                         "device, RCCL all-gather of the means from C) -- the shape a C++ host uses")
    ap.add_argument("--group-devices", default="",
                    help="--launcher group: comma-separated HIP device per shard (default 0..N-1); naming a device twice "
                         "rehearses N shards on fewer GPUs (the gather is then skipped: RCCL needs one rank per device)")
    ap.add_argument("--split-streams", type=int, default=1, choices=[0, 1],
                    help="1 (engine default): launches over 16 384 ... 262 143 filters run as two halves on two streams "
                         "(ukfb_config.split_streams); 0: one launch on one stream (same-box A/B)")
    ap.add_argument("--bucket-models", type=int, default=1, choices=[0, 1],
                    help="pose-mixed: 1 (engine default) groups the filters by update class on the device before the launch, "
                         "0 launches in filter order (ukfb_config.bucket_models; same-box A/B)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(args):
    """One child process per rank, started before this process has made any GPU call (it never does)."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "UKFB_BENCH_CHILD": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    rc = 0
    out0 = ""
    try:
        pending = set(range(args.gpus))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if r == 0:
                    out0 = procs[0].stdout.read()
                if code != 0:
                    rc = rc or code
                    for q in pending:      # a dead rank would leave the others in a barrier forever
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------- CPU legs
def _threads_available():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _cpu_quota():
    """CPUs' worth of time the container's cgroup may use per period (None: unlimited / unknown)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(args):
    """Time the oracle (reported baseline only; never the product path): all hardware threads, then one core."""
    import numpy as np
    from oracle import capi
    import slam_pose_estimation_amd as spe
    native = capi.native_lib()
    build = "-O3 -march=native -fopenmp (built on this host)" if native is not None else \
        "-O3 -march=x86-64-v3 -ffp-contract=off (native build failed)"
    prec = 0 if args.precision == "f64" else 1
    R = spe.synth.pose_default_process_noise()
    acc_cov = 0.01 * np.eye(3)

    def run(n, threads, seconds):
        mu, cov = spe.synth.pose_initial(n)
        inputs = [spe.synth.pose_cycle_inputs(n, k, mu[:, :3]) for k in range(N_RING)]
        capi.pose_predict(mu[:64], cov[:64], R, inputs[0][0][:64], acc_cov, DT, prec=prec, threads=1)   # untimed touch
        cycles, el = 0, 0.0
        t0 = time.perf_counter()
        while el < seconds and cycles < 2000:
            acc, z, Q = inputs[cycles % N_RING]
            mu, cov, _ = capi.pose_predict(mu, cov, R, acc, acc_cov, DT, prec=prec, threads=threads)
            mu, cov, _ = capi.pose_update(mu, cov, 0, z, Q, prec=prec, threads=threads)
            cycles += 1
            el = time.perf_counter() - t0
        return n * cycles / el, cycles, el

    with capi.using(native if native is not None else capi.lib()):
        hw = max(1, min(_threads_available(), capi.max_threads()))
        n_all = args.cpu_sample_filters
        # "All hardware threads" is not automatically the fastest: a container whose CPU share (cgroup quota or
        # cpuset pressure) is below its visible threads is throttled when every thread spins.  Probes over a ladder of
        # thread counts -- at the SAME batch size as the final samples -- pick the count; every probe is reported.
        quota = _cpu_quota()
        ladder = sorted({t for t in (hw, hw // 2, hw // 4, 64, 32, 16, 8, int((quota or 0) + 0.999)) if 1 < t <= hw} | {hw})
        sweep = {}
        for t in ladder:
            v, _, _ = run(n_all, t, max(0.5, args.cpu_sample_seconds / 6))
            sweep[t] = v
        t_best = max(sweep, key=sweep.get)
        # the figure: MEDIAN of three full samples at that thread count (one sample under a cgroup quota is not a stable number)
        samples = [run(n_all, t_best, args.cpu_sample_seconds / 3) for _ in range(3)]
        vals = sorted(v for v, _, _ in samples)
        v_med = vals[1]
        c_b, e_b = sum(c for _, c, _ in samples), sum(e for _, _, e in samples)
        n_one = max(64, min(n_all, 4096))
        v_one, c_one, e_one = run(n_one, 1, args.cpu_sample_seconds / 2)
    spread = (vals[-1] - vals[0]) / v_med if v_med > 0 else 0.0
    probe_dev = abs(sweep[t_best] - v_med) / v_med if v_med > 0 else 0.0
    return {"value": v_med, "unit": "filter-cycles/s", "cores": t_best, "kind": "port",
            "samples": vals, "probe_at_chosen_threads": sweep[t_best],
            # probe and median further apart than 15 %, or the three samples spread by more than 15 %: say so
            "unstable": bool(probe_dev > 0.15 or spread > 0.15),
            "hardware_threads": hw, "cgroup_cpu_quota": quota,
            "thread_sweep": {str(t): v for t, v in sorted(sweep.items())},
            "single_core": {"value": v_one, "cores": 1,
                            "sample": f"{n_one} filters x {c_one} cycles, {e_one:.2f} s"},
            "build": build,
            "sample": f"median of 3 samples, each {n_all} PoseWithVelocity filters x ~{c_b // 3} predict(acc)+position-update cycles, "
                      f"{args.precision}, oracle/ukf_oracle.hpp with OpenMP over filters on {t_best} threads "
                      f"({hw} hardware threads visible, cgroup CPU quota {quota}), {e_b:.2f} s in all; probes of the thread ladder at "
                      f"the same batch size"}


def parity_check(args, spe, eng, first, sample, cycles, orient, start=None):
    """GPU state of filters [first, first+sample) of this rank against an oracle replay of the SAME launches.
    start = None: all `cycles` fused cycles of this run, from the synthetic initial state.
    start = (mu, cov, k0): the last cycles only, from a GPU state downloaded k0 cycles into the run (the fp32
    engine drifts from an fp64 replay over hundreds of cycles of a filter whose orientation is unobserved; the
    recent window shows the per-cycle agreement in the regime the timed region ran in)."""
    import numpy as np
    from oracle import capi
    sy = spe.synth
    f32 = (lambda x: x.astype(np.float32).astype(np.float64)) if args.precision == "f32" else (lambda x: x)
    threads = max(1, min(_threads_available(), capi.max_threads(), 16))   # a GPU box grants about 16 CPUs' worth of time
    m_g, c_g, _ = eng.state(0, sample)
    k0 = 0 if start is None else start[2]

    def replay(oprec):
        """oracle replay of cycles k0..cycles-1; oprec 0 = the fp64 oracle, 1 = its float instantiation"""
        if orient:
            mu, cov = sy.orient_initial(sample, first=first)
            ring = [sy.orient_cycle_inputs(sample, k, mu[:, :4], first=first) for k in range(N_RING)]
            m_o, c_o = (f32(mu), f32(cov)) if start is None else (start[0], start[1])
            for k in range(k0, cycles):
                gyro, acc, z, Q = ring[k % N_RING]
                m_o, c_o, _ = capi.orient_predict(m_o, c_o, sy.orient_process_noise(), f32(acc), f32(gyro), sy.ORIENT_TAU,
                                                  sy.ORIENT_TAU, eng.earth_rotation, DT, prec=oprec, threads=threads)
                m_o, c_o, _ = capi.orient_update(m_o, c_o, f32(z), f32(Q), prec=oprec, threads=threads)
            return m_o, c_o
        mixed = args.workload == "pose-mixed"
        mu, cov = sy.pose_initial(sample, first=first)
        ring = []
        for k in range(N_RING):
            acc, z, Q = sy.pose_cycle_inputs(sample, k, mu[:, :3], first=first, random_q=mixed)
            models = 0
            if mixed:
                models = sy.pose_mixed_models(sample, k, first=first)
                z = sy.pose_measurement_for_model(mu, models, z - mu[:, :3])
            ring.append((acc, z, Q, models))
        R = sy.pose_default_process_noise()
        acc_cov = 0.01 * np.eye(3)
        m_o, c_o = (f32(mu), f32(cov)) if start is None else (start[0], start[1])
        for k in range(k0, cycles):
            acc, z, Q, models = ring[k % N_RING]
            if args.inputs == "tracking":   # the fix of this step: the mean position BEFORE the step + the step's noise
                z = f32(m_o[:, :3] + f32(z - mu[:, :3]))
            m_o, c_o, _ = capi.pose_predict(m_o, c_o, R, None if args.workload == "pose-cv" else f32(acc), acc_cov, DT,
                                            prec=oprec, threads=threads)
            m_o, c_o, _ = capi.pose_update(m_o, c_o, models, f32(z), f32(Q), prec=oprec, threads=threads)
        return m_o, c_o

    m_o, c_o = replay(0)
    em, ec = float(np.abs(m_g - m_o).max()), float(np.abs(c_g - c_o).max())
    tol = TOL[args.precision]
    what = (f"after {cycles} fused cycles of this run (warm-up + timed; the clock pre-roll is undone)" if start is None else
            f"cycles {k0}..{cycles - 1} of this run, from the GPU state downloaded before them")
    out = {"max_abs_mu": em, "max_abs_cov": ec, "tol": tol, "ok": bool(em <= tol and ec <= tol),
           "sample": f"filters {first}..{first + sample - 1} {what}, GPU state vs oracle/ukf_oracle.hpp (fp64) replay "
                     f"of the same input ring"}
    if args.precision == "f32":
        # the same replay by the FLOAT instantiation of the oracle: how far plain fp32 arithmetic of the same algorithm
        # leaves the fp64 one over these cycles (tests/test_gpu_f32_horizon.py).  The fp32 engine is held to <= 2x that.
        m_f, c_f = replay(1)
        fm, fc = float(np.abs(m_f - m_o).max()), float(np.abs(c_f - c_o).max())
        out["float_oracle_vs_fp64"] = {"max_abs_mu": fm, "max_abs_cov": fc}
        out["gpu_vs_float_oracle"] = {"max_abs_mu": float(np.abs(m_g - m_f).max()), "max_abs_cov": float(np.abs(c_g - c_f).max())}
        out["explained_by_fp32_arithmetic"] = bool(em <= max(2 * fm, 2e-6) and ec <= max(2 * fc, 2e-6))
        if not getattr(args, "wide_arithmetic", 0):
            # what fp32 ARITHMETIC holds on these workloads (tests/test_gpu_f32_horizon.py); past it the recursion's own fp32
            # rounding exceeds 1e-4 whatever the kernel does (tests/study_f32_mixed.py, profiles/r04_f32_mixed_ab.txt)
            hz = F32_HORIZON["orient" if orient else "pose"]
            out["horizon_cycles"] = hz
            out["within_horizon"] = bool(cycles - k0 <= hz)
            out["sample"] += (f"; fp32 arithmetic holds north_star's 1e-4 on this workload for {hz} cycles from a common state -- this check "
                              f"spans {cycles - k0}" + ("" if cycles - k0 <= hz else
                              " (beyond it: the distance is fp32 rounding of the recursion itself, see float_oracle_vs_fp64; "
                              "--wide-arithmetic 1 keeps the fp32 HBM format and holds 1e-4 over the whole run)"))
    return out


def running_lib_identity():
    """(first 16 hex digits of the SHA-256 of the engine library this process loaded, the commit it was built from)"""
    import hashlib
    import slam_pose_estimation_amd as spe
    path = spe.engine.LIB_PATH
    try:
        with open(path, "rb") as fh:
            sha = hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        sha = None
    head = None
    try:
        with open(os.path.join(os.path.dirname(path), "BUILD_INFO.json")) as fh:
            head = json.load(fh).get("git_head")
    except Exception:
        pass
    return sha, head


def select_profile_entry(doc, kernel_name, filters_per_launch, cycles_per_launch, running_sha):
    """The committed rocprofv3 --pmc entry for this kernel, launch size and launch shape -- IF it was measured on the
    binary that is running.  Returns (entry or None, exact launch size?, source) where source says which file entry was
    looked at and whether its library hash equals the running one: counters replayed from profiles/ describe a kernel
    only as long as the kernel has not been rebuilt (round-2 verdict, weak point 6)."""
    best, exact = None, False
    for t in (doc or {}).get("entries", []):
        if t.get("kernel") != kernel_name or abs(float(t.get("cycles_per_launch", 1.0)) - float(cycles_per_launch)) > 1e-9:
            continue
        if int(t.get("filters_per_launch", -1)) == int(filters_per_launch):
            best, exact = t, True
            break
        best = best or t
    src = {"lib_sha16": best.get("lib_sha16") if best else None, "build_head": best.get("build_head") if best else None,
           "running_lib_sha16": running_sha,
           "matches_running_lib": bool(best is not None and running_sha is not None and best.get("lib_sha16") == running_sha)}
    return (best if src["matches_running_lib"] else None), exact, src


def load_profile_entry(name, kernel_name, filters_per_launch, cycles_per_launch=1.0, running_sha=None):
    """Committed rocprofv3 --pmc results (profiles/<name>): see select_profile_entry."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            doc = json.load(fh)
    except Exception:
        doc = None
    e, exact, src = select_profile_entry(doc, kernel_name, filters_per_launch, cycles_per_launch, running_sha)
    src["file"] = "profiles/" + name
    return e, exact, src


# ---------------------------------------------------------------------------------------------- one rank
def run_rank(args):
    import numpy as np
    import torch
    import slam_pose_estimation_amd as spe

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    # a process group exists whenever a launcher set RANK (also with one rank: rehearsal of the N>1 plumbing)
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("UKFB_BENCH_FORCE_DIST"))
    have_gpu = (not args.plumbing_only) and torch.cuda.device_count() > 0
    if not have_gpu and not args.plumbing_only:
        raise SystemExit("bench.py: no HIP device visible -- the engine has no CPU path (use --plumbing-only to "
                         "rehearse the multi-rank plumbing on a CPU box)")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if have_gpu else 0
    if have_gpu:
        torch.cuda.set_device(dev_index)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a bounded rendezvous: a rank that never arrives must fail the run, not hang it
        limit = datetime.timedelta(seconds=300)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index), timeout=limit)
        else:
            dist.init_process_group(backend=args.backend, timeout=limit)
    dev = torch.device("cuda", dev_index) if have_gpu else torch.device("cpu")
    coll_dev = dev if (args.backend == "nccl" and have_gpu) else torch.device("cpu")

    prec = spe.F64 if args.precision == "f64" else spe.F32
    wide = 1 if (args.wide_arithmetic and prec == spe.F32) else 0
    tdtype = torch.float64 if prec == spe.F64 else torch.float32
    total = args.filters * world if args.scaling == "weak" else args.filters
    first, per = spe.shard_range(total, world, rank)
    orient = args.workload == "orient"
    S = 14 if orient else 13
    tracking = args.inputs == "tracking"
    if tracking and (args.workload not in ("pose", "pose-cv") or args.cycles_per_launch != 1):
        raise SystemExit("--inputs tracking applies to --workload pose / pose-cv with one cycle per launch")

    def fence():
        if dist is not None:
            dist.barrier()
        if have_gpu:
            torch.cuda.synchronize()

    def rank_stats(x):
        """(min, max) over ranks of a per-rank float."""
        if dist is None:
            return x, x
        t = torch.zeros(world, dtype=torch.float64, device=coll_dev)
        t[rank] = x
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.min().item()), float(t.max().item())

    if args.plumbing_only:
        # launch, rendezvous, sharding, barrier, all-reduce and the ragged gather -- no engine, no number
        fence()
        lo, hi = rank_stats(float(rank))
        mu_local = torch.full((per, S), float(rank), dtype=tdtype)
        gathered = spe.gather_means(mu_local, total, world, dist)
        assert gathered.shape == (total, S) and (lo, hi) == (0.0, float(world - 1))
        counts = [spe.shard_range(total, world, r)[1] for r in range(world)]
        assert all(float(gathered[sum(counts[:r]), 0]) == float(r) for r in range(world) if counts[r])
        if rank == 0:
            print(json.dumps({"metric": "UKF predict+update filter-cycles/s", "value": None, "unit": "filter-cycles/s",
                              "n_gpus": world, "plumbing_only": True, "backend": args.backend, "scaling": args.scaling,
                              "config": {"workload": "none (rank plumbing rehearsal)", "filters": total,
                                         "filters_per_gpu": per}}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    # ---- build the shard (host generation in chunks, then resident in HBM)
    CH = 131072
    if orient:
        sy = spe.synth
        eng = spe.BatchOrientationUKF(per, sy.ORIENT_TAU, sy.ORIENT_TAU, sy.ORIENT_LATITUDE, precision=prec,
                                      device=dev.index, lanes_per_filter=args.lanes_per_filter, stream="private",
                                      split_streams=args.split_streams, wide_arithmetic=wide)
        eng.set_process_noise(sy.orient_process_noise())
    else:
        # (tracking inputs: the engine runs on torch's current stream, the per-step generator kernel is a torch op ordered with it)
        eng = spe.BatchPoseUKF(per, precision=prec, device=dev.index, lanes_per_filter=args.lanes_per_filter,
                               stream=None if tracking else "private",
                               bucket_models=args.bucket_models, split_streams=args.split_streams, wide_arithmetic=wide)
    # input rings [N_RING][filters][..], contiguous (a multi-cycle launch addresses its slots inside them); *_d: the slots
    acc_ring = torch.empty((N_RING, per, 3), dtype=tdtype, device=dev)
    gyr_ring = torch.empty((N_RING, per, 3), dtype=tdtype, device=dev) if orient else None
    z_ring = torch.empty((N_RING, per, 3), dtype=tdtype, device=dev)
    Q_ring = torch.empty((N_RING, per, 9), dtype=tdtype, device=dev)
    acc_d = [acc_ring[k] for k in range(N_RING)]
    gyr_d = [gyr_ring[k] for k in range(N_RING)] if orient else None
    z_d = [z_ring[k] for k in range(N_RING)]
    Q_d = [Q_ring[k] for k in range(N_RING)]
    m_ring = torch.empty((N_RING, per), dtype=torch.int32, device=dev) if args.workload == "pose-mixed" else None
    m_d = [m_ring[k] for k in range(N_RING)] if m_ring is not None else None
    for lo in range(0, per, CH):
        hi = min(per, lo + CH)
        if orient:
            mu, cov = spe.synth.orient_initial(hi - lo, first=first + lo)
        else:
            mu, cov = spe.synth.pose_initial(hi - lo, first=first + lo)
        eng.initialize(mu, cov, first=lo)
        for k in range(N_RING):
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
            if tracking:
                z = z - mu[:, :3]      # the noise alone; the position it is added to is the filter's own, step by step
            z_d[k][lo:hi] = torch.from_numpy(z).to(dev, tdtype)
            Q_d[k][lo:hi] = torch.from_numpy(Q.reshape(-1, 9)).to(dev, tdtype)
    if not orient:
        eng.set_acceleration(None, 0.01 * np.eye(3))
    torch.cuda.synchronize()

    done = [0]   # fused cycles applied to the engine so far (the parity replay needs the exact count)

    cv = args.workload == "pose-cv"   # no acceleration latched (the engine's default is NaN): constant-velocity branch
    cpl = [max(1, args.cycles_per_launch)]   # cycles per launch of run_cycles (the extra multi-cycle region changes it)

    def step():
        r = done[0] % N_RING
        if orient:
            eng.bind_orient_inputs_dev(gyr_d[r], acc_d[r])
            eng.cycle_dev(DT, spe.MEAS_ORIENT_BODYVEL3, z_d[r], Q_d[r])
        else:
            if not cv:
                eng.bind_acceleration_dev(acc_d[r])
            if tracking:   # z = current mean position + this step's noise (z_ring holds the noise in this mode)
                torch.add(mu_view[:, :3], z_d[r], out=z_track)
                eng.cycle_dev(DT, spe.MEAS_POS3, z_track, Q_d[r])
            else:
                eng.cycle_dev(DT, spe.MEAS_POS3, z_d[r], Q_d[r], meas_model_dev=m_d[r] if m_d else None)
        done[0] += 1

    def run_cycles(k):
        """exactly k fused cycles: single launches, or launches of cpl cycles (the last one shorter); returns the launches"""
        if cpl[0] == 1:
            for _ in range(k):
                step()
            return k
        launches = 0
        while k > 0:
            c = min(cpl[0], k)
            if orient:
                eng.cycle_multi_dev(c, DT, spe.MEAS_ORIENT_BODYVEL3, z_ring, Q_ring, N_RING, done[0] % N_RING,
                                    in_a_dev=acc_ring, in_b_dev=gyr_ring)
            elif m_ring is not None:
                eng.cycle_multi_mixed_dev(c, DT, m_ring, z_ring, Q_ring, N_RING, done[0] % N_RING, in_a_dev=acc_ring)
            else:
                eng.cycle_multi_dev(c, DT, spe.MEAS_POS3, z_ring, Q_ring, N_RING, done[0] % N_RING,
                                    in_a_dev=None if cv else acc_ring)
            done[0] += c
            k -= c
            launches += 1
        return launches

    def kernel_region(k):
        """mean kernel time (ms) per cycle of k back-to-back cycles: HIP events on the engine's stream"""
        eng.timer_begin()
        run_cycles(k)
        return eng.timer_end() / k

    # Untimed pre-roll with the state put back afterwards.  (1) The FIRST burst of queued launches in a process is reported
    # complete ~50 ms late by the runtime in about one process out of three (GPU timestamps show the kernels back to back; later
    # bursts never, tools/sync_latency.py and DESIGN.md section 5).  (2) The clocks of an idle GPU need a few hundred
    # milliseconds of work to settle: a 20-step region started 30 ms after the first launch measured 1.25 ms per launch on a box
    # whose sustained figure was 1.15 ms.  So: bursts of 64 launches for --clock-warmup-seconds (default 0.4 s), then the
    # initial state is restored, and the W warm-up and K timed steps run over the same cycles of the (non-stationary) workload
    # as without the pre-roll.
    mu_ptr, cov_ptr, _ = eng.device_views()
    ts_ = "<f8" if prec == spe.F64 else "<f4"
    mu_view = torch.as_tensor(_DevArray(mu_ptr, (per, S), ts_), device=dev)
    cov_view = torch.as_tensor(_DevArray(cov_ptr, (per, eng.PK), ts_), device=dev)
    z_track = torch.empty((per, 3), dtype=tdtype, device=dev) if tracking else None
    if args.clock_warmup_seconds > 0 or args.clock_warmup_cycles > 0:
        init_state = (mu_view.clone(), cov_view.clone())
        torch.cuda.synchronize()
        t_pre = time.perf_counter()
        pre_done = 0
        while (pre_done < args.clock_warmup_cycles) if args.clock_warmup_cycles > 0 else (time.perf_counter() - t_pre < args.clock_warmup_seconds):
            burst = min(64, args.clock_warmup_cycles - pre_done) if args.clock_warmup_cycles > 0 else 64
            run_cycles(burst)
            pre_done += burst
            eng.sync()
        mu_view.copy_(init_state[0]); cov_view.copy_(init_state[1])
        torch.cuda.synchronize()
        del init_state
        done[0] = 0
    fence()
    run_cycles(args.warmup)
    fence()
    # snapshot of the warmed-up state (device to device) so that the extra kernel-time regions below start where
    # the timed region started: the headline workload is not stationary (the unobserved orientation covariance
    # grows with every cycle and moves the SO(3) maps onto their wide-angle paths)
    snap = None
    if not args.no_extra_regions:
        snap = (mu_view.clone(), cov_view.clone(), done[0])
        torch.cuda.synchronize()
    fence()                     # every rank starts its clock behind the same barrier, with nothing queued on its device
    eng.timer_begin()           # HIP events on the stream the kernel is launched on
    t0 = time.perf_counter()
    launches = run_cycles(args.steps)
    kernel_ms_total = eng.timer_end()
    if have_gpu:
        torch.cuda.synchronize()
    # every rank's clock stops when ITS K steps are complete on its device; the job's time is the MAX over the ranks (the
    # all-reduce below), all of which started behind the same barrier.  The closing barrier comes after the clock: its own
    # latency (an all-reduce over 8 GPUs) is not part of K steps -- at N = 8 a 20-step region of the 1 M-filter batch is 2.7 ms.
    elapsed_local = time.perf_counter() - t0
    fence()
    elapsed_barrier = time.perf_counter() - t0    # barrier to barrier: K steps + the closing barrier's own latency (rounds 1-2's definition)
    elapsed = elapsed_local
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rank_lo, rank_hi = rank_stats(elapsed_local / args.steps * 1e3)
    _, barrier_hi = rank_stats(elapsed_barrier / args.steps * 1e3)
    status_or = eng.status_summary()
    # which device every rank ran on (the driver's SCALE line verifies itself: N distinct devices, N RCCL ranks)
    rank_devices = None
    try:
        me = {"rank": rank, "device": (dev.index if have_gpu else None)}
        if have_gpu:
            pr = torch.cuda.get_device_properties(dev)
            me["name"] = pr.name
            me["uuid"] = str(getattr(pr, "uuid", "")) or None
            me["pci_bus_id"] = getattr(pr, "pci_bus_id", None)
        if dist is not None:
            allv = [None] * world
            dist.all_gather_object(allv, me)
            rank_devices = allv
        else:
            rank_devices = [me]
    except Exception as ex:   # (a diagnostic: never fail a measurement over it)
        rank_devices = [{"error": repr(ex)}]

    # ---- in-run parity on rank 0's first filters (before anything else advances the state)
    parity = parity_recent = None
    if rank == 0 and not args.no_parity:
        sample = min(args.parity_sample, per)
        parity = parity_check(args, spe, eng, first, sample, done[0], orient)
        m0, c0, _ = eng.state(0, sample)
        k0 = done[0]
        run_cycles(RECENT_CYCLES)
        eng.sync()
        parity_recent = parity_check(args, spe, eng, first, sample, done[0], orient, start=(m0, c0, k0))

    # ---- result gather over RCCL/xGMI (outside the timed region; means only)
    gather_ms = None
    if dist is not None:
        mu_local = mu_view
        if args.backend != "nccl":
            mu_local = mu_local.cpu()
        fence()
        g0 = time.perf_counter()
        gathered = spe.gather_means(mu_local, total, world, dist)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert gathered.shape == (total, S)
        del gathered

    # ---- kernel time under sustained clocks and in a burst (a 20-step timed region alone shows neither)
    kernel_ms = kernel_ms_total / args.steps          # per cycle (= per launch unless --cycles-per-launch > 1)
    kernel_ms_launch = kernel_ms_total / launches
    info = eng.last_launch_info()
    sustained_ms = burst_ms = None
    multi = None
    if not args.no_extra_regions:
        def restore():
            eng.sync()
            mu_view.copy_(snap[0]); cov_view.copy_(snap[1])
            torch.cuda.synchronize()
            done[0] = snap[2]
        sustained_ms = kernel_ms
        if args.steps < SUSTAINED_STEPS:
            restore()
            sustained_ms = kernel_region(SUSTAINED_STEPS)
        restore()
        time.sleep(1.0)    # let the clocks recover
        burst_ms = kernel_region(BURST_STEPS)
        fence()
        if cpl[0] == 1 and info["filters_per_workgroup"] == 4 and not tracking:
            # the same cycles once more as launches of MULTI_CYCLES cycles (ukfb_cycle_multi_dev): the filters stay in LDS
            # between the cycles of a launch.  Reported beside the headline, never as `value`.
            restore()
            cpl[0] = MULTI_CYCLES
            k_multi = (max(args.steps, 64) + MULTI_CYCLES - 1) // MULTI_CYCLES * MULTI_CYCLES
            run_cycles(MULTI_CYCLES)      # first launch of this kernel (code load), untimed
            restore()
            ms = kernel_region(k_multi)
            eng.sync()
            alg_cycle = (ALG_SCALARS_ORIENT if orient else ALG_SCALARS_POSE) * (8 if prec == spe.F64 else 4) * per
            multi = {"cycles_per_launch": MULTI_CYCLES, "cycles": k_multi, "kernel_ms_per_cycle": ms,
                     "filter_cycles_per_s_per_gpu": per / (ms * 1e-3), "kernel": eng.last_launch_info()["kernel"],
                     # NOT an HBM utilisation: SURVEY 8(d)'s algorithmic bytes of ONE cycle over the time of one cycle, the
                     # headline's yardstick applied per cycle so that the two rates compare.  The state crosses HBM once per
                     # LAUNCH here; what a launch really moves (packed state in and out once + 15 input scalars per cycle)
                     # over the launch time is hbm_frac_moved_layout
                     "algorithmic_bytes_per_cycle_over_hbm_peak": alg_cycle / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "hbm_frac_moved_layout": (per * (8 if prec == spe.F64 else 4) * (2 * (S + eng.PK) + MULTI_CYCLES * (18 if orient else 15)))
                     / (ms * MULTI_CYCLES * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "status_or": eng.status_summary(),
                     "note": "same start state and input ring as the timed region; kernel time (HIP events) of this rank"}
            if rank == 0 and not args.no_parity:
                # one more launch of the multi-cycle kernel against the oracle replay of its cycles, from the state downloaded
                # before it (the window of parity_recent: the fp32 engines drift from an fp64 replay over hundreds of cycles)
                sample_m = min(args.parity_sample, per)
                m0_m, c0_m, _ = eng.state(0, sample_m)
                k0_m = done[0]
                run_cycles(MULTI_CYCLES)
                eng.sync()
                multi["parity"] = parity_check(args, spe, eng, first, sample_m, done[0], orient, start=(m0_m, c0_m, k0_m))
            cpl[0] = 1
            fence()

    if rank == 0:
        tsize = 8 if prec == spe.F64 else 4
        value = total * args.steps / elapsed
        cycles_launch = args.steps / launches     # cycles of an average launch (1 unless --cycles-per-launch > 1)
        alg_bytes_launch = (ALG_SCALARS_ORIENT if orient else ALG_SCALARS_POSE) * tsize * per * cycles_launch
        achieved = alg_bytes_launch / (kernel_ms_launch * 1e-3) / 1e9
        lib_sha, build_head = running_lib_identity()
        # counters replayed from profiles/ count only if they were taken on THIS binary and launch shape (a multi-cycle entry
        # holds per-launch counters of its own cycles per launch; a run whose K is no multiple of C ends with one shorter
        # launch: the nominal shape decides)
        traffic_e, exact, traffic_src = load_profile_entry("traffic_latest.json", info["kernel"], per, cpl[0], lib_sha)
        traffic = None
        if traffic_e is not None:   # bytes per launch scale with the filters of the launch (per-filter streams only)
            traffic = float(traffic_e["hbm_bytes_per_launch"]) * (1.0 if exact else per / float(traffic_e["filters_per_launch"]))
        pmc_e, _, pmc_src = load_profile_entry("pmc_latest.json", info["kernel"], per, cpl[0], lib_sha)
        valu = None
        kernel_ms = kernel_ms_launch   # everything below is per launch
        if pmc_e is not None:
            # VALU-issue roofline: wave-instructions per launch x issue cycles per instruction, against what the
            # SIMDs can issue during the kernel's measured duration at the clock the PMC pass observed
            # (GRBM_GUI_ACTIVE / 8 / kernel time, MI355X_MICROARCH.md "DVFS give-back")
            waves = (per + info["filters_per_workgroup"] - 1) // info["filters_per_workgroup"]
            cyc = float(pmc_e["issue_cycles_per_valu_inst"])
            clock_hz = float(pmc_e["clock_mhz"]) * 1e6
            need = float(pmc_e["valu_insts_per_wave"]) * waves * cyc
            have = N_SIMD * clock_hz * kernel_ms * 1e-3
            valu = {"bound": "valu-issue", "valu_insts_per_wave": pmc_e["valu_insts_per_wave"],
                    "issue_cycles_per_inst": cyc, "clock_mhz": pmc_e["clock_mhz"], "simds": N_SIMD,
                    "achieved": need / (kernel_ms * 1e-3) / 1e12, "peak": N_SIMD * clock_hz / 1e12,
                    "unit": "T issue-cycles/s", "frac": need / have, "source": pmc_e.get("source")}
            if pmc_e.get("issue_cycles_per_wave_weighted"):
                # the same with the per-class issue costs measured on this chip (tools/valu_tput.hip): selects, compares,
                # DPP, 64-bit and SGPR-operand instructions take 4 cycles also in fp32, transcendentals 6.5 / 13
                wneed = float(pmc_e["issue_cycles_per_wave_weighted"]) * waves
                valu["issue_cycles_per_wave_weighted"] = pmc_e["issue_cycles_per_wave_weighted"]
                valu["frac_weighted"] = wneed / have
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
            "dtype": "f64" if wide else args.precision,
            "data": "synthetic",
            "config": {"workload": (f"{total} OrientationState UKF filters, fused predict(gyro+acc, dt=0.01)"
                                    f"+body-velocity update per step, {args.precision}, {per} filters per GPU" if orient else
                                    f"{total} PoseWithVelocity UKF filters, fused predict("
                                    + ("constant-velocity branch" if args.workload == "pose-cv" else "acc branch") + ", dt=0.01)+"
                                    + ("per-filter measurement model (9 models, 25 % inactive)" if args.workload == "pose-mixed"
                                       else "PositionMeasurement") + f" update per step, {args.precision}, {per} filters per GPU"
                                    + (", position fixes regenerated per step around the filter's own mean (--inputs tracking)" if tracking else "")),
                       "filters": total, "filters_per_gpu": per,
                       "hbm_format": args.precision, "arithmetic": "f64" if (wide or args.precision == "f64") else "f32",
                       "lanes_per_filter": 64 // max(1, info["filters_per_workgroup"]),
                       "parallelism": f"filter-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # --cycles-per-launch > 1: `achieved` / `frac` above are per-cycle algorithmic bytes x cycles of the
                         # launch (comparable with the headline), while the state crosses HBM once per launch -- the bytes a
                         # launch moves in the engine's layout over its duration:
                         "frac_moved_layout": (per * tsize * (2 * (S + eng.PK) + cycles_launch * (18 if orient else 15)))
                         / (kernel_ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         "valu_source": pmc_src,
                         "kernel": info["kernel"], "kernel_ms_per_launch": kernel_ms_launch,
                         "cycles_per_launch": cycles_launch,
                         "kernel_ms_per_launch_sustained": sustained_ms * cycles_launch if sustained_ms else None,
                         "kernel_ms_per_launch_burst": burst_ms * cycles_launch if burst_ms else None,
                         "frac_sustained": (alg_bytes_launch / (sustained_ms * cycles_launch * 1e-3) / 1e9 / HBM_PEAK_GBS) if sustained_ms else None,
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "lds_bytes_per_workgroup": info["lds_bytes"],
                         "valu": valu},
            "lib_sha16": lib_sha, "build_head": build_head,
            "status_or": status_or,
            "rccl_ranks": (dist.get_world_size() if (dist is not None and args.backend == "nccl") else None),
            "backend": (args.backend if dist is not None else None),
            "ms_per_step_rank_min": rank_lo, "ms_per_step_rank_max": rank_hi,
            "ms_per_step_barrier": barrier_hi,
            "rank_devices": rank_devices,
            "gather_ms": gather_ms,
            "parity": parity,
            "parity_recent": parity_recent,
            "multi_cycle": multi,
        }
        if not args.no_cpu_baseline and args.workload == "pose":
            # rank 0's host cores, also when N > 1 (the other ranks wait in the final barrier; nothing is being timed any more)
            out["cpu_baseline"] = cpu_baseline(args)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def run_group(args):
    """--launcher group: ONE process, N shards through ukfb_group_* (include/ukf_batch.h).  Same workload, same step and
    the same JSON line as the process-per-GPU path; `value` = all filters x K cycles / wall time between two group syncs."""
    import numpy as np
    import torch
    import slam_pose_estimation_amd as spe
    if args.workload != "pose":
        raise SystemExit("--launcher group times the headline workload (--workload pose)")
    if torch.cuda.device_count() == 0:
        raise SystemExit("bench.py: no HIP device visible -- the engine has no CPU path")
    devices = [int(d) for d in args.group_devices.split(",")] if args.group_devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit("--group-devices must name one device per shard (--gpus)")
    prec = spe.F64 if args.precision == "f64" else spe.F32
    tdtype = torch.float64 if prec == spe.F64 else torch.float32
    total = args.filters * args.gpus if args.scaling == "weak" else args.filters
    grp = spe.UKFGroup(spe.MODEL_POSE, prec, total, devices)
    grp.configure(split_streams=args.split_streams)
    CH = 131072
    rings = []     # per shard: (acc [N_RING][n][3], z, Q [N_RING][n][9]) on the shard's device
    for sh in grp.shards:
        dev = torch.device("cuda", sh["device"])
        n, first = sh["count"], sh["first"]
        acc_r = torch.empty((N_RING, n, 3), dtype=tdtype, device=dev)
        z_r = torch.empty((N_RING, n, 3), dtype=tdtype, device=dev)
        Q_r = torch.empty((N_RING, n, 9), dtype=tdtype, device=dev)
        for lo in range(0, n, CH):
            hi = min(n, lo + CH)
            mu, cov = spe.synth.pose_initial(hi - lo, first=first + lo)
            grp.initialize(mu, cov, first=first + lo)
            for k in range(N_RING):
                acc, z, Q = spe.synth.pose_cycle_inputs(hi - lo, k, mu[:, :3], first=first + lo)
                acc_r[k, lo:hi] = torch.from_numpy(acc).to(dev, tdtype)
                z_r[k, lo:hi] = torch.from_numpy(z).to(dev, tdtype)
                Q_r[k, lo:hi] = torch.from_numpy(Q.reshape(-1, 9)).to(dev, tdtype)
        rings.append((acc_r, z_r, Q_r))

    def reinitialise():
        for sh in grp.shards:
            for lo in range(0, sh["count"], CH):
                hi = min(sh["count"], lo + CH)
                mu, cov = spe.synth.pose_initial(hi - lo, first=sh["first"] + lo)
                grp.initialize(mu, cov, first=sh["first"] + lo)
    grp.set_acceleration(None, 0.01 * np.eye(3))
    for d in set(devices):
        torch.cuda.synchronize(d)
    done = [0]

    def run_cycles(k):
        for _ in range(k):
            r = done[0] % N_RING
            grp.bind_acceleration_dev([x[0][r] for x in rings])
            grp.cycle_dev(DT, spe.MEAS_POS3, [x[1][r] for x in rings], [x[2][r] for x in rings])
            done[0] += 1

    # clock pre-roll (the initial state is put back afterwards, as in the process-per-GPU path), warm-up, timed region
    if args.clock_warmup_seconds > 0:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.clock_warmup_seconds:
            run_cycles(64)
            grp.sync()
        reinitialise()
        done[0] = 0
        run_cycles(8)          # (the re-upload took seconds: a short burst before the warm-up steps)
        grp.sync()
        reinitialise()
        done[0] = 0
    run_cycles(args.warmup)
    grp.sync()
    grp.timer_begin()
    t0 = time.perf_counter()
    run_cycles(args.steps)
    kernel_ms_max, kernel_ms_shard = grp.timer_end()
    grp.sync()
    elapsed = time.perf_counter() - t0
    status_or = grp.status_summary()
    gather_ms = None
    if len(set(devices)) == len(devices):
        outs = [torch.empty((total, 13), dtype=tdtype, device=torch.device("cuda", d)) for d in devices]
        grp.gather_means(outs)      # first call: communicator + staging
        grp.sync()
        g0 = time.perf_counter()
        grp.gather_means(outs)
        grp.sync()
        gather_ms = (time.perf_counter() - g0) * 1e3
        m_all, _, _ = grp.state(0, min(total, 4096))
        assert np.array_equal(outs[-1][: m_all.shape[0]].double().cpu().numpy(), m_all)
    parity = None
    if not args.no_parity:
        eng0 = grp.shards[0]["engine"]
        parity = parity_check(args, spe, eng0, grp.shards[0]["first"], min(args.parity_sample, grp.shards[0]["count"]), done[0], False)
    tsize = 8 if prec == spe.F64 else 4
    per = max(s["count"] for s in grp.shards)
    alg_bytes_launch = ALG_SCALARS_POSE * tsize * per
    kernel_ms_launch = kernel_ms_max / args.steps
    info = grp.shards[0]["engine"].last_launch_info()
    out = {"metric": "UKF predict+update filter-cycles/s, PoseWithVelocity filters", "value": total * args.steps / elapsed,
           "unit": "filter-cycles/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
           "dtype": args.precision, "data": "synthetic", "launcher": "group",
           "config": {"workload": f"{total} PoseWithVelocity UKF filters, fused predict(acc branch, dt=0.01)+PositionMeasurement "
                                  f"update per step, {args.precision}, {per} filters per shard, one process / {args.gpus} shards "
                                  f"on devices {devices} (ukfb_group_*)",
                      "filters": total, "filters_per_gpu": per, "devices": devices, "parallelism": f"filter-sharded x{args.gpus}"},
           "roofline": {"bound": "hbm", "achieved": alg_bytes_launch / (kernel_ms_launch * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": alg_bytes_launch / (kernel_ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "kernel": info["kernel"], "kernel_ms_per_launch": kernel_ms_launch,
                        "kernel_ms_per_launch_by_shard": [x / args.steps for x in kernel_ms_shard],
                        "note": "per device: the slowest shard's HIP-event time per cycle and the bytes of one shard"},
           "status_or": status_or, "rccl_ranks": (len(devices) if gather_ms is not None else None), "gather_ms": gather_ms,
           "parity": parity, "cpu_baseline": None}
    print(json.dumps(out), flush=True)
    grp.close()
    return 0


def main():
    args = parse()
    if args.launcher == "group" and "RANK" not in os.environ and not args.plumbing_only:
        return run_group(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_children(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
