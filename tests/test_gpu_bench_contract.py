"""bench.py's one-line JSON contract (driver side): keys, types and the two extra objects, on a small workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = run_bench("--filters", "16384", "--steps", "5", "--warmup", "2", "--cpu-sample-filters", "256",
                  "--cpu-sample-seconds", "0.5")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert key in d and isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    # PMC figures are replayed from profiles/ only if they were measured on the library that is running (hash in the line)
    ts, vs = r["traffic_source"], r["valu_source"]
    assert ts["file"] == "profiles/traffic_latest.json" and len(d["lib_sha16"]) == 16 and ts["running_lib_sha16"] == d["lib_sha16"]
    assert (r["traffic"] is not None) == ts["matches_running_lib"] and (r["valu"] is not None) == vs["matches_running_lib"]
    if r["traffic"] is not None:
        assert r["traffic"] > 0                                      # scaled to this launch size
    assert r["kernel_ms_per_launch"] > 0 and r["kernel_ms_per_launch_sustained"] > 0 and r["kernel_ms_per_launch_burst"] > 0
    v = r["valu"]
    if v is not None:
        assert v["bound"] == "valu-issue" and 0 < v["frac"] and v["valu_insts_per_wave"] > 0 and v["clock_mhz"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert c["single_core"]["cores"] == 1 and c["single_core"]["value"] > 0 and "march=native" in c["build"]
    assert d["value"] > 0 and abs(d["value"] - 16384 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-6
    assert d["status_or"] == 0 and d["scaling"] == "strong" and d["rccl_ranks"] is None
    assert d["config"]["filters"] == 16384 and d["config"]["filters_per_gpu"] == 16384
    for key in ("parity", "parity_recent"):      # in-run parity against the oracle, fp64 tolerance of north_star
        p = d[key]
        assert p["tol"] == 1e-9 and p["ok"] is True and p["max_abs_mu"] <= 1e-9 and p["max_abs_cov"] <= 1e-9, p
    m = d["multi_cycle"]                          # extra region: the same cycles as launches of 8 (never the headline value)
    assert m["cycles_per_launch"] == 8 and m["cycles"] % 8 == 0 and m["kernel_ms_per_cycle"] > 0 and m["status_or"] == 0
    assert "multicycle" in m["kernel"] and r["cycles_per_launch"] == 1 and "multicycle" not in r["kernel"]
    assert m["parity"]["ok"] is True and m["parity"]["max_abs_cov"] <= 1e-9 and m["algorithmic_bytes_per_cycle_over_hbm_peak"] > m["hbm_frac_moved_layout"] > 0


def test_bench_other_workloads_run():
    d = run_bench("--workload", "orient", "--precision", "f32", "--filters", "8192", "--steps", "3", "--warmup", "1",
                  "--no-cpu-baseline")
    assert d["dtype"] == "f32" and "OrientationState" in d["metric"] and d["status_or"] == 0
    assert d["parity_recent"]["ok"] is True and d["parity_recent"]["tol"] == 1e-4
    d = run_bench("--workload", "pose-mixed", "--filters", "8192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert d["value"] > 0 and d["parity"]["ok"] is True and d["multi_cycle"]["status_or"] == d["status_or"]
    # SURVEY 8(d)'s secondary run: the constant-velocity branch (no acceleration latched)
    d = run_bench("--workload", "pose-cv", "--filters", "8192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert "constant-velocity" in d["config"]["workload"] and d["status_or"] == 0 and d["parity"]["ok"] is True
    # position fixes regenerated on the device, step by step, around the filter's own mean (inputs regenerated
    # per cycle, nothing crosses PCIe inside the timed region); the oracle replay regenerates them the same way
    for prec, tol in (("f64", 1e-9), ("f32", 1e-4)):
        d = run_bench("--inputs", "tracking", "--precision", prec, "--filters", "8192", "--steps", "12", "--warmup", "3", "--no-cpu-baseline")
        assert "tracking" in d["config"]["workload"] and d["status_or"] == 0 and d["multi_cycle"] is None
        assert d["parity"]["ok"] is True and d["parity"]["tol"] == tol and d["parity_recent"]["ok"] is True


def test_bench_cycles_per_launch():
    """K timed cycles as launches of C cycles (ukfb_cycle_multi_dev): exactly K cycles, parity against the oracle replay"""
    d = run_bench("--filters", "8192", "--steps", "13", "--warmup", "3", "--cycles-per-launch", "4", "--no-cpu-baseline")
    r = d["roofline"]
    assert d["steps"] == 13 and "multicycle" in r["kernel"] and abs(r["cycles_per_launch"] - 13 / 4) < 1e-9
    assert d["status_or"] == 0 and d["parity"]["ok"] is True and d["parity_recent"]["ok"] is True
    assert "after 16 fused cycles" in d["parity"]["sample"]
    assert abs(d["value"] - 8192 * 13 / (d["ms_per_step"] * 13e-3)) / d["value"] < 1e-6


def test_single_rank_rccl_rehearsal():
    """The N > 1 code path (RCCL process group, barrier, all-reduce MAX, gather) with one rank on the one GPU."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = run_bench("--gpus", "1", "--filters", "16384", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra-regions",
                  env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "UKFB_BENCH_FORCE_DIST": "1"})
    assert d["rccl_ranks"] == 1 and d["backend"] == "nccl" and d["gather_ms"] is not None and d["status_or"] == 0
    assert d["ms_per_step_rank_min"] <= d["ms_per_step_rank_max"]


def test_two_self_launched_ranks_share_the_gpu_over_gloo():
    """`python bench.py --gpus 2` with no wrapper: two child ranks, strong scaling (8 192 filters each), engine on
    the same device, rank plumbing over gloo (a one-GPU box cannot form a 2-rank RCCL communicator)."""
    d = run_bench("--gpus", "2", "--backend", "gloo", "--filters", "16384", "--steps", "3", "--warmup", "1",
                  "--no-cpu-baseline", "--no-extra-regions")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["filters_per_gpu"] == 8192
    assert d["rccl_ranks"] is None and d["backend"] == "gloo" and d["gather_ms"] is not None and d["status_or"] == 0
    assert d["value"] > 0 and d["parity"]["ok"] is True


def test_group_launcher_one_process_two_shards():
    """`--launcher group`: ONE process drives the shards through ukfb_group_* (the C++ host's multi-GPU shape).  On a 1-GPU
    box: two shards on device 0 (gather skipped), and a 1-device group whose gather runs through RCCL."""
    for extra, ranks in (["--gpus", "2", "--group-devices", "0,0"], None), (["--gpus", "1"], 1):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher", "group", "--filters", "32768", "--steps", "8",
                              "--warmup", "2", "--clock-warmup-seconds", "0.05", "--no-cpu-baseline"] + extra,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads(out.stdout.strip().splitlines()[-1])
        assert d["launcher"] == "group" and d["value"] > 0 and d["status_or"] == 0 and d["config"]["filters"] == 32768
        assert d["parity"]["ok"] is True and d["rccl_ranks"] == ranks
        assert (d["gather_ms"] is not None) == (ranks is not None)
