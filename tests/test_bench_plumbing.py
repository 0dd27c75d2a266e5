"""bench.py's N > 1 launch path on a CPU box: the plain command `python bench.py --gpus 2 ...` must start its own
ranks (no torch.distributed.run wrapper), rendezvous on 127.0.0.1 over gloo, shard the filters (strong scaling by
default), run barrier / all-reduce / ragged gather and print ONE JSON line from rank 0.  `--plumbing-only` rehearses
exactly that and nothing else: there is no engine without a GPU and no number is reported."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=e)
    return out


@pytest.mark.timeout(400)
def test_plain_entry_starts_its_own_ranks_gloo():
    out = _run("--gpus", "2", "--backend", "gloo", "--plumbing-only", "--filters", "101")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["plumbing_only"] is True and d["value"] is None and d["backend"] == "gloo"
    assert d["scaling"] == "strong" and d["config"]["filters"] == 101 and d["config"]["filters_per_gpu"] == 51   # ragged


@pytest.mark.timeout(400)
def test_weak_scaling_option_and_three_ranks():
    out = _run("--gpus", "3", "--backend", "gloo", "--plumbing-only", "--filters", "7", "--scaling", "weak")
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 3 and d["scaling"] == "weak" and d["config"]["filters"] == 21 and d["config"]["filters_per_gpu"] == 7


def test_without_a_gpu_the_real_bench_refuses_to_run():
    """No CPU fallback: without --plumbing-only and without a HIP device bench.py must fail loudly."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible here")
    out = _run("--steps", "1", "--warmup", "0", "--filters", "64", "--no-cpu-baseline")
    assert out.returncode != 0 and "no HIP device" in (out.stderr + out.stdout)


def test_rank_mismatch_is_an_error():
    out = _run("--gpus", "4", "--plumbing-only", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_a_failing_rank_fails_the_launcher_and_does_not_hang():
    """A rank that dies (here: an unknown backend) must take the launcher down with a non-zero exit code, not leave the
    other ranks waiting in a rendezvous."""
    out = _run("--gpus", "2", "--backend", "no-such-backend", "--plumbing-only", "--filters", "8")
    assert out.returncode != 0 and not [l for l in out.stdout.strip().splitlines() if l.startswith("{")]


@pytest.mark.timeout(400)
def test_under_torch_distributed_run_every_process_is_a_rank():
    """The driver's own N > 1 command line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`.  RANK is set, so bench.py must NOT start children again; rank 0 prints
    the one JSON line, the other ranks print nothing."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1", "--backend", "gloo", "--plumbing-only", "--filters", "64"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["plumbing_only"] is True and d["backend"] == "gloo" and d["config"]["filters_per_gpu"] == 32


def test_replayed_counters_are_tied_to_the_binary():
    """bench.py replays PMC figures from profiles/ only for the kernel, launch size, launch shape AND library hash they were
    measured on; anything else reports null with the reason in `*_source`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    doc = {"entries": [
        {"kernel": "ukf_kernel16<f64,pose,cycle>", "filters_per_launch": 1048576, "cycles_per_launch": 1.0, "lib_sha16": "aaaa",
         "build_head": "h1", "hbm_bytes_per_launch": 1.8e9},
        {"kernel": "ukf_kernel16<f64,pose,cycle>", "filters_per_launch": 65536, "cycles_per_launch": 1.0, "lib_sha16": "aaaa",
         "hbm_bytes_per_launch": 1.1e8},
        {"kernel": "ukf_kernel16<f64,pose,multicycle>", "filters_per_launch": 1048576, "cycles_per_launch": 8.0, "lib_sha16": "aaaa",
         "hbm_bytes_per_launch": 2.4e9}]}
    e, exact, src = bench.select_profile_entry(doc, "ukf_kernel16<f64,pose,cycle>", 65536, 1.0, "aaaa")
    assert e["hbm_bytes_per_launch"] == 1.1e8 and exact and src["matches_running_lib"] and src["lib_sha16"] == "aaaa"
    e, exact, src = bench.select_profile_entry(doc, "ukf_kernel16<f64,pose,cycle>", 131072, 1.0, "aaaa")
    assert e is not None and not exact and src["matches_running_lib"]            # another launch size: scaled by the caller
    e, exact, src = bench.select_profile_entry(doc, "ukf_kernel16<f64,pose,cycle>", 1048576, 1.0, "bbbb")
    assert e is None and not src["matches_running_lib"] and src["lib_sha16"] == "aaaa" and src["running_lib_sha16"] == "bbbb"
    e, _, src = bench.select_profile_entry(doc, "ukf_kernel16<f64,pose,multicycle>", 1048576, 1.0, "aaaa")
    assert e is None and src["lib_sha16"] is None                                # other launch shape: no entry at all
    e, _, src = bench.select_profile_entry(None, "x", 1, 1.0, "aaaa")
    assert e is None and not src["matches_running_lib"]
    # entries written before the hash existed never match
    e, _, src = bench.select_profile_entry({"entries": [{"kernel": "k", "filters_per_launch": 4}]}, "k", 4, 1.0, "aaaa")
    assert e is None and src["lib_sha16"] is None
    # the committed files are readable by the lookup (whatever binary they belong to)
    for name in ("traffic_latest.json", "pmc_latest.json"):
        _, _, src = bench.load_profile_entry(name, "ukf_kernel16<f64,pose,cycle>", 1048576, 1.0, "0000")
        assert src["file"] == "profiles/" + name and src["matches_running_lib"] is False
