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
