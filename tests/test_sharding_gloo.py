"""N > 1 path on CPU: two gloo ranks each own a filter range (shard_range), advance it with the CPU
oracle standing in for the GPU engine (tests may use the oracle), and gather the means exactly as
bench.py does over RCCL.  The gathered result must equal the unsharded computation bit for bit:
filters are independent, so there is no data-path collective to get wrong."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import slam_pose_estimation_amd as spe
    from oracle import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = spe.shard_range(total, world, rank)
    mu, cov = spe.synth.pose_initial(count, first=first)
    acc, z, Q = spe.synth.pose_cycle_inputs(count, 0, mu[:, :3], first=first)
    R = spe.synth.pose_default_process_noise()
    m, c, s1 = capi.pose_predict(mu, cov, R, acc, 0.01 * np.eye(3), 0.01)
    m, c, s2 = capi.pose_update(m, c, 0, z, Q)
    assert (s1 == 0).all() and (s2 == 0).all()
    full = spe.gather_means(torch.from_numpy(m), total, world, dist)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), full.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shard_and_gather_equals_single_rank(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import slam_pose_estimation_amd as spe
    from oracle import capi
    total, world = 101, 2            # ragged split: 51 + 50
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    mu, cov = spe.synth.pose_initial(total)
    acc, z, Q = spe.synth.pose_cycle_inputs(total, 0, mu[:, :3])
    R = spe.synth.pose_default_process_noise()
    m, c, _ = capi.pose_predict(mu, cov, R, acc, 0.01 * np.eye(3), 0.01)
    m, c, _ = capi.pose_update(m, c, 0, z, Q)
    assert got.shape == (total, 13)
    assert np.array_equal(got, m)
