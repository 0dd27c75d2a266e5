"""Filter-range sharding across ranks (one process per GPU).  Filters are independent
(UnscentedKalmanFilter.hpp:150: every filter owns its own ukf object), so the batch splits into
contiguous ranges with no data-path collective; the only exchange is a final gather of results."""
from __future__ import annotations


def shard_range(total: int, world: int, rank: int):
    """Contiguous range [first, first + count) of `total` filters owned by `rank` of `world`.
    The first total % world ranks own one extra filter."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_means(local_mu, total: int, world: int, dist=None):
    """all_gather of the per-rank mean arrays (torch tensors [count, S]) into [total, S] on every rank.
    Ranges may be ragged (shard_range), so shards are padded to the largest count for the collective."""
    import torch
    if dist is None or world == 1:
        return local_mu
    counts = [shard_range(total, world, r)[1] for r in range(world)]
    mx = max(counts)
    pad = torch.zeros((mx, local_mu.shape[1]), dtype=local_mu.dtype, device=local_mu.device)
    pad[: local_mu.shape[0]] = local_mu
    out = torch.empty((world * mx, local_mu.shape[1]), dtype=local_mu.dtype, device=local_mu.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)
