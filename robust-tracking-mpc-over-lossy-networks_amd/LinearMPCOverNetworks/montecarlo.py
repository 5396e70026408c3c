"""Sharding of a Monte-Carlo sweep over ranks (one process per GPU) and the one collective of
the path: gathering per-trajectory statistics.

The reference runs `for i in p_loss: for l_mc in N_MC: for t in T:` in one Python process
(results_linear_system.py:165-209); trajectories never interact (estimator/actuator are
re-created per run, :186-188), so they are sharded with no data-path collective.  What is
exchanged is what the script aggregates afterwards (:291 tracking error, :268-270 failure
counts, :305-315 timing/iteration statistics): a few numbers per trajectory, once per sweep.
"""
from __future__ import annotations

import numpy as np


def trajectory_table(p_loss, n_mc: int):
    """Global trajectory index -> (p_loss index, seed index), p_loss-minor so that every
    contiguous shard sees all loss rates (balanced iteration counts)."""
    p_loss = np.asarray(p_loss, dtype=np.float64)
    n = len(p_loss) * int(n_mc)
    g = np.arange(n)
    return g % len(p_loss), g // len(p_loss)


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_statistics(local, n_total: int, rank: int, world: int, group=None):
    """All-gather of a (n_local, k) tensor of per-trajectory statistics into the global
    (n_total, k) table, identical on every rank.  Backend-agnostic: `nccl` (= RCCL over xGMI) on
    the GPUs, `gloo` in the CPU tests.  Shards of unequal size are padded to the largest."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, expected {sizes[rank]}")
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)
