"""
Multi-GPU layout of the path: the env batch shards embarrassingly.  One process per GPU, contiguous
env-index ranges per rank, no collective on the data path; the only exchange is the gather of completed-episode return
statistics (RCCL over xGMI when the tensors live on the GPU, gloo on CPU tensors in the tests).
"""
from typing import Tuple


def shard(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first global env index, number of envs) of `rank`: contiguous ranges [rank*N/W, (rank+1)*N/W), remainder to the
    first ranks.  Pass `first` to VecSimEnv.set_index_offset so that env i behaves the same for every world size."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(int(n_total), int(world))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def gather_episode_stats(count, retsum, lensum, group=None):
    """All-gather of per-rank completed-episode statistics.

    count / retsum / lensum: per-env accumulators of this rank (1-D torch tensors, any device; VS_EPSTAT_*).  They are
    reduced on the device first, so three doubles per rank go on the wire -- latency-bound, issued once per episode
    wave, never per step.  Returns dict(per_rank=[W,3] tensor (retsum, count, lensum), episodes, mean_return, mean_length)."""
    import torch
    import torch.distributed as dist

    mine = torch.stack([retsum.double().sum(), count.double().sum(), lensum.double().sum()])
    # (with a process group the collective runs whatever its size: a one-rank RCCL communicator is the cheapest rehearsal of
    # the library load / communicator set-up the 8-GPU run depends on, tests/test_gpu_rccl.py)
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "gloo":
            mine = mine.cpu()  # CPU rehearsal of the multi-rank path; RCCL takes the device tensor as it is
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, mine, group=group)
        per_rank = torch.stack(parts)
    else:
        per_rank = mine[None]
    tot = per_rank.sum(dim=0)
    n = max(float(tot[1]), 1.0)
    return dict(per_rank=per_rank.cpu(), episodes=int(tot[1]), mean_return=float(tot[0]) / n, mean_length=float(tot[2]) / n)


def gather_returns(returns, group=None):
    """All-gather of a fixed-size per-rank vector of episode returns (N/W fp32 per rank); every rank
    must pass the same length.  Returns the concatenation in rank order."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return returns.clone()
    out = torch.empty(dist.get_world_size(group) * returns.numel(), dtype=returns.dtype, device=returns.device)
    dist.all_gather_into_tensor(out, returns.contiguous(), group=group)
    return out
