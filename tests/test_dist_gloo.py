"""The N > 1 path on CPU: two processes, gloo backend.  Covers the sharding layout and the episode-return gather that
bench.py and the sampler use over RCCL on the GPUs (the env stepping itself needs no collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simurlacra_amd.dist import gather_episode_stats, gather_returns, shard

    n_total = 1000
    first, count = shard(n_total, rank, world)
    # synthetic per-env accumulators that are a function of the GLOBAL env index: the gathered statistics must not
    # depend on the number of ranks
    g = torch.arange(first, first + count, dtype=torch.float64)
    cnt = (g % 3).to(torch.int32)
    ret = (g * 0.5 * cnt).to(torch.float32)
    ln = (cnt * 7).to(torch.int32)
    ep = gather_episode_stats(cnt, ret, ln)
    allr = gather_returns(torch.full((4,), float(rank)))
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "stats.npy"), np.array([ep["episodes"], ep["mean_return"], ep["mean_length"]]))
        np.save(os.path.join(out_dir, "per_rank.npy"), ep["per_rank"].numpy())
        np.save(os.path.join(out_dir, "allr.npy"), allr.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gather_gloo(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    g = np.arange(1000, dtype=np.float64)
    cnt = g % 3
    ret = (g * 0.5 * cnt).astype(np.float32).astype(np.float64)
    stats = np.load(tmp_path / "stats.npy")
    assert stats[0] == cnt.sum()
    assert stats[1] == pytest.approx(ret.sum() / cnt.sum(), rel=1e-9)
    assert stats[2] == pytest.approx(7.0)
    per_rank = np.load(tmp_path / "per_rank.npy")
    assert per_rank.shape == (2, 3) and per_rank[:, 1].sum() == cnt.sum()
    assert np.array_equal(np.load(tmp_path / "allr.npy"), [0, 0, 0, 0, 1, 1, 1, 1])


def test_single_process_gather_is_identity():
    from simurlacra_amd.dist import gather_episode_stats, gather_returns

    ep = gather_episode_stats(torch.tensor([1, 2]), torch.tensor([3.0, 5.0]), torch.tensor([10, 20]))
    assert ep["episodes"] == 3 and ep["mean_return"] == pytest.approx(8 / 3) and ep["mean_length"] == 10
    assert torch.equal(gather_returns(torch.tensor([1.0, 2.0])), torch.tensor([1.0, 2.0]))
