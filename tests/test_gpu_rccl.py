"""RCCL itself on the GPU box: a one-rank NCCL (= RCCL on ROCm) process group in a FRESH child process, the two collectives
of this path on device tensors.  The 8-GPU scaling run is the driver's to launch; this makes sure it is not the first time
librccl loads, a communicator initialises and an all-gather of device tensors completes.
One rank only: RCCL refuses two ranks on one GPU, and the box has one."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, os.environ["VS_ROOT"])
import torch
import torch.distributed as dist
assert torch.cuda.is_available()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
from simurlacra_amd.dist import gather_episode_stats, gather_returns, shard
import simurlacra_amd as vs
from simurlacra_amd import _lib as L
# real per-env accumulators of a handle, as bench.py hands them over: zero-copy device tensors of libvecsim's buffers
n = 4096
env = vs.VecSimEnv("bob", n, dt=0.01, max_steps=500)
first, count = shard(n, 0, 1)
env.set_index_offset(first)
env.set_auto_reset(True, seed=1)
env.reset(seed=2)
env.step_random(600, seed=3)   # every lane finishes at least one episode
env.sync()
cnt, rs, ls = (env.tensor(w)[0, :n] for w in (L.VS_EPSTAT_COUNT, L.VS_EPSTAT_RETSUM, L.VS_EPSTAT_LENSUM))
assert cnt.is_cuda and rs.is_cuda
ep = gather_episode_stats(cnt, rs, ls)
torch.cuda.synchronize()
ret = env.tensor(L.VS_RETURNS)[0, :n].clone()
allr = gather_returns(ret)
torch.cuda.synchronize()
el = torch.tensor([1.25], device="cuda:0", dtype=torch.float64)
parts = [torch.zeros_like(el)]
dist.all_gather(parts, el)   # the timing exchange of bench.py
dist.barrier()
out = dict(episodes=ep["episodes"], per_rank=ep["per_rank"].tolist(), want=[float(rs.double().sum()), float(cnt.double().sum()), float(ls.double().sum())],
           returns_equal=bool(torch.equal(allr, ret)), returns_device=str(allr.device), el=float(parts[0].item()),
           nccl_version=str(torch.cuda.nccl.version()))
env.close()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
@pytest.mark.timeout(420)
def test_rccl_one_rank_process_group_runs_the_collectives_of_this_path():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               VS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    res = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=400, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import json

    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["episodes"] >= 4096 and out["episodes"] == int(out["want"][1])
    assert out["per_rank"] == [out["want"]]  # one row: this rank's (return sum, count, length sum), through the all-gather
    assert out["returns_equal"] and out["returns_device"].startswith("cuda")
    assert out["el"] == 1.25
