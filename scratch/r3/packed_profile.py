"""sample_packed() of 65 536 QQube rollouts with DummyPolicy: wall time per call; run under rocprofv3 --kernel-trace --stats for the
share of the rollout kernel, vs_rollout_lengths and k_pack_traj"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simurlacra_amd as vs
from simurlacra_amd.policies import DummyPolicy
from simurlacra_amd.sampling import ParallelRolloutSampler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = vs.QQubeSwingUpSim(dt=0.004, max_steps=4000)
s = ParallelRolloutSampler(env, DummyPolicy(env.spec), 8, min_rollouts=n, seed=0)
s.sample_packed()
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = s.sample_packed(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = sum(q.total_steps for q in p)
    print(f"DummyPolicy {n} rollouts: {tot} env steps in {dt * 1e3:.2f} ms = {tot / dt:.3e} env-steps/s; lengths mean {tot / n:.0f}", flush=True)
