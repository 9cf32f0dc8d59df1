"""sample_packed() of 65 536 QQube rollouts (64 x 64 network in the kernel, then DummyPolicy) -- run under
rocprofv3 --kernel-trace --stats to see the share of k_rollout_fnn / k_rollout*, k_pack_traj and the torch index kernels"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simurlacra_amd as vs
from simurlacra_amd.policies import DummyPolicy, FNNPolicy
from simurlacra_amd.sampling import ParallelRolloutSampler

env = vs.QQubeSwingUpSim(dt=0.004, max_steps=4000)
torch.manual_seed(0)
for pol in (FNNPolicy(env.spec, [64, 64], torch.tanh, featurize=False), DummyPolicy(env.spec)):
    s = ParallelRolloutSampler(env, pol, 8, min_rollouts=65536, seed=0)
    s.sample_packed()
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = s.sample_packed(); torch.cuda.synchronize()
        print(type(pol).__name__, sum(q.total_steps for q in p), "env steps in", round(time.perf_counter() - t0, 4), "s", flush=True)
