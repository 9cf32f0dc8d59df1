"""host-side profile of one sample_packed() call (65 536 QQube rollouts, DummyPolicy)"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simurlacra_amd as vs
from simurlacra_amd.policies import DummyPolicy
from simurlacra_amd.sampling import ParallelRolloutSampler
env = vs.QQubeSwingUpSim(dt=0.004, max_steps=4000)
s = ParallelRolloutSampler(env, DummyPolicy(env.spec), 8, min_rollouts=65536, seed=0)
for _ in range(2):
    s.sample_packed()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
p = s.sample_packed(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
