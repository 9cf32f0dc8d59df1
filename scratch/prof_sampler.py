import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
import simurlacra_amd as vs
from simurlacra_amd.policies import DummyPolicy
from simurlacra_amd.sampling import ParallelRolloutSampler
env = vs.QQubeSwingUpSim(dt=0.004, max_steps=4000)
s = ParallelRolloutSampler(env, DummyPolicy(env.spec), 8, min_rollouts=4096, seed=0)
s.sample(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); ros = s.sample(); el = time.perf_counter() - t0
pr.disable()
print("seconds", el, "env-steps/s", sum(len(r) for r in ros) / el)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
