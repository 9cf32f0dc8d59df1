"""where a sample() call with the network in the kernel spends its time (cProfile + coarse timers)"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
import simurlacra_amd as vs
from simurlacra_amd.sampling import ParallelRolloutSampler
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = vs.QQubeSwingUpSim(dt=0.004, max_steps=4000)
pol = vs.FNNPolicy(env.spec, [64, 64], torch.tanh, featurize=False)
s = ParallelRolloutSampler(env, pol, 8, min_rollouts=n, seed=0)
s.sample(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); ros = s.sample(); el = time.perf_counter() - t0
pr.disable()
print("rollouts", n, "seconds", el, "env-steps", sum(len(r) for r in ros), "env-steps/s", sum(len(r) for r in ros) / el)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
