import sys; sys.path.insert(0, '.')
import numpy as np, torch
import simurlacra_amd as vs
from simurlacra_amd.policies import DummyPolicy
from simurlacra_amd.sampling import ParallelRolloutSampler
for wrap in (0, 1, 2):
    env = vs.OneMassOscillatorSim(dt=0.02, max_steps=300)
    if wrap == 1:
        env = vs.ActDelayWrapper(env, delay=1)
    if wrap == 2:
        env = vs.GaussianObsNoiseWrapper(vs.GaussianActNoiseWrapper(env, noise_std=np.array([1.0])), noise_std=np.array([0.1, 0.1]))
    for rep in range(3):
        s2 = ParallelRolloutSampler(env, DummyPolicy(env.spec), 2, min_steps=2000, seed=1, batch_lanes=64)
        s3 = ParallelRolloutSampler(env, DummyPolicy(env.spec), 5, min_steps=2000, seed=1, batch_lanes=16)
        r2, r3 = s2.sample(), s3.sample()
        same = len(r2) == len(r3) and all(np.array_equal(a.rewards, b.rewards) and np.array_equal(a.observations, b.observations) for a, b in zip(r2, r3))
        print(wrap, rep, len(r2), len(r3), same, [len(r) for r in r2][:8], [len(r) for r in r3][:8])
