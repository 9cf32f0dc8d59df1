import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
import simurlacra_amd as vs
from simurlacra_amd.policies import DummyPolicy, Policy
from simurlacra_amd.sampling import ParallelRolloutSampler

class MLP(Policy):
    def __init__(self, spec, hidden=64):
        super().__init__(spec)
        o, a = spec.obs_space.flat_dim, spec.act_space.flat_dim
        self.net = torch.nn.Sequential(torch.nn.Linear(o, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, a))
    def forward(self, obs):
        return self.net(obs)

for name, kw, n in (("qq-su", dict(dt=0.004, max_steps=4000), 4096), ("bob", dict(dt=0.01, max_steps=500), 4096), ("omo", dict(dt=0.02, max_steps=300), 16384)):
    env = vs.ENV_CLASSES[name](**kw)
    for pol_name, pol in (("DummyPolicy (fused)", DummyPolicy(env.spec)), ("MLP 64x64 (policy in the loop)", MLP(env.spec))):
        s = ParallelRolloutSampler(env, pol, 8, min_rollouts=n, seed=0)
        s.sample()  # warm-up (handle creation)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ros = s.sample()
        el = time.perf_counter() - t0
        steps = sum(len(r) for r in ros)
        print(json.dumps(dict(env=name, policy=pol_name, rollouts=len(ros), env_steps=steps, seconds=round(el, 3), env_steps_per_s=round(steps / el))))
