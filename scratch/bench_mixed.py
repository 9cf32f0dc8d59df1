import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
import simurlacra_amd as vs
KW = {"bob": dict(dt=0.01, max_steps=500), "qq-su": dict(dt=0.004, max_steps=4000), "qcp-su": dict(dt=0.002, max_steps=8000)}
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
per = (n_total // 3 // 256) * 256
members = [vs.VecSimEnv(nm, per, **KW[nm]) for nm in ("qq-su", "qcp-su", "bob")]
for m in members:
    m.set_params(np.tile(vs.nominal_params(m.name), (m.n_envs, 1)))
    m.set_auto_reset(True, seed=1)
    m.reset(seed=2)
mixed = vs.MixedVecSimEnv(members)
for rec in (True, False):
    chunk = 50
    ms = mixed.time_random(chunk, record=rec, iters=10)
    solo = sum(m.time_step_kernel(iters=10, k_steps=chunk, record=rec) for m in members)
    print(json.dumps(dict(workload=f"mixed qq+qcp+bob, {3*per} envs, {chunk} steps/launch, record={rec}", mixed_ms=ms, separate_ms_sum=solo,
                          env_steps_per_s=3 * per * chunk / (ms * 1e-3))))
