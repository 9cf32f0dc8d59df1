import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs
from bench import ENV_KW
for n in (32768,):
    for var in ("k_rollout", "k_rollout_ws64"):
        for rec in (0, 1):
            env = vs.VecSimEnv("qbb", n, **ENV_KW["qbb"])
            env.set_params(np.tile(vs.nominal_params("qbb"), (n, 1)))
            env.set_auto_reset(True, seed=1); env.reset(seed=2); env.set_rollout_variant(var)
            if rec: env.set_traj_capacity(600)
            env.step_random(100, seed=3, record=bool(rec)); env.sync()
            ms = env.time_step_kernel(iters=20, k_steps=100, record=bool(rec))
            print(f"qbb n={n} {var:15s} rec={rec} {ms*1e3:7.1f} us/100 steps")
            env.close()
