"""what the in-kernel reset path costs per family: the shipped library against a diagnostic build whose resets restart
from a fixed state (-DVS_ABLATE_RESET -> scratch/libvecsim_noreset.so); run once per library via VS_LIB_PATH"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs
from simurlacra_amd import _lib as L
from bench import ENV_KW
for name, n in (("qbb", 32768), ("qbb", 65536), ("bob", 65536), ("qq-su", 65536), ("omo", 65536)):
    for var in ("k_rollout", "k_rollout_ws", "k_rollout_ws64"):
        env = vs.VecSimEnv(name, n, **ENV_KW[name])
        env.set_params(np.tile(vs.nominal_params(name), (n, 1)))
        env.set_auto_reset(True, seed=1); env.reset(seed=2); env.set_rollout_variant(var)
        env.set_traj_capacity(600)
        for _ in range(5):
            env.step_random(100, seed=3, record=True)
        env.sync()
        ms = env.time_step_kernel(iters=20, k_steps=100, record=True)
        c, r, l = env.episode_stats()
        print(f"{name} n={n} {var:15s} {ms*1e3:7.1f} us/100 steps   mean episode length {l.sum()/max(c.sum(),1):.1f}")
        env.close()
