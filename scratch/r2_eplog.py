import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs
from simurlacra_amd import _lib as L
name = sys.argv[1] if len(sys.argv) > 1 else "pend"
kw = dict(dt=0.02, max_steps=25, init_state=np.array([0.1, 0.2])) if name == "pend" else dict(dt=0.004, max_steps=25)
n = 1000
for rep in range(3):
    for per_env, mode in ((False, 1), (True, 2)):
        res = {}
        for variant in ("k_rollout", "k_rollout_ws", "k_rollout_ws64"):
            e = vs.VecSimEnv(name, n, **kw)
            e.set_record_mode(mode)
            if per_env:
                e.set_params(np.tile(vs.nominal_params(name), (n, 1)))
            e.set_rollout_variant(variant)
            e.set_auto_reset(True, seed=17)
            e.set_episode_log(True)
            e.reset(seed=1)
            e.set_traj_capacity(38)
            t = 0
            for k in (7, 1, 30):
                e.set_traj_offset(t)
                e.step_random(k, seed=4, record=True)
                t += k
            e.set_traj_offset(0)
            e.step_random(5, seed=9, record=False)
            r, l, ix = e.episodes()
            res[variant] = sorted(zip(ix.tolist(), l.tolist(), r.tolist()))
            z = sum(1 for x in res[variant] if x[1] == 0)
            print(rep, per_env, mode, variant, "episodes", len(r), "zero-length entries", z, "first", res[variant][:2], flush=True)
            e.close()
        print("   equal:", res["k_rollout"] == res["k_rollout_ws"], res["k_rollout"] == res["k_rollout_ws64"])
