"""Fused-kernel variants over batch sizes (us per 100 recorded steps; 400 steps per launch).
usage: python scratch/r3_sweep.py [family ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs  # noqa: E402
from bench import ENV_KW  # noqa: E402

fams = sys.argv[1:] or ["qq-su", "omo", "pend"]
VARS = ("k_rollout", "k_rollout_ws64", "k_rollout_ws", "k_rollout_ws64g", "k_rollout_ws256g")
DR = {"qq-su": [("mass_pend_pole", "normal", 0.024, 0.0048, 1e-3, np.inf), ("length_pend_pole", "normal", 0.129, 0.0258, 1e-3, np.inf)]}
for name in fams:
    for live in ((False, True) if name in DR else (False,)):
        for n in (1024, 4096, 8192, 16384, 32768, 49152, 65536, 81920, 98304):
            for rec in (1, 0):
                row = []
                for var in VARS:
                    env = vs.VecSimEnv(name, n, **ENV_KW[name])
                    env.set_params(np.tile(vs.nominal_params(name), (n, 1)))
                    if live:
                        env.set_randomizer(DR[name])
                    env.set_auto_reset(True, seed=1)
                    env.reset(seed=2)
                    env.set_rollout_variant(var)
                    if env.rollout_variant() != var:
                        row.append("   n/a")
                        env.close()
                        continue
                    if rec:
                        env.set_traj_capacity(400)
                    for _ in range(2):
                        env.step_random(400, seed=3, record=bool(rec))
                    env.sync()
                    ms = min(env.time_step_kernel(iters=5, k_steps=400, record=bool(rec)) for _ in range(3))
                    row.append(f"{ms * 1e3 / 4:6.1f}")
                    env.close()
                print(f"{name} live={int(live)} n={n:6d} rec={rec}: " + "  ".join(f"{v[9:] or 'plain'}={r}" for v, r in zip(VARS, row)), flush=True)
