"""Where the three waves of the three-role kernel spend their cycles (diagnostic build -DVS_WS_STAMP -> scratch/libvecsim_stamp.so).
usage: VS_LIB_PATH=scratch/libvecsim_stamp.so python scratch/r3_stamps.py [family ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs  # noqa: E402
from bench import ENV_KW  # noqa: E402

CASES = [("qq-su", 65536, "k_rollout_ws256g", 1), ("qq-su", 65536, "k_rollout_ws64g", 1), ("qq-su", 65536, "k_rollout_ws64", 1),
         ("qq-su", 65536, "k_rollout_ws256g", 0), ("qq-su", 4096, "k_rollout_ws64g", 1), ("omo", 65536, "k_rollout_ws256g", 1),
         ("bob", 65536, "k_rollout_ws256g", 1), ("pend", 65536, "k_rollout_ws256g", 1), ("qcp-su", 65536, "k_rollout_ws256g", 1),
         ("qbb", 32768, "k_rollout_ws64g", 1)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]]
LABELS = {0: ("read actions", "4 steps", "barrier"), 1: ("work_off", "draw+refill", "barrier"), 2: ("draw+refill", "obs_off", "barrier")}
for name, n, var, rec in CASES:
    env = vs.VecSimEnv(name, n, **ENV_KW[name])
    env.set_params(np.tile(vs.nominal_params(name), (n, 1)))
    env.set_auto_reset(True, seed=1)
    env.reset(seed=2)
    env.set_rollout_variant(var)
    if rec:
        env.set_traj_capacity(500)
    for _ in range(3):
        env.step_random(100, seed=3, record=bool(rec))
    env.sync()
    ms = env.time_step_kernel(iters=10, k_steps=100, record=bool(rec))
    buf = np.zeros((env.ld // 64, 3, 4), dtype=np.uint64)
    env._check(env._lib.vs_copy_to_host(env._h, 99, buf.ctypes.data_as(C.c_void_p)), "dbg")
    env.close()
    nb = float(buf[0, 0, 3])
    print(f"{name} n={n} {var} rec={rec}: {ms * 1e3:.1f} us per 100 steps")
    for role in range(3 if var.endswith("g") else 2):
        m = buf[:, role, :3].astype(np.float64).mean(axis=0) / nb
        print(f"   {'PCG'[role]}: " + ", ".join(f"{lab} {v:7.1f} cyc/batch" for lab, v in zip(LABELS[role], m)) + f"  (sum {m.sum():.0f} = {m.sum() / 4:.0f} per step)")
