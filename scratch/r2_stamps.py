"""Where the two waves of k_rollout_ws spend their cycles (diagnostic build -DVS_WS_STAMP -> scratch/libvecsim_stamp.so).
usage: VS_LIB_PATH=scratch/libvecsim_stamp.so python scratch/r2_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs  # noqa: E402
from bench import ENV_KW  # noqa: E402

CASES = (("qq-su", 65536, "k_rollout_ws", 1), ("qq-su", 65536, "k_rollout_ws", 0), ("qq-su", 65536, "k_rollout_ws64", 1),
         ("qq-su", 4096, "k_rollout_ws64", 1), ("qq-su", 4096, "k_rollout_ws64", 0), ("qcp-su", 65536, "k_rollout_ws", 1),
         ("qbb", 32768, "k_rollout_ws64", 1), ("qbb", 32768, "k_rollout_ws64", 0), ("bob", 65536, "k_rollout_ws", 1),
         ("omo", 65536, "k_rollout_ws64", 1))
if len(sys.argv) > 1:
    CASES = tuple(c for c in CASES if c[0] in sys.argv[1:])
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
    buf = np.zeros((env.ld // 64, 2, 4), dtype=np.uint64)
    env._check(env._lib.vs_copy_to_host(env._h, 99, buf.ctypes.data_as(C.c_void_p)), "dbg")
    env.close()
    nb = float(buf[0, 0, 3])
    tot = buf[:, :, :3].sum(axis=2).mean(axis=0)  # cycles per role in the loop
    print(f"{name} n={n} {var} rec={rec}: {ms * 1e3:.1f} us per 100 steps; loop cycles P {tot[0]:.0f} C {tot[1]:.0f} (=> {tot[0] / (ms * 1e6):.2f} cycles/ns)")
    for role, labels in ((0, ("read actions", "4 steps", "barrier")), (1, ("work_off", "draw", "barrier"))):
        m = buf[:, role, :3].astype(np.float64).mean(axis=0) / nb
        print(f"   {'PC'[role]}: " + ", ".join(f"{lab} {v:7.1f} cyc/batch" for lab, v in zip(labels, m)) + f"  (sum {m.sum():.0f} = {m.sum() / 4:.0f} per step)")
