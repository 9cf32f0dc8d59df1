"""Live domain randomisation in k_rollout_ws: cycles per role and how many resets the stock served (diagnostic build
-DVS_WS_STAMP -> scratch/libvecsim_stamp.so).  usage: VS_LIB_PATH=scratch/libvecsim_stamp.so python scratch/r4_dr_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs  # noqa: E402
from bench import ENV_KW  # noqa: E402

CASES = [("qcp-su", 65536, "k_rollout_ws", 7), ("qcp-su", 65536, "k_rollout_ws", 0), ("qcp-su", 65536, "k_rollout_ws64", 7),
         ("qq-su", 65536, "k_rollout_ws", 7), ("qq-su", 65536, "k_rollout_ws", 0)]
for name, n, var, k in CASES:
    env = vs.VecSimEnv(name, n, **ENV_KW[name])
    env.set_params(np.tile(vs.nominal_params(name), (n, 1)))
    if k:
        env.set_randomizer(vs.create_default_randomizer(vs.ENV_CLASSES[name](**ENV_KW[name])).device_specs()[:k])
    env.set_auto_reset(True, seed=1)
    env.reset(seed=2)
    env.set_rollout_variant(var)
    env.set_traj_capacity(500)
    for _ in range(30):
        env.step_random(100, seed=3, record=True)
    env.sync()
    buf0 = np.zeros((env.ld // 64, 3, 4), dtype=np.uint64)
    env._check(env._lib.vs_copy_to_host(env._h, 99, buf0.ctypes.data_as(C.c_void_p)), "dbg")
    ms = env.time_step_kernel(iters=10, k_steps=100, record=True)
    buf = np.zeros((env.ld // 64, 3, 4), dtype=np.uint64)
    env._check(env._lib.vs_copy_to_host(env._h, 99, buf.ctypes.data_as(C.c_void_p)), "dbg")
    env.close()
    nb = float(buf[0, 0, 3])
    a0 = buf[:, 0, 0] - buf0[:, 0, 0]   # 11 launches
    st, dr = (a0 & np.uint64(0xFFFFFFFF)).astype(np.int64), (a0 >> np.uint64(32)).astype(np.int64)
    ne = 256 if "64" not in var else 64
    wg = (st + dr).reshape(-1, ne // 64).sum(axis=1)
    wgd = dr.reshape(-1, ne // 64).sum(axis=1)
    print(f"{name} n={n} {var} live-dr {k}: {ms * 1e3:.1f} us per 100 steps; resets over 11 launches: {st.sum()} from the stock, {dr.sum()} drawn on P; "
          f"per workgroup: mean {wg.mean():.1f} max {wg.max()} resets, drawn on P mean {wgd.mean():.2f} max {wgd.max()}")
    if not var.endswith("g"):
        cyc, cnt = (buf[:, 2, 0] - buf0[:, 2, 0]).sum(), (buf[:, 2, 1] - buf0[:, 2, 1]).sum()
        print(f"   reset branch on P: {cnt} passes, {cyc / max(cnt, 1):.0f} cycles each")
    for role in range(3 if var.endswith("g") else 2):
        cols = buf[:, role, 1 if role == 0 else 0:3].astype(np.float64) / nb
        m = cols.mean(axis=0)
        tot = cols[:, :-1].sum(axis=1)   # busy cycles per batch (without the barrier wait)
        print(f"   {'PCG'[role]}: " + ", ".join(f"{v:7.1f}" for v in m) + f" cyc/batch (sum {m.sum():.0f}); busy per wave: mean {tot.mean():.0f} max {tot.max():.0f}")
