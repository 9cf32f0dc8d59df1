"""Which workgroups make a launch under live randomisation late?  (-DVS_WS_STAMP build; cartpole, 65 536 envs, 64-env workgroups,
7 live-randomised parameters, launches of 400 steps.)  Per wave: resets served from the stock / drawn on the physics wave during ONE
launch, cycles inside its reset branch, busy cycles of its physics wave -- the slowest waves listed with their event counts."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs  # noqa: E402
from bench import ENV_KW  # noqa: E402

name, n, chunk = "qcp-su", 65536, 400
for k in (7, 0):
    env = vs.VecSimEnv(name, n, **ENV_KW[name])
    env.set_params(np.tile(vs.nominal_params(name), (n, 1)))
    if k:
        env.set_randomizer(vs.create_default_randomizer(vs.ENV_CLASSES[name](**ENV_KW[name])).device_specs()[:k])
    env.set_auto_reset(True, seed=1)
    env.reset(seed=2)
    env.set_rollout_variant("k_rollout_ws64")
    env.set_traj_capacity(chunk)
    for _ in range(60):   # 24 000 steps: three generations of episodes
        env.step_random(chunk, seed=3, record=True)
    env.sync()

    def snap():
        buf = np.zeros((env.ld // 64, 3, 4), dtype=np.uint64)
        env._check(env._lib.vs_copy_to_host(env._h, 99, buf.ctypes.data_as(C.c_void_p)), "dbg")
        return buf

    rows = []
    for launch in range(6):
        b0 = snap()
        env.timer_start()
        env.step_random(chunk, seed=3, record=True)
        ms = env.timer_stop()
        b1 = snap()
        a = b1[:, 0, 0] - b0[:, 0, 0]
        st, dr = (a & np.uint64(0xFFFFFFFF)).astype(np.int64), (a >> np.uint64(32)).astype(np.int64)
        rc = (b1[:, 2, 0] - b0[:, 2, 0]).astype(np.int64)      # cycles inside the reset branch of the P wave (this launch)
        rp = (b1[:, 2, 1] - b0[:, 2, 1]).astype(np.int64)      # passes through it
        busy = b1[:, 0, 1].astype(np.int64)                     # P busy cycles of the launch (overwritten per launch)
        wait = b1[:, 0, 2].astype(np.int64)
        order = np.argsort(-busy)[:6]
        print(f"live-dr {k} launch {launch}: {ms * 1e3:.1f} us; P busy cycles mean {busy.mean():.0f} max {busy.max()} (+{(busy.max() / busy.mean() - 1) * 100:.1f} %); "
              f"resets per wave mean {(st + dr).mean():.2f} max {(st + dr).max()}, drawn on P mean {dr.mean():.2f} max {dr.max()}; reset-branch cycles per wave mean {rc.mean():.0f} max {rc.max()}")
        for w in order:
            print(f"      wave {w:5d}: busy {busy[w]} (+{busy[w] - int(busy.mean())}), reset branch {rc[w]} cycles in {rp[w]} passes, {st[w]} from stock, {dr[w]} drawn on P, barrier wait {wait[w]}")
    env.close()
