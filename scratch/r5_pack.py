"""vs_pack_traj alone: 65 536 QQube lanes x 1 000 recorded steps (record mode 2), full-length and ragged rollouts"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simurlacra_amd as vs
n, T = 65536, 1000
e = vs.VecSimEnv("qq-su", n, dt=0.004, max_steps=4000)
e.set_auto_reset(False); e.reset(seed=1); e.set_record_mode(2); e.set_traj_capacity(T)
e.step_random(T, seed=2, record=True); e.sync()
F = e.traj_layout()[0]
ar = torch.arange(n, device="cuda")
for tag, length in (("full", torch.full((n,), T, device="cuda", dtype=torch.int64)),
                    ("ragged (uniform 1..T)", 1 + (ar * 7919) % T)):
    start = torch.cumsum(length, 0) - length
    total = int(length.sum())
    for _ in range(2):
        pk = e.pack_traj(n, T, length, start, total=total)
    e.sync(); e.timer_start()
    for _ in range(5):
        pk = e.pack_traj(n, T, length, start, total=total)
    ms = e.timer_stop() / 5
    print(json.dumps(dict(lib=os.environ.get("VS_LIB_PATH", "shipped").split("/")[-1], lengths=tag, steps=total, ms=round(ms, 3),
                          alg_GBs=round(2 * 4 * F * total / (ms * 1e-3) / 1e9))), flush=True)
