"""Three-role kernel (P | C | G waves): bit-equality against k_rollout on a small batch, then timings per variant.
usage: python scratch/r3_g3.py [family ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs  # noqa: E402
from bench import ENV_KW  # noqa: E402

L = vs._lib
fams = sys.argv[1:] or ["qq-su", "omo", "bob", "qcp-su", "qbb", "pend"]
VARS = ("k_rollout", "k_rollout_ws64", "k_rollout_ws", "k_rollout_ws64g", "k_rollout_ws256g")
for name in fams:
    kw = dict(ENV_KW[name])
    kw["max_steps"] = 25
    for ar in (True, False):
        for mode in (1, 2):
            outs = []
            for var in ("k_rollout", "k_rollout_ws64g", "k_rollout_ws256g"):
                e = vs.VecSimEnv(name, 1000, **kw)
                e.set_record_mode(mode)
                e.set_params(np.tile(vs.nominal_params(name), (1000, 1)))
                e.set_rollout_variant(var)
                e.set_auto_reset(ar, seed=17)
                e.reset(seed=1)
                e.set_traj_capacity(38)
                t = 0
                for k in (7, 1, 30):
                    e.set_traj_offset(t)
                    e.step_random(k, seed=4, record=True)
                    t += k
                e.set_traj_offset(0)
                e.step_random(5, seed=9, record=False)
                tr = e.traj(38)
                fin = {w: e.get(w) for w in (L.VS_STATE, L.VS_OBS, L.VS_STEPCOUNT, L.VS_RETURNS, L.VS_REW, L.VS_DONE)}
                st = e.episode_stats()
                outs.append((tr, fin, st, e.rollout_variant()))
                e.close()
            for o in outs[1:]:
                bad = [k for k in outs[0][0] if not np.array_equal(outs[0][0][k], o[0][k])]
                bad += [w for w in outs[0][1] if not np.array_equal(outs[0][1][w], o[1][w])]
                bad += [j for j, (x, y) in enumerate(zip(outs[0][2], o[2])) if not np.array_equal(x, y)]
                print(f"{name} ar={ar} mode={mode} {o[3]}: {'OK' if not bad else 'MISMATCH ' + str(bad)}", flush=True)
for name in fams:
    for n in (65536, 4096):
        for rec in (1, 0):
            row = []
            for var in VARS:
                env = vs.VecSimEnv(name, n, **ENV_KW[name])
                env.set_params(np.tile(vs.nominal_params(name), (n, 1)))
                env.set_auto_reset(True, seed=1)
                env.reset(seed=2)
                env.set_rollout_variant(var)
                if env.rollout_variant() != var:
                    row.append("   n/a")
                    env.close()
                    continue
                if rec:
                    env.set_traj_capacity(500)
                for _ in range(3):
                    env.step_random(400, seed=3, record=bool(rec))
                env.sync()
                ms = min(env.time_step_kernel(iters=10, k_steps=400, record=bool(rec)) for _ in range(3))
                row.append(f"{ms * 1e3 / 4:6.1f}")
                env.close()
            print(f"{name} n={n} rec={rec} us/100 steps: " + "  ".join(f"{v}={r}" for v, r in zip(VARS, row)), flush=True)
