import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs
from simurlacra_amd import _lib as L
name = "pend"
kw = dict(dt=0.02, max_steps=25, init_state=np.array([0.1, 0.2]))
n = 1000
for rep in range(6):
    e = vs.VecSimEnv(name, n, **kw)
    e.set_rollout_variant("k_rollout_ws")
    e.set_auto_reset(True, seed=17)
    e.set_episode_log(True)
    e.reset(seed=1)
    e.set_traj_capacity(38)
    rec = rep % 2 == 0
    e.step_random(38, seed=4, record=rec)
    e.sync()
    cap = 1 << 16
    cnt = np.zeros(1, dtype=np.uint32)
    e._lib.vs_copy_to_host(e._h, L.VS_EP_COUNT, cnt.ctypes.data_as(C.c_void_p))
    ln = np.empty(cap, dtype=np.int32); ix = np.empty(cap, dtype=np.int32)
    e._lib.vs_copy_to_host(e._h, L.VS_EP_LENGTHS, ln.ctypes.data_as(C.c_void_p))
    e._lib.vs_copy_to_host(e._h, L.VS_EP_ENVIDX, ix.ctypes.data_as(C.c_void_p))
    nz = np.flatnonzero(ln)
    m = int(cnt[0])
    missing = sorted(set(range(n)) - set(ix[nz].tolist()))
    print(f"rep {rep} rec={rec}: count {m}, nonzero slots {len(nz)} (max index {nz.max() if len(nz) else -1}), zero slots below count: {np.flatnonzero(ln[:m] == 0)[:8]}..., "
          f"envs never logged: {len(missing)} e.g. {missing[:6]} .. {missing[-3:]}; es_count sum {e.get(L.VS_EPSTAT_COUNT).sum()}", flush=True)
    e.close()
