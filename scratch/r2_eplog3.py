import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs
from simurlacra_amd import _lib as L
mode = sys.argv[1]
name = "pend"
kw = dict(dt=0.02, max_steps=25, init_state=np.array([0.1, 0.2]))
n = 1000
e = vs.VecSimEnv(name, n, **kw)
e.set_rollout_variant({"plain": "k_rollout", "ws64": "k_rollout_ws64"}.get(mode, "k_rollout_ws"))
e.set_auto_reset(True, seed=17)
e.set_episode_log(True)
e.reset(seed=1)
rec = mode != "norec"
if rec:
    e.set_traj_capacity(38)
if mode == "sync":
    e.sync()
if mode == "sleep":
    import time, torch
    e.sync(); torch.cuda.synchronize(); time.sleep(1.0)
if mode == "prefill":  # write a pattern into the log arrays from the host side first
    import torch
    from simurlacra_amd.vec_env import _DevArray
    for which in (L.VS_EP_LENGTHS, L.VS_EP_ENVIDX):
        t_ = torch.as_tensor(_DevArray(e._lib.vs_get(e._h, which), (1 << 16,), "<i4", e), device="cuda:0")
        t_.fill_(-7)
    torch.cuda.synchronize()
if mode == "warm":  # a first launch without finishing episodes
    e.step_random(3, seed=4, record=rec)
    e.sync()
    e.reset(seed=1)
e.step_random(38, seed=4, record=rec)
e.sync()
cap = 1 << 16
cnt = np.zeros(1, dtype=np.uint32)
e._lib.vs_copy_to_host(e._h, L.VS_EP_COUNT, cnt.ctypes.data_as(C.c_void_p))
ln = np.empty(cap, dtype=np.int32); ix = np.empty(cap, dtype=np.int32)
e._lib.vs_copy_to_host(e._h, L.VS_EP_LENGTHS, ln.ctypes.data_as(C.c_void_p))
e._lib.vs_copy_to_host(e._h, L.VS_EP_ENVIDX, ix.ctypes.data_as(C.c_void_p))
if mode == "prefill":
    print("   prefill: entries still -7 below count:", int((ln[:int(cnt[0])] == -7).sum()), "zeros:", int((ln[:int(cnt[0])] == 0).sum()))
nz = np.flatnonzero(ln)
missing = sorted(set(range(n)) - set(ix[nz].tolist()))
print(f"{mode}: count {int(cnt[0])}, nonzero slots {len(nz)}, never logged {len(missing)} {missing[:3]}..{missing[-3:]}; zero slots {np.flatnonzero(ln[:int(cnt[0])] == 0)[[0, -1]] if len(missing) else ''}", flush=True)
# second read of the same buffers (nothing ran in between), and a read through torch's own copy path
ln2 = np.empty(cap, dtype=np.int32)
e._lib.vs_copy_to_host(e._h, L.VS_EP_LENGTHS, ln2.ctypes.data_as(C.c_void_p))
import torch
ptr = e._lib.vs_get(e._h, L.VS_EP_LENGTHS)
from simurlacra_amd.vec_env import _DevArray
t = torch.as_tensor(_DevArray(ptr, (cap,), "<i4", e), device="cuda:0")
ln3 = t.cpu().numpy()
print(f"   second read nonzero {np.count_nonzero(ln2)}, torch read nonzero {np.count_nonzero(ln3)}, torch sum on device {int((t != 0).sum())}")

if os.environ.get("VS_LIB_PATH", "").endswith("logdbg.so"):
    dbg = np.zeros(e.ld, dtype=np.uint64)
    e._lib.vs_copy_to_host(e._h, 99, dbg.ctypes.data_as(C.c_void_p))
    base, slot = (dbg >> np.uint64(32)).astype(np.int64)[:n], (dbg & np.uint64(0xFFFFFFFF)).astype(np.int64)[:n]
    bad = [w for w in range(0, n, 64) if len(set(slot[w:w + 64].tolist())) != len(slot[w:w + 64])]
    print("   kernel-side: distinct slots", len(set(slot.tolist())), "waves with colliding slots:", bad[:5],
          "example wave", (bad[0], base[bad[0]:bad[0] + 4].tolist(), slot[bad[0]:bad[0] + 4].tolist(), slot[bad[0] + 60:bad[0] + 64].tolist()) if bad else None)
