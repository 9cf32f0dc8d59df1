"""achieved one-step errors of the HIP path against the reference's golden step cases, per family (to set the tolerances)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simurlacra_amd as vs
from simurlacra_amd import _lib as L
from oracle import cpu_ref
from bench import ENV_KW as KW

for name in KW:
    g = np.load(os.path.join(ROOT, "tests", "golden", f"step_{name.replace('-', '_')}.npz"))
    n = g["state"].shape[0]
    env = vs.VecSimEnv(name, n, **KW[name])
    ref = cpu_ref.make_ref(name, **KW[name])
    env.set_params(g["params"].astype(np.float32))
    env.reset(init_state=g["state"].astype(np.float32))
    if g["hidden"].shape[1]:
        env.put(L.VS_HIDDEN, g["hidden"].astype(np.float32))
    env.put(L.VS_STEPCOUNT, g["curr_step"].astype(np.int32))
    env.step(torch.from_numpy(g["act"].astype(np.float32)).cuda())
    shi = ref.bounds(g["params"])[1]
    st = env.get(L.VS_STATE).astype(np.float64)
    es = np.abs(st - g["nstate"])
    # inputs are rounded to fp32 first: that alone moves the result by ~6e-8 relative of the inputs' magnitude
    print(f"{name:7s} state: max |err|/|exp| (|exp| > 1e-2 bound) {np.where(np.abs(g['nstate']) > 1e-2 * shi, es / np.maximum(np.abs(g['nstate']), 1e-300), 0).max():.2e}"
          f"  max |err|/bound {np.max(es / shi):.2e}   ", end="")
    ob = env.get(L.VS_OBS).astype(np.float64)
    eo = np.abs(ob - g["obs"])
    rw = env.get(L.VS_REW).astype(np.float64)
    er = np.abs(rw - g["rew"]) / np.maximum(np.abs(g["rew"]), 1e-300)
    big = np.abs(g["rew"]) > 1e-30
    print(f"obs max abs {eo.max():.2e}  rew max rel {er[big].max():.2e} (min |rew| {np.abs(g['rew'][big]).min():.1e})", end="")
    if g["hidden"].shape[1]:
        eh = np.abs(env.get(L.VS_HIDDEN).astype(np.float64) - g["nhidden"])
        print(f"  hidden max abs {eh.max():.2e} rel {np.max(eh / np.maximum(np.abs(g['nhidden']), 1e-2)):.2e}", end="")
    print()
    env.close()
