"""Round-2 kernel sweep on one GPU: fused-rollout variants x sizes x record modes, kernel time by HIP events.
usage: python scratch/r2_sweep.py [section ...]   sections: probes qq fam cfg big"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import simurlacra_amd as vs  # noqa: E402
from simurlacra_amd import _lib as L  # noqa: E402
from bench import ENV_KW, DIMS, bytes_fused_step  # noqa: E402

SECTIONS = sys.argv[1:] or ["probes", "qq", "fam", "cfg", "big"]


def run(name, n, variant, rec, chunk=100, per_env=True, live=0, iters=20, gib=1.0):
    kw = ENV_KW[name]
    env = vs.VecSimEnv(name, n, **kw)
    if per_env:
        env.set_params(np.tile(vs.nominal_params(name), (n, 1)))
    if live:
        rz = vs.create_default_randomizer(vs.ENV_CLASSES[name](**kw))
        env.set_randomizer(rz.device_specs()[:live])
    env.set_auto_reset(True, seed=1)
    env.reset(seed=2)
    env.set_rollout_variant(variant)
    got = env.rollout_variant()
    if variant is not None and got != variant:
        env.close()
        return None
    if rec:
        env.set_record_mode(rec)
        F = env.traj_layout()[0]
        slot = chunk * F * env.ld * 4
        slots = max(1, int(np.ceil(gib * 2 ** 30 / slot)))
        env.set_traj_capacity(chunk * slots)
    env.step_random(chunk, seed=3, record=bool(rec))
    env.sync()
    ms = env.time_step_kernel(iters=iters, k_steps=chunk, record=bool(rec))
    errs = env.error_count()
    env.close()
    rate = n * chunk / (ms * 1e-3)
    d = dict(DIMS[name])
    bps = bytes_fused_step(d, chunk, rec) if rec < 2 else (4 * (d["O"] + 2 * d["A"] + 1 + d["S"] + d["H"]) + 0.125)
    print(f"{name:7s} n={n:8d} {('auto:' if variant is None else '') + got:20s} rec={rec} chunk={chunk:3d} live={live} | {ms * 1e3:9.1f} us/launch "
          f"{ms * 1e6 / chunk:8.1f} ns/step | {rate:.3e} env-steps/s | {rate * bps / 1e9:7.0f} GB/s alg | nan={errs}", flush=True)
    return ms


if "probes" in SECTIONS:
    g = ctypes.c_float()
    lib = L.load()
    for nbytes in (1 << 28, 1 << 30, 1 << 32):
        lib.vs_membw_probe(0, nbytes, 10, ctypes.byref(g))
        c = g.value
        lib.vs_memwrite_probe(0, nbytes, 10, ctypes.byref(g))
        print(f"probe {nbytes >> 20:5d} MiB: copy {c:7.0f} GB/s  write {g.value:7.0f} GB/s", flush=True)

if "qq" in SECTIONS:
    for n in (4096, 16384, 32768, 65536):
        for rec in (0, 1, 2):
            for var in ("k_rollout", "k_rollout_ws", "k_rollout_ws64"):
                run("qq-su", n, var, rec)
    for n in (131072, 262144):
        for rec in (0, 1, 2):
            run("qq-su", n, "k_rollout", rec)
    run("qq-su", 65536, "k_rollout_ws", 1, per_env=False)

if "fam" in SECTIONS:
    for name in ("omo", "bob", "qcp-su", "qbb", "qq-st", "pend", "bob-d", "qcp-st"):
        for var in ("k_rollout", "k_rollout_ws", "k_rollout_ws64"):
            run(name, 65536, var, 1)

if "cfg" in SECTIONS:
    for var in ("k_rollout", "k_rollout_ws", "k_rollout_ws64"):
        run("qcp-su", 65536, var, 1, live=7)  # config 3
    for var in ("k_rollout", "k_rollout_ws", "k_rollout_ws64"):
        run("qbb", 32768, var, 1)  # config 4 shard
    for var in ("k_rollout", "k_rollout_ws", "k_rollout_ws64"):
        run("qq-su", 65536, var, 1, live=7)

if "auto" in SECTIONS:  # is the automatic choice the fastest?  every family x size x (auto, the three pinned kernels)
    for name in ("omo", "bob", "qq-su", "qcp-su", "qbb", "qq-st", "pend", "bob-d"):
        for n in (32768, 65536, 98304):
            for var in (None, "k_rollout", "k_rollout_ws", "k_rollout_ws64") + (
                    ("k_rollout_ws64g", "k_rollout_ws256g") if name in ("omo", "qq-su", "qq-st", "pend") else ()):
                if name == "qbb" and n > 32768 and var == "k_rollout_ws64":
                    continue
                run(name, n, var, 1)
    for name, live in (("qq-su", 7), ("qcp-su", 7)):
        for n in (32768, 65536, 98304):
            for var in (None, "k_rollout", "k_rollout_ws", "k_rollout_ws64"):
                run(name, n, var, 1, live=live)

if "mid" in SECTIONS:
    for n in (49152, 81920, 98304, 131072):
        for rec in (0, 1):
            for var in ("k_rollout", "k_rollout_ws", "k_rollout_ws64"):
                run("qq-su", n, var, rec)

if "big" in SECTIONS:
    for n in (524288, 1048576):
        for rec in (0, 1):
            run("qq-su", n, "k_rollout", rec, chunk=20, gib=2.0)
