import sys; sys.path.insert(0, '.')
import numpy as np, torch
import simurlacra_amd as vs
L = vs._lib
def run(noise, n=64, T=300, chunk=100):
    e = vs.VecSimEnv("omo", n, dt=0.02, max_steps=300)
    e.set_params_uniform(dict(mass=1.0, stiffness=30.0, damping=0.5))
    if noise:
        e.set_act_pipeline(noise_std=[1.0], seed=5)
    e.set_auto_reset(False)
    e.reset(seed=9)
    e.set_traj_capacity(T)
    for t in range(0, T, chunk):
        e.set_traj_offset(t)
        e.step_random(chunk, seed=4, record=True)
    e.set_traj_offset(0)
    tr = e.traj(T)
    done = tr["done"].astype(bool)
    first = np.where(done.any(0), done.argmax(0), T - 1) + 1
    e.close()
    return first, tr
f0, t0 = run(False)
for rep in range(3):
    f1, t1 = run(True)
    print("lengths plain", f0[:10], "noisy", f1[:10], "act equal", np.array_equal(t0["act"][:20], t1["act"][:20]))
    print(" act[0:5,0]", t1["act"][:5, 0, 0], "obs", t1["obs"][:3, :, 0].ravel())
