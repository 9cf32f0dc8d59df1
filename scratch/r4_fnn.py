"""vs_step_policy alone (k_rollout_fnn): a 64 x 64 tanh network + exploration noise in the kernel, k steps per launch,
auto-reset, records on.  VS_FNN_SHAPE=64|256|mfma pins the workgroup shape / the matrix-core path (one value per process)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simurlacra_amd as vs  # noqa: E402
from bench import ENV_KW  # noqa: E402

hidden = [int(x) for x in os.environ.get("FNN_HIDDEN", "64,64").split(",")]
for name, n in (("qq-su", 4096), ("qq-su", 16384), ("qq-su", 32768), ("qq-su", 65536), ("qq-su", 262144), ("qcp-su", 65536), ("qbb", 32768)):
    e = vs.VecSimEnv(name, n, **ENV_KW[name])
    O, A = e.dims["O"], e.dims["A"]
    net = vs.FNN(O, A, hidden, torch.tanh)
    e.set_policy_fnn(net.param_values, hidden, "tanh", noise_std=np.full(A, 0.1, dtype=np.float32))
    e.set_auto_reset(True, seed=1)
    e.reset(seed=2)
    e.set_traj_capacity(200)
    for _ in range(2):
        e.step_policy(200, record=True, noise_seed=3)
    e.sync()
    e.timer_start()
    for _ in range(5):
        e.step_policy(200, record=True, noise_seed=3)
    ms = e.timer_stop() / 5
    print(json.dumps(dict(shape=os.environ.get("VS_FNN_SHAPE", "auto"), env=name, envs=n, net="x".join(map(str, hidden)) + " tanh + noise",
                          us_per_step=round(ms * 1e3 / 200, 3), env_steps_per_s=round(n * 200 / (ms * 1e-3)))), flush=True)
    e.close()
