"""
vs_step_policy: rollout() with a feed-forward network policy evaluated inside the fused kernel (P/sampling/rollout.py:185-258 with
act = policy(obs); FNN.forward P/policies/feed_back/fnn.py:139-160; the fork's FNNPolicy.forward featurisation fnn.py:219-222).

What is checked, through the C-ABI:
  * the recorded action of every step equals the torch network on the recorded observation of that step -- fp32 FMAs in a
    different summation order than torch's GEMM and a v_exp-based tanh: |act - torch| <= 1e-5 (1 + |torch|) asserted (measured 3.4e-6), printed with -s;
  * everything else is the step kernel's: vs_step fed with the recorded actions from the same initial state reproduces the
    recorded observations, states, rewards and done flags BIT FOR BIT (with and without auto-reset, launches split unevenly);
  * exploration noise: (act - network(obs)) / std is N(0, 1) by its moments and does not depend on how the steps are cut
    into launches;
  * the sampler: ParallelRolloutSampler with an FNNPolicy takes the fused path and returns rollouts whose every step is
    consistent with the policy and whose first steps equal the torch-in-the-loop path's.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

KW = {"omo": dict(dt=0.02, max_steps=40), "bob": dict(dt=0.01, max_steps=40), "qq-su": dict(dt=0.004, max_steps=40),
      "qcp-su": dict(dt=0.002, max_steps=40), "qbb": dict(dt=0.01, max_steps=40), "qq-st": dict(dt=0.01, max_steps=40),
      "pend": dict(dt=0.02, max_steps=40, init_state=np.array([0.1, 0.2]))}
NL = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, None: None}


@pytest.fixture(scope="module")
def vs():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import simurlacra_amd

    return simurlacra_amd


def dev(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).cuda()


def make_net(vs, in_dim, out_dim, hidden, nonlin, out_nonlin, gain, seed):
    from simurlacra_amd.policies import FNN

    torch.manual_seed(seed)
    net = FNN(in_dim, out_dim, hidden, [NL[f] for f in nonlin] if isinstance(nonlin, list) else NL[nonlin],
              output_nonlin=NL[out_nonlin])
    with torch.no_grad():
        net.output_layer.weight.mul_(gain)  # actions that reach (and leave) the action box
    return net


def features(obs, idx, feat):
    x = obs[..., idx] if idx is not None else obs
    if feat:
        x = np.concatenate([x[..., 0:1], np.sin(x[..., 1:2]), np.cos(x[..., 1:2]), x[..., 2:]], axis=-1)
    return x


CASES = [  # family, hidden sizes, hidden nonlin, output nonlin, featurisation, visible rows, output gain
    ("qq-su", [64, 64], "tanh", None, False, None, 20.0),
    ("qq-su", [32], "relu", "tanh", False, [0, 2, 4, 5], 3.0),
    ("qcp-su", [64, 64], "tanh", None, True, None, 30.0),
    ("qbb", [40, 24, 16], ["tanh", "relu", "sigmoid"], None, False, None, 10.0),
    ("omo", [16, 16, 16, 16], "tanh", None, False, None, 100.0),
    ("bob", [64], "tanh", None, True, None, 20.0),
    ("pend", [8, 64], "relu", None, False, [2, 0], 5.0),
    ("qq-su", [32, 24], "tanh", None, False, None, 10.0),  # two narrow layers: the automatic choice is the matrix-core shape
]


@pytest.mark.parametrize("shape", [None, "256", "mfma"])
@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_policy_kernel_against_torch_and_the_step_kernel(vs, case, auto_reset, shape):
    """shape: how k_rollout_fnn evaluates the network -- None: the automatic choice (here: 64-env workgroups, lane = hidden unit
    on the vector ALU; the matrix-core shape for two hidden layers of at most 32 units), '256': the same in 256-env workgroups, 'mfma': the hidden layers on the matrix cores
    (v_mfma_f32_32x32x2_f32, fp32; one and two hidden layers, tiles of a narrow layer skipped)"""
    L = vs._lib
    name, hidden, nonlin, out_nonlin, feat, idx, gain = CASES[case]
    if shape is not None and len(hidden) > 2:
        pytest.skip("three and four hidden layers run the 64-env shape whatever is asked")
    n, splits = 700, (7, 1, 30, 12)
    T = sum(splits)
    O, A = vs.env_dims(name)["O"], vs.env_dims(name)["A"]
    n_vis = len(idx) if idx is not None else O
    net = make_net(vs, n_vis + int(feat), A, hidden, nonlin, out_nonlin, gain, seed=case)
    params = torch.nn.utils.parameters_to_vector(net.parameters())
    per_env = case % 2 == 0
    envs = []
    for _ in range(2):
        e = vs.VecSimEnv(name, n, **KW[name])
        if per_env:
            e.set_params(np.tile(vs.nominal_params(name), (n, 1)))
        e.set_auto_reset(auto_reset, seed=31)
        e.reset(seed=5 + case)
        envs.append(e)
    pol, ref = envs
    pol.set_policy_fnn(params, hidden, nonlin, out_nonlin, feat=feat, obs_idx=idx)
    pol.set_policy_shape(shape)
    pol.set_record_mode(2)
    pol.set_traj_capacity(T)
    t = 0
    for k in splits:
        pol.set_traj_offset(t)
        pol.step_policy(k, record=True)
        t += k
    tr = pol.traj(T)
    # (1) the network: recorded action against torch on the recorded observation
    with torch.no_grad():
        want = net(torch.from_numpy(features(tr["obs"], idx, feat).astype(np.float32))).numpy()
    err = np.abs(tr["act"] - want) / (1.0 + np.abs(want))
    print(f"{name} {hidden} {nonlin} shape {shape}: max |act - torch| / (1 + |torch|) = {err.max():.2e}; |act| up to {np.abs(want).max():.1f}")
    assert err.max() < 1e-5  # measured: <= 3.4e-6 on the vector ALU, <= 4.6e-6 on the matrix cores
    # (2) the step: vs_step with the recorded actions from the same initial state, bit for bit
    alive = np.ones(n, dtype=bool)
    for t in range(T):
        assert np.array_equal(ref.get(L.VS_OBS)[alive], tr["obs"][t][alive]), (name, t)
        assert np.array_equal(ref.get(L.VS_STATE)[alive], tr["state"][t][alive]), (name, t)
        ref.step(dev(tr["act"][t]))
        assert np.array_equal(ref.get(L.VS_REW)[alive], tr["rew"][t][alive]), (name, t)
        assert np.array_equal(ref.get(L.VS_DONE).astype(bool)[alive], tr["done"][t].astype(bool)[alive]), (name, t)
        if not auto_reset:
            alive &= ~tr["done"][t].astype(bool)
    for which in (L.VS_STATE, L.VS_HIDDEN, L.VS_STEPCOUNT, L.VS_RETURNS):
        assert np.array_equal(ref.get(which)[alive], pol.get(which)[alive]), (name, which)
    assert tr["done"].any() and pol.error_count() == 0
    if auto_reset:
        for x, y in zip(pol.episode_stats(), ref.episode_stats()):
            assert np.array_equal(x, y)
    for e in envs:
        e.close()


def test_policy_kernel_exploration_noise(vs):
    name, n, T = "qbb", 4096, 24
    O, A = vs.env_dims(name)["O"], vs.env_dims(name)["A"]
    net = make_net(vs, O, A, [32, 32], "tanh", None, 1.0, seed=3)
    params = torch.nn.utils.parameters_to_vector(net.parameters())
    std = np.array([0.3, 0.05], dtype=np.float32)
    out = []
    for splits in ((24,), (5, 19)):
        e = vs.VecSimEnv(name, n, **KW[name])
        e.set_auto_reset(True, seed=2)
        e.reset(seed=3)
        e.set_policy_fnn(params, [32, 32], "tanh", None, noise_std=std)
        e.set_traj_capacity(T)
        t = 0
        for k in splits:
            e.set_traj_offset(t)
            e.step_policy(k, record=True, noise_seed=77)
            t += k
        out.append(e.traj(T))
        e.close()
    a, b = out
    for key in a:
        assert np.array_equal(a[key], b[key]), key  # the noise is keyed by (env, episode, step), not by the launch
    with torch.no_grad():
        mean = net(torch.from_numpy(a["obs"].astype(np.float32))).numpy()
    z = (a["act"] - mean) / std
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs((z[..., 0] * z[..., 1]).mean()) < 0.01
    assert abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3.0) < 0.1
    c = vs.VecSimEnv(name, n, **KW[name])
    c.set_auto_reset(True, seed=2)
    c.reset(seed=3)
    c.set_policy_fnn(params, [32, 32], "tanh", None, noise_std=std)
    c.set_traj_capacity(T)
    c.step_policy(T, record=True, noise_seed=78)
    assert not np.array_equal(c.traj(T)["act"], a["act"])  # another seed, another draw
    c.close()


def test_policy_kernel_argument_errors(vs):
    e = vs.VecSimEnv("qq-su", 64, **KW["qq-su"])
    with pytest.raises(RuntimeError):
        e.step_policy(1)  # no network yet
    with pytest.raises(vs.ValueErr):
        e.set_policy_fnn(np.zeros(10), [128], "tanh")  # wider than the kernel's 64
    with pytest.raises(vs.ValueErr):
        e.set_policy_fnn(np.zeros(10), [8], "tanh")  # parameter count does not match 6 -> 8 -> 1
    e.set_policy_fnn(np.zeros(6 * 8 + 8 + 8 + 1), [8], "tanh")
    e.step_policy(3)
    e.set_act_pipeline(delay=1)
    with pytest.raises(RuntimeError):
        e.step_policy(1)  # wrapper pipeline on the handle
    e.set_act_pipeline(delay=0)
    e.set_policy_fnn(None, [])
    with pytest.raises(RuntimeError):
        e.step_policy(1)
    e.close()
    d = vs.VecSimEnv("bob-d", 64, dt=0.01, max_steps=10)
    with pytest.raises(vs.ValueErr):
        d.set_policy_fnn(np.zeros(4 * 8 + 8 + 8 + 1), [8], "tanh")  # discrete actions
    d.close()


@pytest.mark.parametrize("envname", ["qq-su", "qcp-su"])
def test_sampler_takes_the_fused_policy_path(vs, envname):
    from simurlacra_amd.policies import FNNPolicy, NormalActNoiseExplStrat, fnn_kernel_spec

    cls = {"qq-su": vs.QQubeSwingUpSim, "qcp-su": vs.QCartPoleSwingUpSim}[envname]
    env = cls(dt=KW[envname]["dt"], max_steps=60)
    torch.manual_seed(0)
    policy = FNNPolicy(env.spec, [32, 32], torch.tanh, featurize=envname == "qcp-su")
    assert fnn_kernel_spec(policy) is not None
    fused = vs.ParallelRolloutSampler(env, policy, 1, min_rollouts=300, seed=4)
    loop = vs.ParallelRolloutSampler(env, policy, 1, min_rollouts=300, seed=4, fuse_policy=False)
    ros_f, ros_l = fused.sample(), loop.sample()
    assert len(ros_f) == len(ros_l) == 300
    pol = policy.to("cpu")
    worst = 0.0
    for rf, rl in zip(ros_f, ros_l):
        # same initial state, and every fused step is the policy's action on the recorded observation
        assert np.array_equal(rf.states[0], rl.states[0])
        with torch.no_grad():
            want = pol(torch.from_numpy(np.asarray(rf.observations[:-1], dtype=np.float32))).numpy()
        worst = max(worst, float((np.abs(rf.actions - want) / (1 + np.abs(want))).max()))
        assert rf.states.shape == (len(rf) + 1, env.state_space.flat_dim) and rf.actions_applied.shape == rf.actions.shape
        k = min(5, len(rf), len(rl))  # the first steps agree with the torch-in-the-loop path (before rounding differences grow)
        np.testing.assert_allclose(rf.observations[:k], rl.observations[:k], rtol=2e-4, atol=2e-5)
    assert worst < 1e-5
    # with exploration noise: still the fused path; eval=True keeps the noise -- like the reference (rollout() only calls
    # policy.eval(); StochasticActionExplStrat.forward samples whatever the mode) and like the torch-in-the-loop path here
    noisy = NormalActNoiseExplStrat(policy, std_init=0.5)
    assert fnn_kernel_spec(noisy)["noise_std"] is not None
    smp = vs.ParallelRolloutSampler(env, noisy, 1, min_rollouts=64, seed=4)
    r_noise, r_eval = smp.sample(), smp.sample(eval=True)
    smp_torch = vs.ParallelRolloutSampler(env, noisy, 1, min_rollouts=64, seed=4, fuse_policy=False)
    r_eval_torch = smp_torch.sample(eval=True)

    def noise_of(rollouts):
        pdev = next(pol.parameters()).device  # (the torch-in-the-loop sampler has moved the policy to the GPU)
        with torch.no_grad():
            return np.concatenate([ro.actions - pol(torch.from_numpy(np.asarray(ro.observations[:-1], dtype=np.float32)).to(pdev)).cpu().numpy()
                                   for ro in rollouts]).ravel()

    z_noise, z_eval, z_torch = noise_of(r_noise), noise_of(r_eval), noise_of(r_eval_torch)
    for z in (z_noise, z_eval, z_torch):  # N(0, 0.5^2) on every path, evaluation mode or not
        assert len(z) > 500 and abs(z.std() - 0.5) < 0.06 and abs(z.mean()) < 0.08, (len(z), z.std(), z.mean())
