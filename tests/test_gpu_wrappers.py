"""GPU part of the wrapper-stack tests: the action / observation pipeline fused into the step
kernels (vs_set_act_pipeline, vs_set_obs_pipeline) through the C-ABI, against

  (1) trajectories the reference's own wrapper objects produced (tests/golden/chains.npz) for every deterministic part:
      delay queues, (de-)normalisation, noise means, partial observations -- tolerances as in tests/test_gpu_parity.py;
  (2) the statistics of the Philox noise (the reference draws from NumPy's global RNG, so noise values cannot be equal
      sample by sample: mean / std / independence are what the wrappers specify);
  (3) invariances: fused rollout kernel == step kernel bit for bit, shards == whole batch, replay determinism.
"""
import numpy as np
import pytest

from oracle import cpu_ref
from test_wrapper_chains import KW, build_chain, load_chains, oracle_stages

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def vs():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import simurlacra_amd

    return simurlacra_amd


def f32(x):
    return np.asarray(x, dtype=np.float32)


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(f32(x))).cuda()


DETERMINISTIC = ["qq_delay3", "qq_norm_delay_bias", "qq_obsnorm_over_bias", "bob_partial_norm_delay"]
NOISY = ["qbb_noise_over_norm", "qbb_noise_norm_partial", "qcp_delay_over_noise", "omo_everything"]


@pytest.mark.parametrize("tag", DETERMINISTIC + NOISY)
def test_fused_chain_against_reference_trajectories(vs, golden_dir, tag):
    """Episodes of the reference's wrapped envs as lanes.  The inner state is re-synchronised to the reference before
    every step (one-step tolerances; the delay ring and the step counters live on the device across the steps).
    For the stacks with random noise the noise std is set to zero on BOTH sides of the comparison: the oracle's
    sequential WrappedRef (pinned against the reference with the noise on, tests/test_wrapper_chains.py) supplies the
    expected values, so that delay / normalisation / bias handling is checked on those stacks too."""
    L = vs._lib
    g, spec, _ = load_chains(golden_dir)
    name, stages = spec[tag]["env"], spec[tag]["stages"]
    quiet = [dict(st, std=[0.0] * len(st["std"])) if "std" in st else st for st in stages]
    chain = build_chain(name, quiet)
    fc = vs.fuse_wrappers(chain)
    ref = cpu_ref.make_ref(name, **KW[name])
    n = g[f"{tag}__s0"].shape[0]
    P = np.tile(g[f"{tag}__params"], (n, 1))
    wr = cpu_ref.WrappedRef(ref, oracle_stages(chain, quiet), lambda m, w: np.zeros((m, w)))
    env = vs.VecSimEnv(name, n, **KW[name])
    fc.apply(env, seed=3)
    keep = fc.keep
    state, hidden = g[f"{tag}__s0"], g[f"{tag}__h0"]
    env.reset(init_state=f32(state))
    obs0 = wr.reset(state)
    if not name.startswith("qcp"):  # QCartPoleSim.reset returns the 4-D state instead of the observation (Q5)
        np.testing.assert_allclose(env.get(L.VS_OBS)[:, keep], obs0, rtol=1e-5, atol=2e-6)
    if tag in DETERMINISTIC:  # ... which is what the reference returned
        np.testing.assert_allclose(env.get(L.VS_OBS)[:, keep], g[f"{tag}__obs0"], rtol=1e-5, atol=2e-6)
    if ref.H:
        hidden = env.get(L.VS_HIDDEN).astype(np.float64)
    T = int(g[f"{tag}__length"].min())
    for t in range(T):
        act = g[f"{tag}__act"][:, t]
        exp = wr.step(state, hidden, act, P, np.full(n, t))
        env.step(dev(act))
        slo, shi, _, _ = ref.bounds(P)
        tol = 1e-5 * np.abs(exp["state"]) + 2e-6 * np.maximum(1.0, np.abs(shi))
        assert (np.abs(env.get(L.VS_STATE) - exp["state"]) <= tol).all(), (tag, t)
        np.testing.assert_allclose(env.get(L.VS_OBS)[:, keep], exp["obs"], rtol=2e-5, atol=2e-5, err_msg=f"{tag} t={t}")
        np.testing.assert_allclose(env.get(L.VS_REW), exp["rew"], rtol=5e-5, atol=1e-12)
        assert np.array_equal(env.get(L.VS_DONE).astype(bool), exp["done"])
        if tag in DETERMINISTIC:
            np.testing.assert_allclose(exp["obs"], g[f"{tag}__obs"][:, t], rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(env.get(L.VS_REW), g[f"{tag}__rew"][:, t], rtol=5e-5, atol=1e-12)
        state, hidden = exp["state"], exp["hidden"]
        env.put(L.VS_STATE, f32(state))  # re-synchronise (fp64 -> fp32 rounding only)
        if ref.H:
            env.put(L.VS_HIDDEN, f32(hidden))
    assert env.error_count() == 0
    env.close()


def test_observation_noise_statistics_and_keys(vs):
    """z recovered from VS_OBS is N(0, 1), independent across envs, dims and steps; a pure function of
    (seed, global env index, episode, step)"""
    L = vs._lib
    n, T = 32768, 12
    kw = KW["qq-su"]
    ref = cpu_ref.make_ref("qq-su", **kw)
    scale = np.array([1.0, 2.0, 0.5, 1.0, 0.05, 0.04])
    shift = np.array([0.0, -1.0, 0.25, 0.0, 0.1, 0.0])
    std = np.array([0.1, 0.2, 0.05, 0.3, 1.5, 0.01])
    env = vs.VecSimEnv("qq-su", n, **kw)
    env.set_obs_pipeline(scale, shift, std, seed=99)
    env.reset(seed=5)

    def z_now(e):
        inner = ref.observe(e.get(L.VS_STATE).astype(np.float64))
        return (e.get(L.VS_OBS).astype(np.float64) - (inner * scale + shift)) / std

    zs = [z_now(env)]
    act = dev(np.zeros((n, 1)))
    for t in range(T):
        env.step(act)
        zs.append(z_now(env))
    z = np.stack(zs)  # [T + 1, n, O]
    assert abs(z.mean()) < 4.0 / np.sqrt(z.size) + 1e-3
    np.testing.assert_allclose(z.std(axis=(0, 1)), 1.0, atol=0.01)
    np.testing.assert_allclose(z.mean(axis=(0, 1)), 0.0, atol=0.01)
    assert abs((z ** 4).mean() - 3.0) < 0.05  # Gaussian kurtosis
    flat = z.reshape(-1, 6)
    c = np.corrcoef(flat.T)
    assert np.abs(c - np.eye(6)).max() < 0.01  # across dims (Box-Muller pairs included)
    lag = (z[1:] * z[:-1]).mean()
    assert abs(lag) < 0.005  # across steps
    assert abs((z[:, 1:] * z[:, :-1]).mean()) < 0.005  # across neighbouring envs
    # same seed, same envs -> same noise; other seed -> other noise; a shard with an index offset == its slice
    env2 = vs.VecSimEnv("qq-su", n, **kw)
    env2.set_obs_pipeline(scale, shift, std, seed=99)
    env2.reset(seed=5)
    for t in range(T):
        env2.step(act)
    assert np.array_equal(env2.get(L.VS_OBS), env.get(L.VS_OBS))
    env2.set_obs_pipeline(scale, shift, std, seed=100)
    assert not np.array_equal(env2.get(L.VS_OBS), env.get(L.VS_OBS))
    half = n // 2
    sh = vs.VecSimEnv("qq-su", half, **kw)
    sh.set_index_offset(half)
    sh.set_obs_pipeline(scale, shift, std, seed=99)
    sh.reset(seed=5)
    for t in range(T):
        sh.step(act[half:])
    assert np.array_equal(sh.get(L.VS_OBS), env.get(L.VS_OBS)[half:])
    for e in (env, env2, sh):
        e.close()


def test_action_noise_statistics_through_the_linear_oscillator(vs):
    """OneMassOscillator is linear with forward Euler (one_mass_oscillator.py:113-114): the applied force is recovered
    from the state change, so the action noise can be measured.  ActNorm inside / outside the noise wrapper."""
    L = vs._lib
    n = 65536
    kw = KW["omo"]
    m_, k_, d_ = 1.0, 30.0, 0.5
    for normed in (False, True):
        env = vs.VecSimEnv("omo", n, **kw)
        env.set_act_norm(True)
        env.set_act_pipeline(delay=0, noise_mean=[0.1], noise_std=[0.2], noise_normed=normed, seed=7)
        s0 = np.tile(np.array([[-0.7, 0.05]]), (n, 1))
        env.reset(init_state=f32(s0))
        env.step(dev(np.full((n, 1), 0.25)))  # policy action in [-1, 1]; act bound = stiffness = 30
        s1 = env.get(L.VS_STATE).astype(np.float64)
        x, v = np.float32(-0.7).astype(np.float64), np.float32(0.05).astype(np.float64)
        om2, twozo = k_ / m_, d_ / m_
        force = m_ * ((s1[:, 1] - v) / kw["dt"] + om2 * x + twozo * v)
        unit = 30.0 if normed else 1.0
        assert abs(force.mean() - (0.25 * 30.0 + 0.1 * unit)) < 0.02 * unit + 2e-3
        assert abs(force.std() - 0.2 * unit) < 0.01 * unit + 2e-3
        zz = (force - force.mean()) / force.std()
        assert abs((zz ** 3).mean()) < 0.05 and abs((zz ** 4).mean() - 3.0) < 0.1
        assert abs((zz[1:] * zz[:-1]).mean()) < 0.02
        # the reward sees the processed action (the inner env's step computes it): rew = -(e'Qe + a'Ra), R = 1e-6
        rew = env.get(L.VS_REW).astype(np.float64)
        exp_rew = -(10.0 * x * x + 1e-2 * v * v + 1e-6 * force ** 2)
        np.testing.assert_allclose(rew, exp_rew, rtol=2e-4, atol=1e-6)
        env.close()


@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("name", ["qq-su", "qbb", "omo"])
def test_fused_rollout_equals_step_kernel_with_pipeline(vs, name, auto_reset):
    """vs_step_random (k_rollout) and vs_step (k_step) run the same pipeline code on the same Philox streams: records
    and final buffers are bit-identical, including delay ring contents across episode boundaries and noise keys"""
    L = vs._lib
    n, T = 1024, 48
    kw = dict(KW[name], max_steps=20)  # time-outs inside the window
    A, O = {"qq-su": (1, 6), "qbb": (2, 8), "omo": (1, 2)}[name]
    a = vs.VecSimEnv(name, n, **kw)
    b = vs.VecSimEnv(name, n, **kw)
    for e in (a, b):
        e.set_act_norm(True)
        e.set_act_pipeline(delay=3, noise_mean=np.full(A, 0.05), noise_std=np.full(A, 0.1), noise_normed=True,
                           noise_after_delay=(name == "qbb"), seed=11)
        e.set_obs_pipeline(np.linspace(0.5, 1.5, O), np.linspace(-0.1, 0.1, O), np.linspace(0.01, 0.05, O), seed=12)
        e.set_auto_reset(auto_reset, seed=17)
        e.reset(seed=1)
    assert np.array_equal(a.get(L.VS_OBS), b.get(L.VS_OBS))
    a.step_random(T, seed=4, record=True)
    tr = a.traj(T)
    alive = np.ones(n, dtype=bool)
    for t in range(T):
        assert np.array_equal(b.get(L.VS_OBS)[alive], tr["obs"][t][alive]), t
        b.step(dev(tr["act"][t]))
        assert np.array_equal(b.get(L.VS_REW)[alive], tr["rew"][t][alive]), t
        assert np.array_equal(b.get(L.VS_DONE).astype(bool)[alive], tr["done"][t].astype(bool)[alive])
        if not auto_reset:
            alive &= ~tr["done"][t].astype(bool)
    assert tr["done"].any()
    assert np.abs(tr["act"]).max() <= 1.0  # the record holds the policy's (normalised) action
    for which in (L.VS_STATE, L.VS_OBS, L.VS_HIDDEN, L.VS_STEPCOUNT, L.VS_RETURNS):
        assert np.array_equal(a.get(which)[alive], b.get(which)[alive])
    a.close()
    b.close()


def test_delay_queue_restarts_with_zero_actions_after_auto_reset(vs):
    """ActDelayWrapper.reset refills the queue with zeros (action_delay.py:87-93): after an auto-reset the first
    `delay` applied actions are zero again, whatever the ring still holds"""
    L = vs._lib
    n, delay = 256, 4
    kw = dict(dt=0.02, max_steps=6)
    env = vs.VecSimEnv("omo", n, **kw)
    env.set_act_pipeline(delay=delay)
    env.set_auto_reset(True, seed=3)
    s0 = np.tile(np.array([[-0.7, 0.0]]), (n, 1))
    env.reset(init_state=f32(s0))
    forces = []
    for t in range(14):
        before = env.get(L.VS_STATE).astype(np.float64)
        step_before = env.get(L.VS_STEPCOUNT)
        env.step(dev(np.full((n, 1), 10.0 + t)))
        after = env.get(L.VS_STATE).astype(np.float64)
        fresh = env.get(L.VS_STEPCOUNT) == 0  # lanes that were reset in this launch: their state is a new init state
        f = 1.0 * ((after[:, 1] - before[:, 1]) / 0.02 + 30.0 * before[:, 0] + 0.5 * before[:, 1])
        f[fresh] = np.nan
        forces.append((step_before.copy(), f))
    for t, (sb, f) in enumerate(forces):
        ok = ~np.isnan(f)
        early = ok & (sb < delay)
        late = ok & (sb >= delay)
        np.testing.assert_allclose(f[early], 0.0, atol=2e-3)
        if late.any():
            np.testing.assert_allclose(f[late], 10.0 + t - delay, atol=2e-3)
    assert any((sb < delay).all() for sb, _ in forces[6:])  # a second episode was observed
    env.close()


def test_pipeline_argument_errors_and_jacobian_refusal(vs):
    L = vs._lib
    env = vs.VecSimEnv("qq-su", 64, **KW["qq-su"])
    with pytest.raises(vs.ValueErr):
        env.set_act_pipeline(delay=L.VS_MAX_ACT_DELAY + 1)
    with pytest.raises(vs.ValueErr):
        env.set_act_pipeline(noise_std=[np.nan])
    with pytest.raises(Exception):
        env.set_act_pipeline(noise_std=[-1.0])
    env.set_act_pipeline(delay=2)
    with pytest.raises(Exception, match="pipeline"):
        env.step_jac(dev(np.zeros((64, 1))))
    env.set_act_pipeline(delay=0)
    env.step_jac(dev(np.zeros((64, 1))))
    env.set_obs_pipeline(noise_std=np.full(6, 0.1))
    with pytest.raises(Exception, match="pipeline"):
        env.step_jac(dev(np.zeros((64, 1))))
    env.set_obs_pipeline()  # identity again: VS_OBS is the bare observation
    ref = cpu_ref.make_ref("qq-su", **KW["qq-su"])
    np.testing.assert_allclose(env.get(L.VS_OBS), ref.observe(env.get(L.VS_STATE).astype(np.float64)), rtol=1e-6, atol=5e-7)
    env.close()


def test_sampler_runs_a_wrapped_chain_on_the_device(vs, golden_dir):
    """ParallelRolloutSampler over ObsPartial(ObsNorm(ActNorm(ActDelay(env)))): observation width, bounds, and the
    same rollouts whatever the batch size"""
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import ParallelRolloutSampler

    env = vs.ENV_CLASSES["bob"](dt=0.01, max_steps=60)
    chain = vs.ObsPartialWrapper(vs.ObsNormWrapper(vs.ActNormWrapper(vs.ActDelayWrapper(env, delay=2))), mask=[0, 0, 0, 1])
    pol = DummyPolicy(chain.spec)
    s1 = ParallelRolloutSampler(chain, pol, 4, min_rollouts=96, seed=3)
    s2 = ParallelRolloutSampler(chain, pol, 4, min_rollouts=96, seed=3, batch_lanes=32)
    r1, r2 = s1.sample(), s2.sample()
    assert len(r1) == len(r2) == 96
    for x, y in zip(r1, r2):
        assert np.array_equal(x.observations, y.observations) and np.array_equal(x.actions, y.actions)
        assert np.array_equal(x.rewards, y.rewards)
    ro = r1[0]
    assert ro.observations.shape == (len(ro) + 1, 3) and ro.actions.shape == (len(ro), 1)
    assert np.abs(ro.actions).max() <= 1.0
    allobs = np.concatenate([r.observations[:-1] for r in r1])
    assert np.abs(allobs).max() <= 1.0 + 1e-5  # normalised, inside the state box until the last step

    class Lin(vs.Policy):
        def forward(self, obs):
            return 0.1 * obs[..., :1]

    s3 = ParallelRolloutSampler(chain, Lin(chain.spec), 4, min_rollouts=16, seed=3)
    r3 = s3.sample()
    assert r3[0].observations.shape[1] == 3 and len(r3) == 16
    # policy in the loop: a_t = 0.1 * obs_t[0] recorded, and the recorded observation is what the policy saw
    np.testing.assert_allclose(r3[0].actions[:, 0], 0.1 * r3[0].observations[:-1, 0], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("wrap", ["delay", "noise"])
def test_sampler_is_repeatable_and_batch_invariant_with_a_pipeline(vs, wrap):
    """regression: the sampler's torch reads of the record buffers are ordered with the kernels (the handle's stream is
    a blocking stream / torch's current stream), so repeated sampling and different batch cuts give identical rollouts"""
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import ParallelRolloutSampler

    env = vs.OneMassOscillatorSim(dt=0.02, max_steps=300)
    if wrap == "delay":
        env = vs.ActDelayWrapper(env, delay=1)
    else:
        env = vs.GaussianObsNoiseWrapper(vs.GaussianActNoiseWrapper(env, noise_std=np.array([1.0])),
                                         noise_std=np.array([0.1, 0.1]))
    runs = []
    for lanes in (64, 16, 64):
        s = ParallelRolloutSampler(env, DummyPolicy(env.spec), 2, min_steps=2000, seed=1, batch_lanes=lanes)
        runs.append(s.sample())
    for other in runs[1:]:
        assert len(other) == len(runs[0])
        for a, b in zip(runs[0], other):
            assert np.array_equal(a.rewards, b.rewards) and np.array_equal(a.observations, b.observations)
    lengths = [len(r) for r in runs[0]]
    assert min(lengths) < 100 < max(lengths) <= 300  # random forcing fails some rollouts early, as without wrappers
