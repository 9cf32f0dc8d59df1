"""Wrapper stacks around the pysim envs: action normalisation / noise / delay, observation
normalisation / noise / partial observation.

CPU part (this file, no GPU):
  * the oracle's sequential restatement (oracle/cpu_ref.WrappedRef) against trajectories the reference's own wrapper
    objects produced (tests/golden/chains.npz, written by oracle/gen_golden.py), noise replayed draw by draw;
  * the package's wrapper classes (`_process_act`, `_process_obs`, spaces) against the same trajectories;
  * `fuse_wrappers`: the fixed pipeline the kernels implement reproduces every stack (applied actions and returned
    observations) when fed the same normal draws.
The GPU part is in tests/test_gpu_wrappers.py."""
import json
import os

import numpy as np
import pytest

import simurlacra_amd as vs
from oracle import cpu_ref
from simurlacra_amd import wrappers as W

KW = {"omo": dict(dt=0.02, max_steps=300), "bob": dict(dt=0.01, max_steps=500), "qq-su": dict(dt=0.004, max_steps=4000),
      "qcp-su": dict(dt=0.002, max_steps=8000), "qbb": dict(dt=0.01, max_steps=500)}


def load_chains(golden_dir):
    g = np.load(os.path.join(golden_dir, "chains.npz"))
    return g, json.loads(str(g["spec"])), int(g["seed"])


def chain_tags(golden_dir=os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")):
    return list(load_chains(golden_dir)[1])


def oracle_stages(env_obj_chain, stages):
    """spec -> WrappedRef stage tuples; label-based ObsNorm overrides are resolved with the package's own spaces"""
    out = []
    ws = [w for w in vs.all_envs(env_obj_chain) if isinstance(w, W.EnvWrapper)]
    for st, w in zip(stages, ws):
        k = st["kind"]
        if k == "act_norm":
            out.append(("act_norm",))
        elif k == "act_delay":
            out.append(("act_delay", st["delay"]))
        elif k == "act_noise":
            out.append(("act_noise", st["mean"], st["std"]))
        elif k == "obs_norm":
            out.append(("obs_norm", w.ov_lb, w.ov_ub))
        elif k == "obs_noise":
            out.append(("obs_noise", st["mean"], st["std"]))
        elif k == "obs_partial":
            out.append(("obs_partial", w.keep_mask))
    return out


def build_chain(name, stages):
    """the package's wrapper objects, innermost first, from the same spec the reference objects were built from"""
    env = vs.ENV_CLASSES[name](**KW[name])
    for st in reversed(stages):
        k = st["kind"]
        if k == "act_norm":
            env = vs.ActNormWrapper(env)
        elif k == "act_delay":
            env = vs.ActDelayWrapper(env, delay=st["delay"])
        elif k == "act_noise":
            env = vs.GaussianActNoiseWrapper(env, noise_mean=np.array(st["mean"]), noise_std=np.array(st["std"]))
        elif k == "obs_norm":
            env = vs.ObsNormWrapper(env, explicit_lb=st.get("lb"), explicit_ub=st.get("ub"))
        elif k == "obs_noise":
            env = vs.GaussianObsNoiseWrapper(env, noise_std=np.array(st["std"]), noise_mean=np.array(st["mean"]))
        elif k == "obs_partial":
            env = vs.ObsPartialWrapper(env, mask=st.get("mask"), idcs=st.get("idcs"))
    return env


def replay(g, tag, name, stages, seed, on_step):
    """drive `on_step(ep, t, state, hidden, act, curr_step) -> dict(state, hidden, ...)` along the golden episodes,
    seeding NumPy's global RNG exactly like the generator did"""
    n_ep = g[f"{tag}__s0"].shape[0]
    for ep in range(n_ep):
        state, hidden = g[f"{tag}__s0"][ep][None], g[f"{tag}__h0"][ep][None]
        np.random.seed(seed + 1000 * ep)
        on_step(ep, -1, state, hidden, None, None)
        for t in range(int(g[f"{tag}__length"][ep])):
            np.random.seed(seed + 1000 * ep + t + 1)
            out = on_step(ep, t, state, hidden, g[f"{tag}__act"][ep, t][None], np.array([t]))
            state, hidden = out["state"], out["hidden"]


@pytest.mark.parametrize("tag", chain_tags())
def test_oracle_wrapped_ref_against_reference_trajectories(golden_dir, tag):
    g, spec, seed = load_chains(golden_dir)
    name, stages = spec[tag]["env"], spec[tag]["stages"]
    ref = cpu_ref.make_ref(name, **KW[name])
    chain = build_chain(name, stages)
    wr = cpu_ref.WrappedRef(ref, oracle_stages(chain, stages), lambda n, w: np.random.randn(w)[None].repeat(n, 0))
    P = g[f"{tag}__params"][None]

    def on_step(ep, t, state, hidden, act, curr):
        if t < 0:
            np.testing.assert_allclose(wr.reset(state)[0], g[f"{tag}__obs0"][ep], rtol=1e-11, atol=1e-13)
            return None
        out = wr.step(state, hidden, act, P, curr)
        np.testing.assert_allclose(out["state"][0], g[f"{tag}__state"][ep, t], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(out["obs"][0], g[f"{tag}__obs"][ep, t], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(out["rew"][0], g[f"{tag}__rew"][ep, t], rtol=1e-9, atol=1e-300)
        assert bool(out["done"][0]) == bool(g[f"{tag}__done"][ep, t])
        if ref.H:
            np.testing.assert_allclose(out["hidden"][0], g[f"{tag}__hidden"][ep, t], rtol=1e-9, atol=1e-11)
        return out

    replay(g, tag, name, stages, seed, on_step)


@pytest.mark.parametrize("tag", chain_tags())
def test_package_wrapper_objects_against_reference_trajectories(golden_dir, tag):
    """the host-side classes process one env's action / observation like the reference's (the inner env step itself
    needs the GPU, so the inner step is the oracle's)"""
    g, spec, seed = load_chains(golden_dir)
    name, stages = spec[tag]["env"], spec[tag]["stages"]
    ref = cpu_ref.make_ref(name, **KW[name])
    chain = build_chain(name, stages)
    ws = [w for w in vs.all_envs(chain) if isinstance(w, W.EnvWrapper)]
    P = g[f"{tag}__params"][None]
    np.testing.assert_allclose(chain.obs_space.bound_lo, g[f"{tag}__obs_lo"], rtol=1e-14)
    np.testing.assert_allclose(chain.obs_space.bound_up, g[f"{tag}__obs_hi"], rtol=1e-14)
    np.testing.assert_allclose(chain.act_space.bound_lo, g[f"{tag}__act_lo"], rtol=1e-14)
    np.testing.assert_allclose(chain.act_space.bound_up, g[f"{tag}__act_hi"], rtol=1e-14)

    def proc_obs(obs):
        for w in reversed(ws):
            if isinstance(w, W.EnvWrapperObs):
                obs = w._process_obs(obs)
        return obs

    def on_step(ep, t, state, hidden, act, curr):
        if t < 0:
            for w in ws:
                if isinstance(w, W.ActDelayWrapper):
                    w._act_queue = [np.zeros(w.act_space.shape)] * w.delay  # what its reset() does after the inner reset
            np.testing.assert_allclose(proc_obs(ref.reset_obs(state)[0]), g[f"{tag}__obs0"][ep], rtol=1e-11, atol=1e-13)
            return None
        a = act[0]
        for w in ws:
            if isinstance(w, W.EnvWrapperAct):
                a = w._process_act(a)
        out = ref.step(state, hidden, a[None], P, curr)
        np.testing.assert_allclose(out["state"][0], g[f"{tag}__state"][ep, t], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(proc_obs(out["obs"][0]), g[f"{tag}__obs"][ep, t], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(out["rew"][0], g[f"{tag}__rew"][ep, t], rtol=1e-9, atol=1e-300)
        return out

    replay(g, tag, name, stages, seed, on_step)


def fused_act(fc, alo, ahi, act, t, ring, z):
    """the kernels' fixed action pipeline (vecsim.hip pipe_act), in NumPy"""
    a = alo + (act + 1) * (ahi - alo) / 2 if fc.act_norm else act.copy()
    nz = (fc.noise_mean + fc.noise_std * z) * (0.5 * (ahi - alo) if fc.noise_normed else 1.0)
    if not fc.noise_after_delay:
        a = a + nz
    if fc.delay > 0:
        slot = t % fc.delay
        prev = ring[slot].copy() if t >= fc.delay else np.zeros_like(a)
        ring[slot] = a
        a = prev
    if fc.noise_after_delay:
        a = a + nz
    return a


@pytest.mark.parametrize("tag", chain_tags())
def test_fused_pipeline_equals_the_sequential_stack(golden_dir, tag):
    """fuse_wrappers() -> (act_norm, delay, noise flags; scale, shift, std, keep) reproduces what the stack of objects
    does: applied action and returned observation, given the same standard-normal draws"""
    g, spec, seed = load_chains(golden_dir)
    name, stages = spec[tag]["env"], spec[tag]["stages"]
    ref = cpu_ref.make_ref(name, **KW[name])
    chain = build_chain(name, stages)
    fc = vs.fuse_wrappers(chain)
    P = g[f"{tag}__params"][None]
    _, _, alo, ahi = ref.bounds(P)
    draws = []
    wr = cpu_ref.WrappedRef(ref, oracle_stages(chain, stages),
                            lambda n, w: draws.append(np.random.randn(w)) or draws[-1][None].repeat(n, 0))
    n_obs_noise = sum(st["kind"] == "obs_noise" for st in stages)
    ring = {}

    def on_step(ep, t, state, hidden, act, curr):
        draws.clear()
        if t < 0:
            wr.reset(state)
            ring[ep] = [np.zeros(ref.A) for _ in range(max(fc.delay, 1))]
            return None
        out = wr.step(state, hidden, act, P, curr)
        z_act = draws[0] if len(draws) > n_obs_noise else np.zeros(ref.A)
        a = fused_act(fc, alo[0], ahi[0], act[0], t, ring[ep], z_act)
        np.testing.assert_allclose(a, out["act_applied"][0], rtol=1e-12, atol=1e-14)
        if n_obs_noise <= 1:  # with two noise stages the fused form is equal in distribution only (variances add)
            z_full = np.zeros(ref.O)
            if n_obs_noise:
                zo = draws[-1]
                vis = np.flatnonzero(visible_at_noise_stage(chain))
                z_full[vis] = zo
            fused = (out["obs_inner"][0] * fc.scale + fc.shift + fc.obs_std * z_full)[fc.keep]
            np.testing.assert_allclose(fused, out["obs"][0], rtol=1e-11, atol=1e-12)
        return out

    replay(g, tag, name, stages, seed, on_step)


def visible_at_noise_stage(chain):
    """mask over the inner env's observation: the entries the (single) obs-noise wrapper of the chain sees"""
    ws = [w for w in vs.all_envs(chain) if isinstance(w, W.EnvWrapper)]
    O = vs.inner_env(chain).obs_space.flat_dim
    idx = np.arange(O)
    for w in reversed(ws):
        if isinstance(w, W.GaussianObsNoiseWrapper):
            break
        if isinstance(w, W.ObsPartialWrapper):
            idx = idx[w.keep_mask]
    m = np.zeros(O, dtype=bool)
    m[idx] = True
    return m


def test_two_noise_stages_add_their_variances(golden_dir):
    g, spec, _ = load_chains(golden_dir)
    chain = build_chain("omo", spec["omo_everything"]["stages"])
    fc = vs.fuse_wrappers(chain)
    # inner noise N(0.05, 0.01) / N(0, 0.1) -> ObsNorm with k = 2 / (ub - lb) = [1, 0.1] -> outer noise N(0, 0.02) / N(0.1, 0.3)
    k = 2.0 / (2 * np.array([1.0, 10.0]))
    np.testing.assert_allclose(fc.scale, k)
    np.testing.assert_allclose(fc.obs_std, np.sqrt((np.array([0.01, 0.1]) * k) ** 2 + np.array([0.02, 0.3]) ** 2))
    np.testing.assert_allclose(fc.shift, np.array([0.05, 0.0]) * k + 0.0 + np.array([0.0, 0.1]))
    assert fc.act_norm and fc.delay == 2 and fc.noise_normed and fc.noise_after_delay  # noise outside the norm, inside the delay
    np.testing.assert_allclose(fc.noise_mean, [0.5])
    np.testing.assert_allclose(fc.noise_std, [2.0])


def test_fuse_wrappers_flags_and_refusals():
    env = vs.ENV_CLASSES["qbb"](**KW["qbb"])
    fc = vs.fuse_wrappers(vs.GaussianActNoiseWrapper(vs.ActNormWrapper(env), noise_std=np.array([0.1, 0.1])))
    assert fc.act_norm and fc.noise_normed and not fc.noise_after_delay and fc.delay == 0
    fc = vs.fuse_wrappers(vs.ActNormWrapper(vs.GaussianActNoiseWrapper(env, noise_std=np.array([0.1, 0.1]))))
    assert fc.act_norm and not fc.noise_normed
    fc = vs.fuse_wrappers(vs.GaussianActNoiseWrapper(vs.ActDelayWrapper(env, delay=4), noise_std=np.array([0.1, 0.1])))
    assert fc.delay == 4 and not fc.noise_after_delay
    fc = vs.fuse_wrappers(vs.DomainRandWrapperLive(vs.ObsPartialWrapper(env, idcs=["x", "y"], keep_selected=True),
                                                   vs.create_default_randomizer(env)))
    assert fc.keep.tolist() == [False, False, True, True, False, False, False, False] and fc.delay == 0
    assert (fc.scale == 1).all() and (fc.shift == 0).all() and (fc.obs_std == 0).all()
    with pytest.raises(NotImplementedError):
        vs.fuse_wrappers(vs.ActDelayWrapper(vs.ActDelayWrapper(env, delay=1), delay=1))
    with pytest.raises(NotImplementedError):
        vs.fuse_wrappers(vs.ActNormWrapper(vs.ActNormWrapper(env)))

    class Odd(W.EnvWrapperObs):
        def _process_obs(self, obs):
            return obs ** 2

    with pytest.raises(NotImplementedError):
        vs.fuse_wrappers(Odd(env))
    # ObsNormWrapper refuses infinite bounds unless they are overridden by label (observation_normalization.py:79-89)
    qq = vs.ENV_CLASSES["qq-su"](**KW["qq-su"])
    qcp = vs.ENV_CLASSES["qcp-su"](**KW["qcp-su"])  # obs bounds [l_rail / 2, 1, 1, inf, inf] (quanser_cartpole.py:92-93)
    with pytest.raises(vs.ValueErr):
        vs.ObsNormWrapper(qcp)
    with pytest.raises(vs.ValueErr):
        vs.ObsNormWrapper(qcp, explicit_ub={"x_dot": 20.0})
    on = vs.ObsNormWrapper(qcp, explicit_lb={"x_dot": -20.0, "theta_dot": -20.0},
                           explicit_ub={"x_dot": 20.0, "theta_dot": 20.0})
    assert np.array_equal(on.obs_space.bound_up, np.ones(5)) and np.array_equal(on.obs_space.bound_lo, -np.ones(5))
    assert np.array_equal(vs.ObsNormWrapper(qq).ov_ub[4:], [20 * np.pi, 20 * np.pi])
    with pytest.raises(vs.ShapeErr):
        vs.GaussianObsNoiseWrapper(qq, noise_std=[0.1, 0.2])
    with pytest.raises(vs.ShapeErr):
        vs.GaussianActNoiseWrapper(qq, noise_std=np.array([0.1, 0.2]))
    with pytest.raises(vs.ShapeErr):
        vs.ObsPartialWrapper(qq, mask=[1, 0])
    d = vs.ActDelayWrapper(qq, delay=2.4)
    assert d.delay == 2
    with pytest.raises(vs.ValueErr):
        d.delay = -1
    dp = d.domain_param
    assert dp["act_delay"] == 2.4
    d.domain_param = dict(act_delay=3)
    assert d.delay == 3
    assert vs.GaussianObsNoiseWrapper(qq, noise_std=np.full(6, 0.1)).domain_param["obs_noise_std"].shape == (6,)
