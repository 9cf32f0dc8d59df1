"""GPU tests of the host-side mirror: the reference's Env surface, DomainRandWrapperLive and ParallelRolloutSampler
running on the HIP kernels.  Shapes follow the reference's own tests: test_environments.py:83-214 (rollout / reset),
environment_wrappers/test_domain_randomization.py:37-56, test_sampling.py:561-700 (determinism across worker counts)."""
import os

import numpy as np
import pytest

from oracle import cpu_ref

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

KW = {"omo": dict(dt=0.02, max_steps=300), "bob": dict(dt=0.01, max_steps=500), "qq-su": dict(dt=0.004, max_steps=4000),
      "qcp-su": dict(dt=0.002, max_steps=8000), "qbb": dict(dt=0.01, max_steps=500),
      "qq-st": dict(dt=0.01, max_steps=500), "qcp-st": dict(dt=0.01, max_steps=300),
      "pend": dict(dt=0.02, max_steps=400, init_state=np.array([0.1, 0.2])), "bob-d": dict(dt=0.01, max_steps=500)}
ENVS = list(KW)


@pytest.fixture(scope="module")
def vs():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import simurlacra_amd

    return simurlacra_amd


@pytest.mark.parametrize("name", ENVS)
def test_single_env_surface_follows_golden_trajectory(vs, golden_dir, name):
    """env.reset(init_state, domain_param) / env.step(act) of ONE env object against the reference trajectory"""
    g = np.load(os.path.join(golden_dir, f"traj_{name.replace('-', '_')}.npz"))
    env = vs.ENV_CLASSES[name](**KW[name])
    names = list(env.get_nominal_domain_param().keys())
    for i in (0, 5):
        dp = dict(zip(names, g["params"][i]))
        obs = env.reset(init_state=g["init"][i].copy(), domain_param=dp)
        np.testing.assert_allclose(obs, g["reset_obs"][i], rtol=1e-5, atol=2e-6)  # qcp: the 4-D state (Q5)
        assert obs.shape == g["reset_obs"][i].shape
        assert env.curr_step == 0
        for t in range(12):
            env.state = g["state"][i, t].copy()  # one-step parity: follow the reference states
            if name in ("qcp-su", "qcp-st", "qbb"):
                env.vec.put(vs._lib.VS_HIDDEN, g["hidden"][i, t][None].astype(np.float32))
            o, r, d, info = env.step(g["act"][i, t].copy())
            assert isinstance(r, float) and isinstance(d, bool) and info == {}
            np.testing.assert_allclose(o, g["obs"][i, t], rtol=1e-5, atol=2e-6)
            assert r == pytest.approx(g["rew"][i, t], rel=5e-5, abs=1e-12)
            assert d == bool(g["done"][i, t])
            assert env.curr_step == t + 1
        np.testing.assert_allclose(env.state, g["state"][i, 12], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("name", ENVS)
def test_rollout_loop_and_init_space(vs, name):
    """test_rollout / test_init_spaces of the reference: hand loop with scaled random actions until done, init samples in
    the init space, observations in the obs space, states in the state space"""
    np.random.seed(0)
    env = vs.ENV_CLASSES[name](**KW[name])
    for _ in range(5):
        obs = env.reset()
        assert env.state_space.contains(env.state)
    obs = env.reset()
    done, n = False, 0
    while not done and n < 50:
        obs, rew, done, _ = env.step(0.1 * env.act_space.sample_uniform())
        assert env.obs_space.contains(obs) or done
        n += 1
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import rollout

    env.max_steps = 20
    ro = rollout(env, DummyPolicy(env.spec), eval=True, seed=0, sub_seed=0, sub_sub_seed=1)
    assert 1 <= len(ro) <= 20 and ro.actions.shape == (len(ro), env.act_space.flat_dim)
    ro2 = rollout(env, DummyPolicy(env.spec), eval=True, seed=0, sub_seed=0, sub_sub_seed=1)
    assert np.array_equal(ro.rewards, ro2.rewards)  # set_seed -> same init state and actions
    # the remaining keyword arguments of the reference's rollout(): per-step wall-clock splits, no_reset, stop_on_done
    ro3 = rollout(env, DummyPolicy(env.spec), eval=True, seed=0, sub_seed=0, sub_sub_seed=1, record_dts=True, no_close=True)
    assert np.array_equal(ro3.rewards, ro.rewards) and len(ro3.dts_policy) == len(ro3.dts_step) == len(ro3.dts_remainder) == len(ro3)
    assert ro3.actions_applied.shape == ro3.actions.shape
    lo, hi = env.act_space.bounds
    assert (ro3.actions_applied >= lo - 1e-5).all() and (ro3.actions_applied <= hi + 1e-5).all()
    state_before = env.state.copy()
    # continue where the last rollout stopped: no reset, zero first observation; raising max_steps keeps state and step count
    ro4 = rollout(env, DummyPolicy(env.spec), no_reset=True, stop_on_done=False, max_steps=len(ro3) + 5)
    assert len(ro4) == 5 and np.array_equal(ro4.states[0], state_before) and env.curr_step == len(ro3) + 5
    assert not ro4.observations[0].any()
    env.dt = 2 * env.dt  # also a plain attribute in the reference
    assert np.array_equal(env.state, ro4.states[-1]) and env.curr_step == len(ro3) + 5
    # test_reset: the same init state gives the same first observation
    s0 = env.init_space.sample_uniform()
    assert np.array_equal(env.reset(init_state=s0), env.reset(init_state=s0))


def test_nan_action_raises(vs):
    env = vs.QQubeSwingUpSim(**KW["qq-su"])
    env.reset()
    with pytest.raises(vs.ValueErr):
        env.step(np.array([np.nan]))


def test_domain_rand_wrapper_live_single_env(vs):
    """params change at every reset unless given explicitly (environment_wrappers/test_domain_randomization.py:37-56)"""
    vs.set_seed(0)
    env = vs.QBallBalancerSim(**KW["qbb"])
    w = vs.DomainRandWrapperLive(env, vs.create_default_randomizer(env))
    w.reset()
    p1 = w.domain_param
    w.reset()
    p2 = w.domain_param
    assert p1 != p2 and p1["gravity_const"] != 9.81
    fixed = dict(p1)
    w.reset(domain_param=fixed)
    assert w.domain_param == fixed
    # the device constants follow: c_max on the device equals the host-side descriptor for the randomised plate
    K = env.vec.get(vs._lib.VS_CONSTS)
    assert K[0, 16] == pytest.approx(w.task.rew_fcn.c_max, rel=3e-6)
    obs, rew, done, _ = w.step(np.array([0.5, -0.5]))
    assert obs.shape == (8,) and 0 < rew <= 1


def test_parallel_sampler_is_deterministic_across_workers_and_batches(vs):
    """identical rollouts for every num_workers / batch size, different rollouts within a batch
    (test_sampling.py:589-650), and a new draw at the next sample() call"""
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import ParallelRolloutSampler

    env = vs.BallOnBeamSim(**KW["bob"])
    pol = DummyPolicy(env.spec)
    res = []
    for nw, bl in ((1, 4096), (4, 4096), (2, 5)):
        s = ParallelRolloutSampler(env, pol, nw, min_rollouts=12, seed=0, batch_lanes=bl)
        res.append(s.sample())
        if nw == 1:
            again = s.sample()
    for other in res[1:]:
        assert len(other) == len(res[0]) == 12
        for a, b in zip(res[0], other):
            assert len(a) == len(b) and np.array_equal(a.rewards, b.rewards) and np.array_equal(a.observations, b.observations)
    lens = {len(r) for r in res[0]}
    rets = {r.undiscounted_return() for r in res[0]}
    assert len(rets) == 12 and len(lens) > 1  # different rollouts within one batch
    assert any(not np.array_equal(a.rewards[:5], b.rewards[:5]) for a, b in zip(res[0], again))  # sample_count moved on
    for r in res[0]:
        assert r.observations.shape == (len(r) + 1, 4) and r.done[-1] and len(r) <= 500


def test_parallel_sampler_matches_oracle_replay(vs):
    """every StepSequence the sampler returns replays exactly through the oracle (obs == state for bob)"""
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import ParallelRolloutSampler

    env = vs.BallOnBeamSim(**KW["bob"])
    w = vs.DomainRandWrapperLive(env, vs.create_default_randomizer(env))
    s = ParallelRolloutSampler(w, DummyPolicy(env.spec), 8, min_rollouts=16, seed=3)
    ros = s.sample()
    ref = cpu_ref.make_ref("bob", **KW["bob"])
    g_all = set()
    for ro in ros:
        P = np.array([[ro.rollout_info["domain_param"][k] for k in ref.param_names]])
        g_all.add(round(float(P[0, 0]), 6))
        for t in range(len(ro)):
            out = ref.step(ro.observations[t][None].astype(np.float64), np.zeros((1, 0)), ro.actions[t][None].astype(np.float64),
                           P, np.array([t]))
            assert ro.rewards[t] == pytest.approx(out["rew"][0], rel=5e-5, abs=1e-12)
            np.testing.assert_allclose(ro.observations[t + 1], out["state"][0], rtol=1e-5, atol=2e-5)
            assert bool(out["done"][0]) == (t == len(ro) - 1) or abs(np.abs(out["state"][0, 0]) - P[0, 4] / 2) < 1e-4
    assert len(g_all) == 16  # DomainRandWrapperLive: every rollout drew its own gravity


def test_parallel_sampler_init_states_domain_params_and_min_steps(vs):
    from simurlacra_amd.policies import DummyPolicy, IdlePolicy
    from simurlacra_amd.sampling import ParallelRolloutSampler

    env = vs.OneMassOscillatorSim(**KW["omo"])
    inits = [np.array([-0.7, 0.0]), np.array([-0.66, 0.05])]
    dps = [dict(mass=1.2), dict(mass=0.8), dict(mass=1.0)]
    s = ParallelRolloutSampler(env, IdlePolicy(env.spec), 3, min_rollouts=6, seed=1)
    ros = s.sample(init_states=inits, domain_params=dps)
    assert len(ros) == 6
    for ro, (ini, dp) in zip(ros, [(i, d) for i in inits for d in dps]):
        np.testing.assert_allclose(ro.observations[0], ini, rtol=1e-6)
        assert ro.rollout_info["domain_param"]["mass"] == pytest.approx(dp["mass"], rel=1e-6)
        assert (ro.actions == 0).all() and len(ro) == 300  # idle policy: the oscillator stays inside, time-out
    # a generic (non-fused) policy path equals the oracle for the idle policy
    ref = cpu_ref.make_ref("omo", **KW["omo"])
    P = ref.nominal_params(1)
    P[0, 0] = 1.2
    st = inits[0][None].astype(np.float32).astype(np.float64)
    for t in range(50):
        out = ref.step(st, np.zeros((1, 0)), np.zeros((1, 1)), P.astype(np.float32).astype(np.float64), np.array([t]))
        st = out["state"]
    np.testing.assert_allclose(ros[0].observations[50], st[0], rtol=1e-4, atol=1e-5)
    # min_steps: rollouts in index order until the step budget is reached (run_collect)
    s2 = ParallelRolloutSampler(env, DummyPolicy(env.spec), 2, min_steps=2000, seed=1, batch_lanes=64)
    ros2 = s2.sample()
    total = sum(len(r) for r in ros2)
    assert total >= 2000 and total - len(ros2[-1]) < 2000
    s3 = ParallelRolloutSampler(env, DummyPolicy(env.spec), 5, min_steps=2000, seed=1, batch_lanes=16)
    ros3 = s3.sample()
    assert len(ros3) == len(ros2) and all(np.array_equal(a.rewards, b.rewards) for a, b in zip(ros2, ros3))


def test_index_offset_makes_shards_equal_the_whole(vs):
    """multi-GPU layout on one GPU: two handles with index offsets 0 and N/2 reproduce one handle of N lanes"""
    L = vs._lib
    n = 4096
    whole = vs.VecSimEnv("qcp-su", n, **KW["qcp-su"])
    parts = [vs.VecSimEnv("qcp-su", n // 2, **KW["qcp-su"]) for _ in range(2)]
    for e, off in [(whole, 0), (parts[0], 0), (parts[1], n // 2)]:
        e.set_index_offset(off)
        e.set_auto_reset(True, seed=5)
        e.reset(seed=9)
        e.step_random(300, seed=11)
    s = whole.get(L.VS_STATE)
    assert np.array_equal(s[: n // 2], parts[0].get(L.VS_STATE)) and np.array_equal(s[n // 2:], parts[1].get(L.VS_STATE))
    c = whole.get(L.VS_EPSTAT_COUNT)
    assert c.sum() > 100 and np.array_equal(c[n // 2:], parts[1].get(L.VS_EPSTAT_COUNT))


def test_vs_step_replays_exactly_from_a_hip_graph(vs):
    """launch-bound policy-in-the-loop stepping captured in a hipGraph (torch.cuda.CUDAGraph): the replayed launches give
    the same states as eager launches, auto-reset included (vs_step takes no host-side counter)"""
    L = vs._lib
    n, k = 8192, 16
    rng = np.random.default_rng(0)
    acts = torch.from_numpy(rng.uniform(-30, 30, (3 * k, n, 1)).astype(np.float32)).cuda()
    eager = vs.VecSimEnv("omo", n, **KW["omo"])
    graphed = vs.VecSimEnv("omo", n, **KW["omo"])
    for e in (eager, graphed):
        e.set_auto_reset(True, seed=4)
        e.reset(seed=2)
    for t in range(3 * k):
        eager.step(acts[t])
    buf = torch.zeros(k, n, 1, device="cuda")
    side = torch.cuda.Stream()
    graphed.use_stream(side.cuda_stream)  # before the capture starts: stream switches synchronise
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for t in range(k):
            graphed.step(buf[t])
    graphed.reset(seed=2)  # the capture itself did not execute anything; start from the same state
    for rep in range(3):
        buf.copy_(acts[rep * k:(rep + 1) * k])
        g.replay()
    torch.cuda.synchronize()
    graphed.use_stream(None)
    for which in (L.VS_STATE, L.VS_STEPCOUNT, L.VS_RETURNS, L.VS_EPSTAT_COUNT, L.VS_EPSTAT_RETSUM):
        assert np.array_equal(eager.get(which), graphed.get(which))
    assert eager.get(L.VS_EPSTAT_COUNT).sum() > 500  # short OMO episodes: auto-resets happened inside the graph


def test_act_norm_fused_in_kernel_matches_reference(vs, golden_dir):
    """VS_FLAG_ACT_NORM (ActNormWrapper fused into the step kernel) against steps of the reference's wrapper, and the
    host-side wrapper on a single env object"""
    L = vs._lib
    g = np.load(os.path.join(golden_dir, "wrappers.npz"))
    for name in ("qq-su", "qbb"):
        tag = name.replace("-", "_")
        n = g[f"{tag}_state"].shape[0]
        env = vs.VecSimEnv(name, n, **KW[name])
        env.set_act_norm(True)
        env.reset(init_state=g[f"{tag}_state"].astype(np.float32))
        if name == "qbb":
            env.put(L.VS_HIDDEN, np.zeros((n, 2), dtype=np.float32))
        env.step(torch.from_numpy(g[f"{tag}_act"].astype(np.float32)).cuda())
        np.testing.assert_allclose(env.get(L.VS_STATE), g[f"{tag}_nstate"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(env.get(L.VS_REW), g[f"{tag}_rew"], rtol=5e-5, atol=1e-12)
        assert np.array_equal(env.get(L.VS_DONE).astype(bool), g[f"{tag}_done"])
        # fused random policy: normalised actions in [-1, 1] are recorded, the env sees the de-normalised ones
        env.reset(seed=1)
        env.step_random(8, seed=2, record=True)
        a = env.traj(8)["act"]
        assert a.min() >= -1 and a.max() <= 1 and a.std() > 0.5
    e = vs.QQubeSwingUpSim(**KW["qq-su"])
    w = vs.ActNormWrapper(e)
    w.reset(init_state=g["qq_su_state"][0].copy())
    obs, rew, done, _ = w.step(g["qq_su_act"][0].copy())
    np.testing.assert_allclose(obs, g["qq_su_obs"][0], rtol=1e-5, atol=1e-5)
    assert rew == pytest.approx(g["qq_su_rew"][0], rel=5e-5, abs=1e-12)


def test_param_buffer_on_device(vs, golden_dir):
    """DomainRandWrapperBuffer on the device: lane i starts at set i mod B and walks the ring at its own resets (cyclic);
    random selection only ever picks members of the buffer; the single-env wrapper follows the reference's order"""
    L = vs._lib
    g = np.load(os.path.join(golden_dir, "wrappers.npz"))
    buf = g["omo_buffer"].astype(np.float32)
    B, n = len(buf), 1000
    env = vs.VecSimEnv("omo", n, **KW["omo"])
    env.set_param_buffer([dict(mass=m, stiffness=k, damping=d) for m, k, d in buf], "cyclic")
    env.set_auto_reset(True, seed=1)
    env.reset(seed=2)
    P = env.get(L.VS_PARAMS)
    assert np.array_equal(P, buf[np.arange(n) % B])
    env.step_random(400, seed=3)
    cnt = env.get(L.VS_EPSTAT_COUNT).astype(np.int64)
    assert cnt.min() >= 1
    assert np.array_equal(env.get(L.VS_PARAMS), buf[(np.arange(n) + cnt) % B])  # one step along the ring per reset
    K = env.get(L.VS_CONSTS)
    np.testing.assert_allclose(K[:, 3], env.get(L.VS_PARAMS)[:, 1], rtol=1e-6)  # act bound = stiffness follows the params
    env.set_param_buffer(buf, "random")
    env.reset(seed=5)
    P = env.get(L.VS_PARAMS)
    idx = np.array([np.where((buf == p).all(axis=1))[0][0] for p in P])
    assert np.bincount(idx, minlength=B).min() > n / B / 2
    env.set_param_buffer(None)
    # single env object: the reference's ring order
    e = vs.OneMassOscillatorSim(**KW["omo"])
    w = vs.DomainRandWrapperBuffer(e, None, selection="cyclic")
    w.buffer = [dict(mass=float(m), stiffness=float(k), damping=float(d)) for m, k, d in g["omo_buffer"]]
    w.ring_idx = 0
    seq = []
    for i in range(12):
        w.reset()
        seq.append([w.domain_param[k] for k in ("mass", "stiffness", "damping")])
    np.testing.assert_allclose(np.array(seq), g["omo_buffer_seq"], rtol=1e-15)


def test_plain_c_host_program_runs_a_rollout(vs, tmp_path):
    """the C-ABI from C on the GPU: tests/capi/capi_demo.c creates 4 096 envs, resets, runs 100 fused steps, copies the
    state back (no Python, no torch in that process)"""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(vs._lib.LIB_PATH)
    exe = str(tmp_path / "capi_demo")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "capi", "capi_demo.c"), "-o", exe, "-L", libdir, "-l:libvecsim.so",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fused steps ok" in out.stdout


@pytest.mark.parametrize("name", ["qbb", "bob"])
def test_wrapper_combination_scenario_of_the_reference(vs, name):
    """Pyrado/tests/environment_wrappers/test_combination.py:64-120 on env objects: a cyclic parameter buffer repeats after
    its length, ObsNorm(ActNorm(env)) rollouts are the normalised plain rollouts, partial observations, action noise and an
    action delay change the rollout the way that test expects (same seeds -> same policy actions)"""
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import rollout

    vs.set_seed(0)
    env = vs.ENV_CLASSES[name](**KW[name])
    env.max_steps = 20
    env_r = vs.DomainRandWrapperBuffer(env, vs.create_default_randomizer(env))
    env_r.fill_buffer(num_domains=3)
    before, after = [], []
    for i in range(4):
        before.append(env_r.domain_param)
        rollout(env_r, DummyPolicy(env_r.spec), eval=True, seed=0)
        after.append(env_r.domain_param)
        assert after[i] != before[i]
    assert after[0] == after[3]

    env.domain_param = type(env).get_nominal_domain_param()
    env_n = vs.ActNormWrapper(env)
    env_n = vs.ObsNormWrapper(env_n)  # all observation bounds of these two envs are finite
    alb, aub = env_n.act_space.bounds
    olb, oub = env_n.obs_space.bounds
    assert all(alb == -1) and all(aub == 1) and all(olb == -1) and all(oub == 1)
    plain = vs.ActNormWrapper(env)
    ro_p = rollout(plain, DummyPolicy(plain.spec), eval=True, seed=0)
    ro_n = rollout(env_n, DummyPolicy(env_n.spec), eval=True, seed=0)
    assert np.allclose(env_n._process_obs(ro_p.observations), ro_n.observations, atol=1e-6)

    labels = env.obs_space.labels
    env_np = vs.ObsPartialWrapper(env_n, idcs=[labels[2], labels[3]])
    ro_np = rollout(env_np, DummyPolicy(env_np.spec), eval=True, seed=0)
    assert ro_np.observations.shape[1] == len(labels) - 2
    keep = [j for j in range(len(labels)) if j not in (2, 3)]
    assert np.allclose(ro_np.observations, ro_n.observations[:, keep], atol=1e-6)

    env_npa = vs.GaussianActNoiseWrapper(env_np, noise_mean=0.5 * np.ones(env_np.act_space.shape),
                                         noise_std=0.1 * np.ones(env_np.act_space.shape))
    ro_npa = rollout(env_npa, DummyPolicy(env_npa.spec), eval=True, seed=0)
    n = min(len(ro_np), len(ro_npa)) + 1
    assert not np.allclose(ro_np.observations[:n], ro_npa.observations[:n])  # the action noise changed the rollout

    env_npd = vs.ActDelayWrapper(env_np, delay=3)
    ro_npd = rollout(env_npd, DummyPolicy(env_npd.spec), eval=True, seed=0)
    n = min(len(ro_np), len(ro_npd))
    assert np.allclose(ro_np.actions[:n], ro_npd.actions[:n])  # same policy actions ...
    assert not np.allclose(ro_np.observations[:n + 1], ro_npd.observations[:n + 1])  # ... applied three steps later
    assert type(vs.inner_env(env_npd)) is type(env) and vs.typed_env(env_npd, vs.ObsPartialWrapper) is not None


def test_cvar_sampler_over_gpu_rollouts(vs):
    """CVaRSampler around the batched sampler: 1 / epsilon times the rollouts are drawn as lanes, the worst quantile is kept"""
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import CVaRSampler, ParallelRolloutSampler

    env = vs.BallOnBeamSim(dt=0.01, max_steps=100)
    inner = ParallelRolloutSampler(env, DummyPolicy(env.spec), 4, min_rollouts=1, seed=2)
    cs = CVaRSampler(inner, epsilon=0.25, min_rollouts=64)
    assert inner.min_rollouts == 256
    ros = cs.sample()
    assert len(ros) == 64
    rets = [r.undiscounted_return() for r in ros]
    assert rets == sorted(rets) and np.mean(rets) < cs.full_stats["full avg return"]


def test_package_before_torch_in_a_fresh_process(vs):
    """`import simurlacra_amd` and a first rollout BEFORE torch is imported by the user: the package loads torch's bundled
    HIP runtime first itself (simurlacra_amd/_lib.py), so both orders work (with libvecsim's runtime loaded first torch
    used to report 'No HIP GPUs are available')"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, simurlacra_amd as vs\n"
            "env = vs.QQubeSwingUpSim(dt=0.004, max_steps=50)\n"
            "obs = env.reset(); obs, rew, done, _ = env.step(np.array([0.5]))\n"
            "import torch\n"
            "print('ok', obs.shape, float(torch.ones(4, device='cuda').sum()))\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok (6,) 4.0" in out.stdout, out.stdout + out.stderr


# ---------------------------------------------------------------------------------------------------------------------------
# the batched rollouts carry everything rollout() keeps per step (P/sampling/rollout.py:237-258, 305-325)
class _DoubleObs:
    """a stateless policy whose arithmetic is exact on the CPU and on the GPU alike: act = 2 * obs[..., :A]"""

    def __init__(self, A):
        self.A = A

    def __call__(self, obs):
        return 2.0 * obs[..., : self.A]


class _Replay:
    """feeds recorded actions, one time step per call: actions [T, n, A]"""

    def __init__(self, actions):
        self.actions, self.t = actions, 0

    def reset(self):
        self.t = 0

    def to(self, dev):
        self.actions = self.actions.to(dev)
        return self

    def __call__(self, obs):
        a = self.actions[min(self.t, len(self.actions) - 1)]
        self.t += 1
        return a


@pytest.mark.parametrize("name", ["omo", "bob", "qq-su", "qcp-su", "qbb", "pend"])
def test_fused_sampler_rollouts_carry_states_applied_actions_and_hidden(vs, name):
    """DummyPolicy (fused on the device, record mode 2): every returned StepSequence has states [T + 1, S], actions_applied
    [T, A] and, for the cartpole, th_ddot [T + 1]; each step replays through the fp64 oracle from the recorded state"""
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import ParallelRolloutSampler

    env = vs.ENV_CLASSES[name](**dict(KW[name], max_steps=60))
    ros = ParallelRolloutSampler(env, DummyPolicy(env.spec), 4, min_rollouts=40, seed=7).sample()
    ref = cpu_ref.make_ref(name, **dict(KW[name], max_steps=60))
    P = ref.nominal_params(1).astype(np.float32).astype(np.float64)
    _, _, alo, ahi = ref.bounds(P)
    assert len(ros) == 40
    for ro in ros:
        T = len(ro)
        assert ro.states.shape == (T + 1, ref.S) and ro.actions_applied.shape == (T, ref.A)
        np.testing.assert_array_equal(ro.states[0], ro.init_state)
        np.testing.assert_array_equal(ro.actions_applied, np.clip(ro.actions, alo[0].astype(np.float32), ahi[0].astype(np.float32)))
        np.testing.assert_allclose(ro.observations, ref.observe(ro.states.astype(np.float64)), rtol=1e-6, atol=5e-7)
        if name == "qcp-su":
            assert ro.th_ddot.shape == (T + 1,) and ro.th_ddot[0] == 0.0
        else:
            assert not hasattr(ro, "th_ddot")
        if name == "qbb":
            continue  # its hidden plate angles are not a StepSequence field: nothing to restart the oracle from
        hid = ro.th_ddot[:-1, None].astype(np.float64) if name == "qcp-su" else np.zeros((T, 0))
        out = ref.step(ro.states[:-1].astype(np.float64), hid, ro.actions.astype(np.float64), np.repeat(P, T, axis=0), np.arange(T))
        np.testing.assert_allclose(ro.rewards, out["rew"], rtol=3e-5, atol=1e-9)
        tol = 1e-5 * np.abs(out["state"]) + 2e-6 * np.maximum(1.0, np.abs(ref.bounds(np.repeat(P, T, axis=0))[1]))
        assert (np.abs(ro.states[1:] - out["state"]) <= tol).all()
        if name == "qcp-su":
            np.testing.assert_allclose(ro.th_ddot[1:], out["hidden"][:, 0], rtol=2e-4, atol=2e-3)
        assert np.array_equal(out["done"][:-1], np.zeros(T - 1, dtype=bool)) and (bool(out["done"][-1]) or T == 60)
    lean = ParallelRolloutSampler(env, DummyPolicy(env.spec), 4, min_rollouts=8, seed=7, full_records=False).sample()
    assert lean[0].states is None and not hasattr(lean[0], "actions_applied")  # the lean record mode stays available
    for a, b in zip(lean, ros):
        np.testing.assert_array_equal(a.observations, b.observations)
        np.testing.assert_array_equal(a.rewards, b.rewards)


@pytest.mark.parametrize("name", ["omo", "bob", "qq-su", "qcp-su", "qbb"])
def test_batched_sampler_equals_the_one_env_rollout_field_by_field(vs, name):
    """policy in the loop: rollout i of ParallelRolloutSampler.sample(init_states) == rollout(env, policy,
    reset_kwargs=dict(init_state=...)) of ONE env object, every field of the StepSequence bit for bit (same step kernel,
    a policy whose arithmetic is exact on both sides); finished lanes of the batch are frozen, not stepped on"""
    from simurlacra_amd.sampling import ParallelRolloutSampler, rollout

    kw = dict(KW[name], max_steps=40)
    env = vs.ENV_CLASSES[name](**kw)
    ref = cpu_ref.make_ref(name, **kw)
    rng = np.random.default_rng(3)
    lo, hi = ref.init_bounds(ref.nominal_params(6)) if name != "bob" else ref.init_bounds(ref.nominal_params(6), 1)
    inits = [x.astype(np.float32).astype(np.float64) for x in rng.uniform(lo, hi)]
    if name != "qbb":  # (its init space is polar: init states of init-space shape only)
        # three rollouts that leave the state space after a few steps: full-state init next to a bound, moving outwards
        p_idx, v_idx, vel = {"omo": (0, 1, 5.0), "bob": (0, 2, 2.0), "qq-su": (0, 2, 10.0), "qcp-su": (0, 2, 0.7)}[name]
        shi = ref.bounds(ref.nominal_params(1))[1][0]
        for frac in (0.9, 0.97, 0.995):
            st = np.zeros(ref.S)
            st[p_idx], st[v_idx] = frac * shi[p_idx], vel
            inits.append(st.astype(np.float32).astype(np.float64))
    pol = _DoubleObs(ref.A)
    batch = ParallelRolloutSampler(env, pol, 2, min_rollouts=len(inits), seed=1).sample(init_states=inits)
    assert len(batch) == len(inits)
    lengths = set()
    for ini, ro_b in zip(inits, batch):
        one = vs.ENV_CLASSES[name](**kw)
        ro_1 = rollout(one, pol, reset_kwargs=dict(init_state=ini.copy()))
        lengths.add(len(ro_1))
        assert len(ro_b) == len(ro_1)
        obs_1 = np.stack([np.asarray(o, dtype=np.float32) for o in ro_1.observations[1:]])
        np.testing.assert_array_equal(ro_b.observations[1:], obs_1)  # (the cartpole's reset() returns its state: quirk Q5)
        for field in ("actions", "actions_applied", "states"):
            np.testing.assert_array_equal(getattr(ro_b, field), np.asarray(getattr(ro_1, field), dtype=np.float32), err_msg=field)
        np.testing.assert_array_equal(ro_b.rewards, np.asarray(ro_1.rewards, dtype=np.float64))
        assert ro_b.done[-1] == ro_1.done[-1]
    assert len(lengths) > 2 or name == "qbb"  # rollouts of different lengths shared the batch: finished lanes were frozen


@pytest.mark.parametrize("name", ["bob", "qq-su", "qcp-su"])
def test_batched_sampler_follows_the_golden_trajectories(vs, golden_dir, name):
    """the reference's own trajectories through the batched sampler: init states and domain parameters of
    tests/golden/traj_*.npz (one rollout per golden trajectory), the reference's actions replayed by the policy ->
    StepSequence.states / rewards / th_ddot against the reference's over the first steps (closed loop: fp32 rounding grows
    along a trajectory, hence the short horizon and the looser bound than the one-step tests)"""
    from simurlacra_amd.sampling import ParallelRolloutSampler

    g = np.load(os.path.join(golden_dir, f"traj_{name.replace('-', '_')}.npz"))
    n, horizon = g["init"].shape[0], 12
    env = vs.ENV_CLASSES[name](**dict(KW[name], max_steps=horizon))
    names = list(env.get_nominal_domain_param().keys())
    long_ones = 0
    for i in range(n):  # one domain-parameter set per sample() call: init_states x domain_params is a Cartesian product
        acts = torch.from_numpy(np.ascontiguousarray(g["act"][i:i + 1, :horizon].transpose(1, 0, 2)).astype(np.float32))
        s = ParallelRolloutSampler(env, _Replay(acts), 1, min_rollouts=1, seed=0)
        (ro,) = s.sample(init_states=[g["init"][i].copy()], domain_params=[dict(zip(names, g["params"][i]))])
        dn = np.flatnonzero(g["done"][i, : int(g["length"][i])])  # (the golden trajectories run a few steps past done)
        t_ref = min(int(dn[0]) + 1 if len(dn) else int(g["length"][i]), horizon)
        assert len(ro) == t_ref  # the rollout ends where the reference's does
        T = t_ref
        long_ones += T >= 5
        shi = np.maximum(1.0, np.abs(g["state"][i, : T + 1]).max(axis=0))  # (rows beyond the episode's end are padding)
        err = np.abs(ro.states[: T + 1] - g["state"][i, : T + 1])
        assert (err <= 2e-5 * np.abs(g["state"][i, : T + 1]) + 1e-5 * shi).all(), (i, float(err.max()))
        np.testing.assert_allclose(ro.rewards[:T], g["rew"][i, :T], rtol=1e-4, atol=1e-10)
        np.testing.assert_array_equal(ro.actions[:T], g["act"][i, :T].astype(np.float32))
        if name == "qcp-su":
            np.testing.assert_allclose(ro.th_ddot[: T + 1], g["hidden"][i, : T + 1, 0], rtol=5e-4, atol=5e-3)
    assert long_ones >= n // 2


@pytest.mark.parametrize("name,policy_kind", [("qq-su", "dummy"), ("qcp-su", "dummy"), ("bob", "fnn"), ("qbb", "fnn")])
def test_sample_packed_holds_what_sample_returns(vs, name, policy_kind):
    """sample_packed(): the same rollouts as sample() (same work list, seeds, order), as packed device tensors -- rollout j's
    slices equal the j-th StepSequence field for field, bit for bit; two batches of lanes when the call exceeds batch_lanes"""
    import torch

    from simurlacra_amd.policies import DummyPolicy, FNNPolicy
    from simurlacra_amd.sampling import PackedRollouts, ParallelRolloutSampler

    env = vs.ENV_CLASSES[name](**dict(KW[name], max_steps=50))
    torch.manual_seed(3)
    pol = DummyPolicy(env.spec) if policy_kind == "dummy" else FNNPolicy(env.spec, [32, 32], torch.tanh, featurize=False)
    ros = ParallelRolloutSampler(env, pol, 2, min_rollouts=70, seed=11, batch_lanes=48).sample()
    packs = ParallelRolloutSampler(env, pol, 2, min_rollouts=70, seed=11, batch_lanes=48).sample_packed()
    assert [len(p) for p in packs] == [48, 22] and all(isinstance(p, PackedRollouts) for p in packs)
    j0 = 0
    for p in packs:
        assert p.first_index == j0 and p.observations.is_cuda and p.offsets.shape == (len(p) + 1,)
        assert p.total_steps == int(p.lengths.sum()) == int(p.offsets[-1])
        assert p.rows.shape[0] == p.total_steps + len(p) and p.actions.shape[0] == p.rows.shape[0]  # fields: views of one matrix
        dense_act = p.actions[p.step_rows()]  # the reference-style concatenation (StepSequence.concat drops the final entries)
        assert dense_act.shape[0] == p.total_steps
        ret = p.undiscounted_returns().cpu().numpy()
        for j in range(len(p)):
            ro = ros[j0 + j]
            st, ob = p.step_slice(j), p.obs_slice(j)
            assert len(ro) == int(p.lengths[j]) and bool(p.done_last[j]) == bool(ro.done[-1])
            np.testing.assert_array_equal(p.observations[ob].cpu().numpy(), ro.observations)
            np.testing.assert_array_equal(p.actions[st].cpu().numpy(), ro.actions)
            np.testing.assert_array_equal(p.rewards[st].cpu().numpy().astype(np.float64), ro.rewards)
            np.testing.assert_array_equal(p.states[ob].cpu().numpy(), ro.states)
            np.testing.assert_array_equal(p.actions_applied[st].cpu().numpy(), ro.actions_applied)
            np.testing.assert_array_equal(p.init_states[j].cpu().numpy(), ro.init_state)
            if name == "qcp-su":
                np.testing.assert_array_equal(p.th_ddot[ob].cpu().numpy(), ro.th_ddot)
            else:
                assert p.th_ddot is None
            np.testing.assert_allclose(ret[j], ro.undiscounted_return(), rtol=2e-5, atol=1e-6)
            want = np.array([ro.rollout_info["domain_param"][k] for k in p.param_names], dtype=np.float32)
            np.testing.assert_array_equal(p.domain_params[j].cpu().numpy(), want)
        j0 += len(p)
    with pytest.raises(vs.ValueErr):
        ParallelRolloutSampler(env, pol, 2, min_steps=100, seed=11).sample_packed()


def test_sampler_owned_arrays_and_close(vs):
    """owned_arrays=True: the rollouts' arrays are pageable copies the caller owns (the default hands out views of one pinned
    block per call); same values either way; close() releases the device handle and the conversion threads and is idempotent"""
    from simurlacra_amd.policies import DummyPolicy
    from simurlacra_amd.sampling import ParallelRolloutSampler

    env = vs.BallOnBeamSim(**dict(KW["bob"], max_steps=40))
    a = ParallelRolloutSampler(env, DummyPolicy(env.spec), 1, min_rollouts=33, seed=5)
    b = ParallelRolloutSampler(env, DummyPolicy(env.spec), 1, min_rollouts=33, seed=5, owned_arrays=True)
    ra, rb = a.sample(), b.sample()
    assert len(ra) == len(rb) == 33
    for x, y in zip(ra, rb):
        for f in ("observations", "actions", "rewards", "states", "actions_applied"):
            np.testing.assert_array_equal(getattr(x, f), getattr(y, f))
    base_a = ra[0].observations.base if ra[0].observations.base is not None else ra[0].observations
    assert any(r.observations.base is not None and np.shares_memory(r.observations, ra[0].observations.base) for r in ra[1:])  # views of ONE block
    assert not np.shares_memory(rb[0].observations, rb[1].observations) or rb[0].observations.base is rb[1].observations.base
    a.close(), b.close()
    a.close()  # idempotent
    assert len(a.sample()) == 33  # a closed sampler builds its handle again


@pytest.mark.parametrize("name", ["omo", "bob", "qq-su", "qcp-su", "qbb", "pend"])
@pytest.mark.parametrize("mode", [1, 2])
def test_pack_traj_kernel_equals_the_index_gather(vs, name, mode):
    """vs_pack_traj (time-major record planes -> rollout-major packed arrays, one kernel) against torch index arithmetic on
    traj_tensors(): ragged batch (n = 300 of ld = 512), lengths from the first done (or the launch's end) and shorter, both record modes,
    rollouts longer than one 64-step chunk of the kernel; final entries = VS_OBS / VS_STATE / VS_HIDDEN of the frozen lanes"""
    import torch

    L = vs._lib
    n, T = 300, 150
    e = vs.VecSimEnv(name, n, **dict(KW[name], max_steps=110))
    e.set_auto_reset(False)
    e.reset(seed=5)
    e.set_record_mode(mode)
    e.set_traj_capacity(T)
    for t0, k in ((0, 70), (70, 80)):
        e.set_traj_offset(t0)
        e.step_random(k, seed=9, record=True)
    e.set_traj_offset(0)
    e.sync()
    tt = e.traj_tensors(T, n)
    done = tt["done"].bool()  # [T, n]
    ar = torch.arange(n, device=done.device)
    first = torch.where(done.any(0), done.to(torch.uint8).argmax(0), torch.full_like(ar, T - 1))
    for Tq in (T, 100, 64, 33):  # vs_rollout_lengths: first set done bit of rows 0 .. Tq - 1 (rows beyond Tq masked out of the last word)
        lens, dl = e.rollout_lengths(n, Tq)
        dq = done[:Tq]
        want_first = torch.where(dq.any(0), dq.to(torch.uint8).argmax(0), torch.full_like(ar, Tq - 1))
        assert torch.equal(lens, want_first + 1) and torch.equal(dl, dq.any(0)), Tq
    length = torch.minimum(first + 1, 1 + (ar * 37) % T)  # (cut further, lane by lane: the kernel moves what the lengths say)
    start = torch.cumsum(length, 0) - length
    assert int(length.min()) == 1 and int(length.max()) > 64  # one-step and multi-chunk rollouts in one batch
    pk = e.pack_traj(n, T, length, start)
    torch.cuda.synchronize()
    total = int(length.sum())
    lane = torch.repeat_interleave(ar, length)
    t_idx = torch.arange(total, device=lane.device) - start[lane]
    fin = {"obs": e.tensor(L.VS_OBS)[:, :n].t(), "state": e.tensor(L.VS_STATE)[:, :n].t(),
           "hidden": e.tensor(L.VS_HIDDEN)[:, :n].t() if e.dims["H"] else None}
    # one matrix rows[total + n, F]: step t of rollout j in row start[j] + j + t, the entry behind the last step in row
    # start[j] + j + length[j] (final observation / state / hidden state, per-step fields 0); the fields are views of it
    F = e.traj_layout()[0]
    assert pk["rows"].shape == (total + n, F) and pk["rows"].is_contiguous()
    step_row = torch.arange(total, device=lane.device) + lane
    fin_row = start + length + ar
    for key in ("act", "rew") + (("act_app",) if mode == 2 else ()):
        assert pk[key].shape[0] == total + n and pk[key].data_ptr() >= pk["rows"].data_ptr()  # (a view)
        assert torch.equal(pk[key][step_row], tt[key][t_idx, lane]), key
        assert float(pk[key][fin_row].abs().max()) == 0.0, key
    for key in ("obs",) + (("state", "hidden") if mode == 2 else ()):
        if key == "hidden" and not e.dims["H"]:
            assert pk[key] is None
            continue
        want = torch.empty(total + n, tt[key].shape[2], device=lane.device)
        want[step_row] = tt[key][t_idx, lane]
        want[fin_row] = fin[key]
        assert torch.equal(pk[key], want), key
    if mode == 1:
        assert "state" not in pk
    e.close()


@pytest.mark.parametrize("name,mode", [("qq-su", 2), ("qq-su", 1), ("qbb", 2), ("omo", 1)])
@pytest.mark.parametrize("misalign", [0, 1, 7])
def test_pack_traj_whole_lines_across_tiles_and_segments(vs, name, mode, misalign):
    """vs_pack_traj writes whole 128-byte lines and carries the piece behind a tile's last line boundary into the next tile:
    rollouts of 1 .. 700 steps (through the kernel's 256-step segments: a segment's last tile flushes, the next segment starts on
    whatever byte it starts), lengths at and around the tile and segment sizes, n not a multiple of 64, and the destination
    matrix at 4-byte alignment only (the C-ABI takes any float*): every row against the index gather, nothing written outside."""
    import ctypes as C

    import torch

    n, T = 130, 700
    e = vs.VecSimEnv(name, n, **dict(KW[name], max_steps=10 ** 6))
    e.set_auto_reset(False)
    e.reset(seed=2)
    e.set_record_mode(mode)
    e.set_traj_capacity(T)
    e.step_random(T, seed=4, record=True)
    e.sync()
    tt = e.traj_tensors(T, n)
    ar = torch.arange(n, device="cuda")
    length = 1 + (ar * 131) % T
    for k, v in enumerate((1, 2, 15, 16, 17, 255, 256, 257, 272, 511, 512, 513, 700, 699, 32, 48)):
        length[k * 8] = v
    start = torch.cumsum(length, 0) - length
    total = int(length.sum())
    F = e.traj_layout()[0]
    guard = 64
    sentinel = -1.2345e30
    buf = torch.full((guard + misalign + (total + n) * F + guard,), sentinel, device="cuda")
    rows = buf[guard + misalign:guard + misalign + (total + n) * F]
    assert rows.data_ptr() % 16 == (4 * misalign) % 16
    e._check(e._lib.vs_pack_traj(e._h, n, T, C.c_void_p(length.data_ptr()), C.c_void_p(start.data_ptr()), C.c_void_p(rows.data_ptr())),
             "vs_pack_traj")
    torch.cuda.synchronize()
    assert bool((buf[:guard + misalign] == sentinel).all()) and bool((buf[guard + misalign + (total + n) * F:] == sentinel).all())
    rows = rows.view(total + n, F)
    lane = torch.repeat_interleave(ar, length)
    t_idx = torch.arange(total, device="cuda") - start[lane]
    step_row = torch.arange(total, device="cuda") + lane
    assert torch.equal(rows[step_row].view(torch.int32), tt["rec"][t_idx, lane].contiguous().view(torch.int32))  # (bit patterns)
    fin = rows[start + length + ar]
    O, A = e.dims["O"], e.dims["A"]
    assert torch.equal(fin[:, :O].contiguous().view(torch.int32), e.tensor(vs._lib.VS_OBS)[:, :n].t().contiguous().view(torch.int32))
    assert float(fin[:, O:O + A + 1].abs().max()) == 0.0  # (action and reward of the entry behind the last step)
    e.close()


@pytest.mark.parametrize("name", ["qq-su", "bob"])
def test_graph_policy_path_equals_the_eager_one(vs, name):
    """graph_policy=True: a torch policy stepped through a replayed hipGraph of 32 (observation, policy, recording step)
    iterations -- a deterministic policy gives the same rollouts as the eager loop, field for field, bit for bit (warm-up steps
    undone by the same reset; the record row comes from the device-side counter); max_steps not a multiple of 32"""
    import torch

    from simurlacra_amd.policies import FNNPolicy
    from simurlacra_amd.sampling import ParallelRolloutSampler

    env = vs.ENV_CLASSES[name](**dict(KW[name], max_steps=75))
    torch.manual_seed(5)
    pol = FNNPolicy(env.spec, [16, 16], torch.tanh, featurize=False)
    a = ParallelRolloutSampler(env, pol, 2, min_rollouts=50, seed=3, fuse_policy=False).sample()
    b = ParallelRolloutSampler(env, pol, 2, min_rollouts=50, seed=3, fuse_policy=False, graph_policy=True).sample()
    c = ParallelRolloutSampler(env, pol, 2, min_rollouts=50, seed=3, fuse_policy=False, graph_policy=True).sample()  # (a second capture)
    assert len(a) == len(b) == len(c) == 50 and (name == "qq-su" or len({len(r) for r in a}) > 1)  # (bob: rollouts of different lengths)
    for ra, rb, rc in zip(a, b, c):
        for other in (rb, rc):
            assert len(ra) == len(other)
            np.testing.assert_array_equal(ra.observations, other.observations)
            np.testing.assert_array_equal(ra.actions, other.actions)
            np.testing.assert_array_equal(ra.rewards, other.rewards)
            np.testing.assert_array_equal(ra.states, other.states)
            np.testing.assert_array_equal(ra.init_state, other.init_state)
