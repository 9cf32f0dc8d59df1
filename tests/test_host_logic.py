"""CPU tests of the host-side mirror of the reference interface: spaces, seeding, domain-randomisation objects,
env descriptions (spaces / task / nominal params vs what the reference produced), wrappers, sampler work lists.
Modelled on Pyrado/tests/test_spaces.py, test_set_seed.py, test_domain_randomization.py, test_sampling.py."""
import json
import os
import pickle

import numpy as np
import pytest

import simurlacra_amd as vs
from simurlacra_amd.sampling import ParallelRolloutSampler, StepSequence

KW = {"omo": dict(dt=0.02, max_steps=300), "bob": dict(dt=0.01, max_steps=500), "qq-su": dict(dt=0.004, max_steps=4000),
      "qcp-su": dict(dt=0.002, max_steps=8000), "qbb": dict(dt=0.01, max_steps=500),
      "qq-st": dict(dt=0.01, max_steps=500), "qcp-st": dict(dt=0.01, max_steps=300),
      "pend": dict(dt=0.02, max_steps=400, init_state=np.array([0.1, 0.2])), "bob-d": dict(dt=0.01, max_steps=500)}
ENVS = list(KW)


def make(name, **extra):
    return vs.ENV_CLASSES[name](**KW[name], **extra)


# ---------------------------------------------------------------------------------------------------- spaces
def test_box_space_contains_project_sample():
    bs = vs.BoxSpace([-1, -2], [1, 2], labels=["a", "b"])
    assert bs.contains(np.array([1.0, -2.0]))  # inclusive bounds (Q9)
    assert not bs.contains(np.array([1.0000001, 0.0]))
    with pytest.raises(vs.ValueErr):
        bs.contains(np.array([np.nan, 0.0]))
    with pytest.raises(vs.ShapeErr):
        bs.contains(np.zeros(3))
    x = np.array([0.5, 0.5])
    assert bs.project_to(x) is x  # same object when inside (box.py:180-184)
    y = bs.project_to(np.array([3.0, -5.0]))
    assert np.array_equal(y, [1.0, -2.0])
    np.random.seed(0)
    for _ in range(100):
        assert bs.contains(bs.sample_uniform())
    assert np.array_equal(bs.bound_abs_up, [1, 2]) and bs.flat_dim == 2
    assert np.array_equal(vs.BoxSpace(-4.5, 4.5, shape=(1,)).bound_up, [4.5])


def test_polar_and_compound_spaces():
    np.random.seed(1)
    ps = vs.Polar2DPosVelSpace(np.array([0.1, -np.pi, -0.02, -0.02]), np.array([0.11, np.pi, 0.02, 0.02]))
    for _ in range(100):
        s = ps.sample_uniform()
        assert 0.1 - 1e-12 <= np.hypot(s[0], s[1]) <= 0.11 + 1e-12 and ps.contains(s)
    cs = vs.CompoundSpace([vs.BoxSpace([-2.0], [-1.0]), vs.BoxSpace([1.0], [2.0])])
    vals = np.array([cs.sample_uniform()[0] for _ in range(200)])
    assert ((np.abs(vals) >= 1) & (np.abs(vals) <= 2)).all() and (vals < 0).any() and (vals > 0).any()
    assert cs.contains(np.array([1.5])) and not cs.contains(np.array([0.0]))


# ---------------------------------------------------------------------------------------------------- seeding
def test_set_seed_kat(golden_dir):
    for b, s, ss, exp in json.load(open(os.path.join(golden_dir, "set_seed.json"))):
        assert vs.derive_seed(b, s, ss) == exp
    assert vs.set_seed(0, 1, 1) == 3918913762 and vs.get_base_seed() == 0  # Pyrado/tests/test_set_seed.py:44
    a = np.random.rand()
    vs.set_seed(0, 1, 1)
    assert np.random.rand() == a
    assert vs.set_seed(None) is None and vs.set_seed(0.5) is None


# ---------------------------------------------------------------------------------------------------- randomizers
@pytest.mark.parametrize("name", ENVS)
def test_default_randomizer_tables_match_reference(golden_dir, name):
    tab = json.load(open(os.path.join(golden_dir, "randomizers.json")))[name]
    env = make(name)
    assert {k: float(v) for k, v in type(env).get_nominal_domain_param().items()} == tab["nominal"]
    rz = vs.create_default_randomizer(env)
    assert len(rz.domain_params) == len(tab["randomizer"])
    for dp, row in zip(rz.domain_params, tab["randomizer"]):
        assert dp.name == row["name"] and type(dp).__name__ == row["kind"]
        spread = dp.std if row["kind"] == "NormalDomainParam" else dp.halfspan
        assert float(dp.mean) == row["mean"] and float(spread) == pytest.approx(row["spread"], rel=1e-15)
        assert float(dp.clip_lo) == row["clip_lo"] and float(dp.clip_up) == row["clip_up"]
    specs = rz.device_specs()
    assert [s[0] for s in specs] == [r["name"] for r in tab["randomizer"]]


def test_domain_param_classes_reproduce_reference_draws(golden_dir):
    """Bernoulli / MultivariateNormal / roundint / clipping: same torch calls as the reference, same values under the
    same torch.manual_seed (tests/golden/domain_params.json, drawn by the reference classes)"""
    import torch

    from simurlacra_amd import domain_randomization as dr

    cases = json.load(open(os.path.join(golden_dir, "domain_params.json")))
    assert {c["cls"] for c in cases} >= {"BernoulliDomainParam", "MultivariateNormalDomainParam", "NormalDomainParam"}
    for c in cases:
        torch.manual_seed(c["seed"])
        dp = getattr(dr, c["cls"])(**c["kwargs"])
        smp = dp.sample(len(c["samples"]))
        assert str(smp[0].dtype) == c["dtype"] and dp.get_field_names() == c["fields"]
        got = [np.asarray(t.detach().numpy(), dtype=np.float64).reshape(-1).tolist() for t in smp]
        assert got == c["samples"]
        assert np.asarray(dp.mean, dtype=np.float64).reshape(-1).tolist() == c["mean"]
    # what the device gets for them (include/vecsim.h vs_dp_spec)
    rz = vs.DomainRandomizer(*[getattr(dr, c["cls"])(**c["kwargs"]) for c in cases])
    specs = rz.device_specs()
    assert specs[0] == ("mass", "bernoulli", 1.0, 3.0, -np.inf, 2.5, 0.3, False)
    assert specs[1][:2] == ("stiffness", "bernoulli") and specs[1][6:] == (0.7, True)
    assert specs[2][:2] == ("damping", "normal") and specs[2][3] == pytest.approx(0.2) and specs[2][4] == 0.3
    assert specs[3] == ("stiffness", "normal", 30.0, 4.0, -np.inf, np.inf, 0.0, True)
    with pytest.raises(NotImplementedError):
        vs.DomainRandomizer(vs.MultivariateNormalDomainParam(name="mass", mean=[1.0, 2.0], cov=[[1.0, 0.0], [0.0, 1.0]])).device_specs()
    with pytest.raises(vs.ShapeErr):
        vs.MultivariateNormalDomainParam(name="mass", mean=[1.0], cov=[1.0])
    with pytest.raises(RuntimeError):
        vs.DomainParam(name="mass").sample(1)


def test_domain_randomizer_formats():
    import torch

    env = make("omo")
    rz = vs.create_default_randomizer(env)
    torch.manual_seed(0)
    rz.randomize(num_samples=3)
    lst = rz.get_params(-1, "list", "numpy")
    assert len(lst) == 3 and set(lst[0]) == {"mass", "stiffness", "damping"} and lst[0]["mass"].dtype == np.float32
    dct = rz.get_params(2, "dict", "torch")
    assert len(dct["mass"]) == 2
    one = rz.get_params(1, "dict", "numpy")
    assert one["mass"].shape == ()
    torch.manual_seed(0)
    rz.randomize(num_samples=3)
    assert rz.get_params(1, "dict", "numpy")["mass"] == one["mass"]  # torch global RNG (Q13)
    with pytest.raises(vs.ValueErr):
        rz.randomize(0)
    with pytest.raises(vs.TypeErr):
        rz.randomize(1.5)
    zr = vs.create_zero_var_randomizer(env)
    zr.randomize(1)
    assert abs(float(zr.get_params(1, "dict", "numpy")["stiffness"]) - 30.0) < 0.05
    for dp in rz.domain_params:  # clipping (domain_parameter.py:123-124)
        dp.adapt("clip_lo", float(dp.mean) * 2)
    rz.randomize(5)
    assert all(float(d["mass"]) == 2.0 for d in rz.get_params(-1, "list", "numpy"))


# ---------------------------------------------------------------------------------------------------- env description
@pytest.mark.parametrize("name", ENVS)
def test_env_spaces_and_task_follow_domain_params(golden_dir, name):
    """spaces / c_max for nominal AND randomised params equal what the reference built (quirk Q11)"""
    g = np.load(os.path.join(golden_dir, f"reset_{name.replace('-', '_')}.npz"))
    env = make(name)
    names = list(env.get_nominal_domain_param().keys())
    for i in range(g["params"].shape[0]):
        env.domain_param = dict(zip(names, g["params"][i]))
        np.testing.assert_allclose(env.state_space.bound_lo, g["state_lo"][i], rtol=1e-15)
        np.testing.assert_allclose(env.state_space.bound_up, g["state_hi"][i], rtol=1e-15)
        np.testing.assert_allclose(env.act_space.bound_up, g["act_hi"][i], rtol=1e-15)
        if name in ("bob", "bob-d"):
            lo = np.concatenate([env.init_space.subspace(0).bound_lo, env.init_space.subspace(1).bound_lo])
            np.testing.assert_allclose(lo, g["init_lo"][i], rtol=1e-14)
        else:
            np.testing.assert_allclose(env.init_space.bound_lo, g["init_lo"][i], rtol=1e-14)
            np.testing.assert_allclose(env.init_space.bound_up, g["init_hi"][i], rtol=1e-14)
        if name in ("bob", "qbb", "bob-d"):
            assert env.task.rew_fcn.c_max == pytest.approx(float(g["c_max"][i]), rel=1e-13)
    assert env.name == name and env.spec.act_space == env.act_space
    assert set(env.supported_domain_param) == set(names)
    env.domain_param = {"act_delay": 3}  # a plain dict.update in the reference (pysim/base.py:117): foreign keys are kept
    assert env.domain_param["act_delay"] == 3 and set(env.supported_domain_param) == set(names)
    with pytest.raises(vs.TypeErr):
        env.domain_param = [1, 2]


def test_env_ctor_errors_and_pickle():
    with pytest.raises(vs.TypeErr):
        vs.QQubeSwingUpSim(dt="a", max_steps=10)
    with pytest.raises(vs.ValueErr):
        vs.QQubeSwingUpSim(dt=-0.1, max_steps=10)
    with pytest.raises(vs.ValueErr):
        vs.QQubeSwingUpSim(dt=0.1, max_steps=0)
    with pytest.raises(vs.TypeErr):
        vs.QQubeSwingUpSim(dt=0.1, max_steps=10, task_args=[1])
    env = vs.QCartPoleSwingUpSim(dt=0.002, max_steps=8000, long=True, wild_init="False")
    assert env.domain_param["pole_mass"] == 0.23 and env.domain_param["pole_length"] == 0.641 / 2
    assert env.init_space.bound_up[0] == 0.02
    env.domain_param = dict(rail_length=1.0)
    e2 = pickle.loads(pickle.dumps(env))  # ctor args + domain params (sim_base.py:115-123)
    assert type(e2) is type(env) and e2.domain_param == env.domain_param and e2.state_space == env.state_space
    assert env.state_space.bound_up[0] == pytest.approx(0.35) and env.state_space.bound_up[2] == 1.0
    env.max_steps = 100
    assert env.max_steps == 100
    with pytest.raises(vs.TypeErr):
        env.max_steps = 1.5
    assert np.array_equal(env.limit_act(np.array([9.0])), [6.0])
    o = vs.QQubeSwingUpSim(dt=0.004).observe(np.array([0.0, np.pi / 2, 1.0, 2.0]))
    np.testing.assert_allclose(o, [0, 1, 1, 0, 1, 2], atol=1e-15)


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = make("qq-su")
    with pytest.raises(RuntimeError, match="no HIP device"):
        env.reset()
    with pytest.raises(RuntimeError, match="no HIP device"):
        env.step(np.zeros(1))


# ---------------------------------------------------------------------------------------------------- wrappers
def test_wrappers_delegate():
    env = make("bob")
    rz = vs.create_default_randomizer(env)
    w = vs.DomainRandWrapperLive(env, rz)
    assert vs.inner_env(w) is env and w.name == "bob" and w.randomizer is rz
    assert isinstance(vs.inner_env(w), vs.SimEnv) and vs.typed_env(w, vs.DomainRandWrapperLive) is w
    assert list(vs.all_envs(w)) == [w, env]
    w.domain_param = dict(beam_length=2.6)
    assert env.domain_param["beam_length"] == 2.6 and w.state_space.bound_up[0] == 1.3  # SURVEY Q11 anchor
    assert w.act_space.bound_up[0] == pytest.approx(38.259)
    assert w.task.rew_fcn.c_max == pytest.approx(3.38531e-05, rel=1e-5)
    assert w.max_steps == 500 and w.dt == 0.01
    with pytest.raises(vs.TypeErr):
        vs.DomainRandWrapperLive(object(), rz)
    with pytest.raises(vs.TypeErr):
        vs.DomainRandWrapperLive(env, "not a randomizer")


# ---------------------------------------------------------------------------------------------------- sampler (host part)
def test_sampler_work_list_and_keys():
    from simurlacra_amd.policies import DummyPolicy

    env = make("omo")
    pol = DummyPolicy(env.spec)
    with pytest.raises(vs.ValueErr):
        ParallelRolloutSampler(env, pol, 2)
    s = ParallelRolloutSampler(env, pol, 4, min_rollouts=5, seed=0)
    assert s.work_list(None, None) == [(None, None)] * 5
    inits = [np.array([-0.7, 0.0]), np.array([-0.66, 0.05])]
    wl = s.work_list(inits, None)  # ceil(5/2)=3 repetitions (parallel_rollout_sampler.py:285-288)
    assert len(wl) == 6 and wl[2][0] is inits[0] and wl[3][0] is inits[1]
    dps = [dict(mass=1.1), dict(mass=0.9), dict(mass=1.0)]
    wl = s.work_list(inits, dps)  # Cartesian product (:296-301)
    assert len(wl) == 6 and wl[0] == (inits[0], dps[0]) and wl[1] == (inits[0], dps[1]) and wl[3] == (inits[1], dps[0])
    s._sample_count = 0
    k0 = s._key()
    s._sample_count = 1
    assert s._key() != k0
    s2 = ParallelRolloutSampler(env, pol, 1, min_rollouts=5, seed=0)
    s2._sample_count = 0
    assert s2._key() == k0  # depends on (seed, sample_count) only, not on num_workers


def test_step_sequence():
    ro = StepSequence(observations=np.zeros((4, 2)), actions=np.zeros((3, 1)), rewards=[1.0, 2.0, 3.0])
    assert len(ro) == 3 and ro.undiscounted_return() == 6.0 and ro.done.tolist() == [False, False, True]
    assert ro.discounted_return(0.5) == 1 + 1 + 0.75
    with pytest.raises(vs.ValueErr):
        StepSequence(observations=np.zeros((3, 2)), actions=np.zeros((3, 1)), rewards=[1.0, 2.0, 3.0])


def test_cvar_selection_and_sampler_counts():
    """select_cvar / CVaRSampler (P/sampling/cvar_sampler.py:40-140): worst epsilon-quantile by discounted return; the inner
    sampler is asked for 1 / epsilon times as much"""
    from simurlacra_amd.sampling import CVaRSampler, select_cvar

    ros = [StepSequence(observations=np.zeros((3, 1)), actions=np.zeros((2, 1)), rewards=[r, 2 * r]) for r in (5.0, -1.0, 3.0, 0.5, 9.0, -4.0, 2.0, 7.0, 1.0, -2.0)]
    worst = select_cvar(list(ros), 0.3)
    assert [ro.undiscounted_return() for ro in worst] == [-12.0, -6.0, -3.0]
    assert [ro.rewards[0] for ro in select_cvar(list(ros), 0.2, gamma=0.5)] == [-4.0, -2.0]
    with pytest.raises(vs.ValueErr):
        select_cvar(list(ros), 0.01)

    class Inner:
        def __init__(self):
            self.counts = None

        def set_min_count(self, min_rollouts=None, min_steps=None):
            self.counts = (min_rollouts, min_steps)

        def reinit(self, env=None, policy=None):
            self.re = (env, policy)

        def sample(self):
            return list(ros)

    inner = Inner()
    cs = CVaRSampler(inner, epsilon=0.2, min_rollouts=4)
    assert inner.counts == (20, None)
    cs.set_min_count(min_steps=1000)
    assert inner.counts == (None, 5000)
    got = cs.sample()
    assert len(got) == 2 and got[0].undiscounted_return() == -12.0
    assert cs.full_stats["full avg return"] == pytest.approx(np.mean([3 * r for r in (5.0, -1.0, 3.0, 0.5, 9.0, -4.0, 2.0, 7.0, 1.0, -2.0)]))
    cs.reinit(env="e", policy="p")
    assert inner.re == ("e", "p")
    with pytest.raises(vs.ValueErr):
        CVaRSampler(inner, epsilon=0.0, min_rollouts=4)


def test_shard_layout():
    from simurlacra_amd.dist import shard

    assert [shard(262144, r, 8) for r in range(8)] == [(r * 32768, 32768) for r in range(8)]  # BASELINE config 4
    parts = [shard(10, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 3), (6, 2), (8, 2)]
    with pytest.raises(ValueError):
        shard(10, 4, 4)


def test_register_with_pyrado_when_the_reference_is_importable():
    """build container only: with the reference importable (stub harness of SURVEY 8(c)), the env classes of this
    package satisfy Pyrado's own isinstance checks after register_with_pyrado()"""
    import sys
    import types

    ref = "/root/reference/Pyrado"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present (GPU box)")
    saved = dict(sys.modules)
    saved_path = list(sys.path)
    try:
        np.float = float
        np.object = object

        def mod(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        class _Blank:
            def __getattr__(self, k):
                return ""

        class Serializable:
            @staticmethod
            def _init(self, locals_):
                pass

        mod("colorama", Style=_Blank(), Fore=_Blank(), Back=_Blank(), init=lambda **k: None)
        mod("ipdb", set_trace=lambda *a, **k: None)
        ias = mod("init_args_serializer", Serializable=Serializable)
        ias.serializable = mod("init_args_serializer.serializable", Serializable=Serializable)
        sys.dont_write_bytecode = True
        sys.path.insert(0, ref)
        from simurlacra_amd.pyrado_compat import register_with_pyrado

        assert register_with_pyrado()
        from pyrado.environment_wrappers.utils import inner_env as pyrado_inner_env
        from pyrado.environments.sim_base import SimEnv as PyradoSimEnv

        env = make("qq-su")
        w = vs.DomainRandWrapperLive(env, vs.create_default_randomizer(env))
        assert isinstance(env, PyradoSimEnv) and isinstance(pyrado_inner_env(w), PyradoSimEnv)
        # rollouts of this package as Pyrado's own StepSequence: what its algorithms call on them
        from pyrado.sampling.step_sequence import StepSequence as PyradoStepSequence
        from pyrado.sampling.step_sequence import discounted_values, gae_returns

        from simurlacra_amd.pyrado_compat import to_pyrado_step_sequences

        rng = np.random.default_rng(0)
        mine = [StepSequence._packed(rng.normal(size=(T + 1, 6)).astype(np.float32), rng.normal(size=(T, 1)).astype(np.float32),
                                     rng.uniform(size=T), ("qq-su", ["g"], np.array([9.81]), 7), True, 0.004, np.zeros(4),
                                     states=rng.normal(size=(T + 1, 4)).astype(np.float32),
                                     actions_applied=rng.normal(size=(T, 1)).astype(np.float32),
                                     th_ddot=rng.normal(size=T + 1).astype(np.float32))
                for T in (5, 9, 3)]
        theirs = to_pyrado_step_sequences(mine)
        assert all(isinstance(r, PyradoStepSequence) for r in theirs)
        for a, b in zip(mine, theirs):
            # what the reference's rollout() puts into its StepSequence (rollout.py:305-325) arrives in Pyrado's container
            assert np.array_equal(b.states, a.states) and np.array_equal(b.actions_applied, a.actions_applied)
            assert np.array_equal(np.asarray(b.th_ddot).reshape(-1), a.th_ddot)
            assert b.length == len(a) and b.undiscounted_return() == pytest.approx(a.undiscounted_return(), rel=1e-12)
            assert b.discounted_return(0.9) == pytest.approx(a.discounted_return(0.9), rel=1e-6)
            assert b.done[-1] and not b.done[:-1].any() and b.rollout_info["env_name"] == "qq-su"
        cat = PyradoStepSequence.concat(theirs)
        assert cat.length == 17 and list(cat.rollout_lengths) == [5, 9, 3]
        cat.torch()
        assert discounted_values(theirs, 0.9).shape[0] == 17
        assert len(list(cat.iterate_rollouts())) == 3
    finally:
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
        sys.path[:] = saved_path
        for attr in ("float", "object"):
            if attr in np.__dict__:
                delattr(np, attr)


def test_domain_rand_wrapper_buffer_and_act_norm_host_side(golden_dir):
    """DomainRandWrapperBuffer ring order and ActNormWrapper spaces / de-normalisation as in the reference
    (environment_wrappers/test_domain_randomization.py:59-88)"""
    g = np.load(os.path.join(golden_dir, "wrappers.npz"))
    env = make("omo")
    w = vs.DomainRandWrapperBuffer(env, None, selection="cyclic")
    with pytest.raises(vs.TypeErr):
        w.fill_buffer(3)  # no randomizer
    w.buffer = [dict(mass=m, stiffness=k, damping=d) for m, k, d in g["omo_buffer"]]
    w.ring_idx = 0
    picked = []
    for i in range(12):  # the reset itself needs the GPU; the ring logic does not
        dp = w.buffer[w.ring_idx]
        picked.append([dp["mass"], dp["stiffness"], dp["damping"]])
        w._ring_idx = (w._ring_idx + 1) % len(w.buffer)
    np.testing.assert_allclose(np.array(picked), g["omo_buffer_seq"], rtol=1e-15)
    with pytest.raises(vs.ValueErr):
        vs.DomainRandWrapperBuffer(env, None, selection="sometimes")
    with pytest.raises(vs.ValueErr):
        w.ring_idx = 99
    rz = vs.create_default_randomizer(env)
    w2 = vs.DomainRandWrapperBuffer(env, rz, selection="random")
    w2.fill_buffer(4)
    assert len(w2.buffer) == 4 and w2.ring_idx == 0 and set(w2.buffer[0]) == {"mass", "stiffness", "damping"}
    an = vs.ActNormWrapper(make("qbb"))
    assert np.array_equal(an.act_space.bound_lo, [-1, -1]) and np.array_equal(an.act_space.bound_up, [1, 1])
    np.testing.assert_allclose(an._process_act(np.array([0.0, 1.0])), [0.0, 3.0])
    np.testing.assert_allclose(an._process_act(np.array([-1.4, 0.5])), [-4.2, 1.5])
    assert vs.inner_env(an).name == "qbb" and an.obs_space == vs.inner_env(an).obs_space


def test_mixed_env_exposes_only_what_the_c_api_has():
    """MixedVecSimEnv wraps the vs_mixed_* entry points and nothing else (per-member data access, parameters, resets and
    Jacobians go through the member handles): a method here without a vs_mixed_* counterpart would hand the vs_mixed
    handle to a function that expects a vs_env"""
    from simurlacra_amd import _lib as L
    from simurlacra_amd.vec_env import MixedVecSimEnv

    public = {k for k, v in vars(MixedVecSimEnv).items() if not k.startswith("_") and callable(v)}
    assert public == {"step_random", "step", "time_random", "sync", "close"}
    assert isinstance(vars(MixedVecSimEnv)["n_envs"], property)
    mixed_api = {k for k in L.exported_symbols() if k.startswith("vs_mixed_")}
    assert mixed_api == {"vs_mixed_create", "vs_mixed_destroy", "vs_mixed_last_error", "vs_mixed_step_random", "vs_mixed_step",
                         "vs_mixed_time_random"}
    assert not hasattr(MixedVecSimEnv, "step_jac") and not hasattr(MixedVecSimEnv, "dims")


def test_fnn_policy_mirror_and_kernel_spec():
    """FNN / FNNPolicy of P/policies/feed_back/fnn.py on the host: layer and parameter order, the fork's featurisation, and
    what fnn_kernel_spec hands to the fused kernel (or refuses)"""
    torch = pytest.importorskip("torch")
    from simurlacra_amd.policies import FNN, FNNPolicy, NormalActNoiseExplStrat, fnn_kernel_spec
    from simurlacra_amd.spaces import BoxSpace, EnvSpec

    net = FNN(6, 1, [32, 16], torch.tanh)
    names = [k for k, _ in net.named_parameters()]
    assert names == ["hidden_layers.0.weight", "hidden_layers.0.bias", "hidden_layers.1.weight", "hidden_layers.1.bias",
                     "output_layer.weight", "output_layer.bias"]
    assert net.param_values.numel() == 6 * 32 + 32 + 32 * 16 + 16 + 16 + 1
    x = torch.randn(5, 6)
    want = net.output_layer(torch.tanh(net.hidden_layers[1](torch.tanh(net.hidden_layers[0](x)))))
    assert torch.equal(net(x), want)
    spec = EnvSpec(BoxSpace(-np.ones(5), np.ones(5)), BoxSpace(-np.ones(1), np.ones(1)), BoxSpace(-np.ones(4), np.ones(4)))
    pol = FNNPolicy(spec, [64, 64], torch.tanh)  # the fork: one input more than observation rows
    assert pol.net.hidden_layers[0].in_features == 6
    obs = torch.randn(3, 5)
    feat = torch.cat([obs[:, :1], torch.sin(obs[:, 1:2]), torch.cos(obs[:, 1:2]), obs[:, 2:]], dim=1)
    assert torch.equal(pol(obs), pol.net(feat)) and pol(obs[0]).shape == (1,)
    plain = FNNPolicy(spec, [8], torch.relu, featurize=False, output_nonlin=torch.tanh)
    assert plain.net.hidden_layers[0].in_features == 5 and torch.equal(plain(obs), plain.net(obs))
    ks = fnn_kernel_spec(pol)
    assert ks["hidden_sizes"] == [64, 64] and ks["hidden_nonlin"] == ["tanh", "tanh"] and ks["feat"] and ks["noise_std"] is None
    assert torch.equal(ks["params"], pol.param_values.detach())
    ks = fnn_kernel_spec(NormalActNoiseExplStrat(plain, std_init=0.2))
    assert ks["output_nonlin"] == "tanh" and ks["hidden_nonlin"] == ["relu"] and np.allclose(ks["noise_std"], 0.2)
    for bad in (FNN(5, 1, [128], torch.tanh), FNN(5, 1, [8] * 5, torch.tanh), FNN(5, 1, [8], torch.nn.functional.elu),
                FNN(5, 1, [8], torch.tanh, dropout=0.1), torch.nn.Linear(5, 1)):
        assert fnn_kernel_spec(bad) is None  # too wide / too deep / unknown nonlinearity / dropout / not an FNN


def test_packed_rollouts_indexing():
    """PackedRollouts: ONE matrix rows[total + n, F], rollout j in rows offsets[j] + j .. offsets[j + 1] + j (its steps, then the
    entry behind them), every field a view of it indexed by the same rows: the slices of rollout j, the rollout of every row and of
    every step, the rows of all steps, per-rollout returns (CPU tensors: the container does not care where they live)"""
    torch = pytest.importorskip("torch")
    from simurlacra_amd.sampling import PackedRollouts

    lengths = torch.tensor([3, 1, 4])
    starts = torch.cumsum(lengths, 0) - lengths
    total, n, F = int(lengths.sum()), 3, 4  # record: [obs (2) | act | rew]
    rows = torch.zeros(total + n, F)
    rows[:, :2] = torch.arange((total + n) * 2, dtype=torch.float32).reshape(total + n, 2)
    p = PackedRollouts(rows=rows, observations=rows[:, :2], actions=rows[:, 2:3], rewards=rows[:, 3], states=None,
                       actions_applied=None, th_ddot=None, lengths=lengths, offsets=torch.cat([starts, starts[-1:] + lengths[-1:]]),
                       total=total, done_last=torch.tensor([True, False, True]), init_states=torch.zeros(n, 2), first_index=0)
    assert len(p) == 3 and p.total_steps == 8
    assert [p.step_slice(j) for j in range(3)] == [slice(0, 3), slice(4, 5), slice(6, 10)]   # the steps of rollout j ...
    assert [p.obs_slice(j) for j in range(3)] == [slice(0, 4), slice(4, 6), slice(6, 11)]    # ... and the entry behind them
    assert p.rollout_index().tolist() == [0, 0, 0, 1, 2, 2, 2, 2]
    assert p.row_rollout_index().tolist() == [0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 2]
    assert p.step_rows().tolist() == [0, 1, 2, 4, 6, 7, 8, 9]
    rows[p.step_rows(), 3] = torch.arange(total, dtype=torch.float32)   # rewards 0 .. 7 on the step rows, 0 on the final entries
    assert p.undiscounted_returns().tolist() == [3.0, 3.0, 22.0]
    assert p.actions[p.step_rows()].shape == (total, 1)  # the dense, reference-style concatenation
