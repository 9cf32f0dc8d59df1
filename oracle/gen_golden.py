"""
oracle/gen_golden.py -- generate golden vectors by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  Imports Pyrado from /root/reference/Pyrado with the local stub harness of SURVEY.md section 8(c)
(three absent third-party modules stubbed in sys.modules, two removed NumPy aliases restored) and dumps, for the five
SimPyEnv families, single-step cases, short trajectories, reset cases and the default-randomizer tables as small
``.npz`` / ``.json`` fixtures under ``tests/golden/``.  The fixtures are data (inputs + the reference's outputs);
no reference source is copied.  Nothing here runs on the GPU box (``/root/reference`` does not exist there).

Usage:  python oracle/gen_golden.py            (deterministic: fixed seeds)
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
REF = "/root/reference/Pyrado"


def _install_stubs():
    sys.dont_write_bytecode = True
    np.float = float  # removed NumPy aliases still used by the reference (Q7)
    np.object = object

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Blank:
        def __getattr__(self, k):
            return ""

    mod("colorama", Style=_Blank(), Fore=_Blank(), Back=_Blank(), init=lambda **k: None)
    mod("ipdb", set_trace=lambda *a, **k: None)

    class Serializable:
        @staticmethod
        def _init(self, locals_):
            pass

    ias = mod("init_args_serializer", Serializable=Serializable)
    ias.serializable = mod("init_args_serializer.serializable", Serializable=Serializable)
    sys.path.insert(0, REF)


_install_stubs()
import torch  # noqa: E402

import pyrado  # noqa: E402
from pyrado.domain_randomization.default_randomizers import create_default_randomizer  # noqa: E402
from pyrado.environments.pysim.ball_on_beam import BallOnBeamDiscSim, BallOnBeamSim  # noqa: E402
from pyrado.environments.pysim.one_mass_oscillator import OneMassOscillatorSim  # noqa: E402
from pyrado.environments.pysim.pendulum import PendulumSim  # noqa: E402
from pyrado.environments.pysim.quanser_ball_balancer import QBallBalancerSim  # noqa: E402
from pyrado.environments.pysim.quanser_cartpole import QCartPoleStabSim, QCartPoleSwingUpSim  # noqa: E402
from pyrado.environments.pysim.quanser_qube import QQubeStabSim, QQubeSwingUpSim  # noqa: E402

# env kwargs = Pyrado/tests/conftest.py:160-193
ENVS = {
    "omo": (OneMassOscillatorSim, dict(dt=0.02, max_steps=300)),
    "bob": (BallOnBeamSim, dict(dt=0.01, max_steps=500)),
    "qq-su": (QQubeSwingUpSim, dict(dt=0.004, max_steps=4000)),
    "qcp-su": (QCartPoleSwingUpSim, dict(dt=0.002, max_steps=8000)),
    "qbb": (QBallBalancerSim, dict(dt=0.01, max_steps=500)),
    # the remaining pysim families, kwargs = Pyrado/tests/conftest.py:164-189
    "qq-st": (QQubeStabSim, dict(dt=0.01, max_steps=500)),
    "qcp-st": (QCartPoleStabSim, dict(dt=0.01, max_steps=300)),
    "pend": (PendulumSim, dict(dt=0.02, max_steps=400, init_state=np.array([0.1, 0.2]))),
    "bob-d": (BallOnBeamDiscSim, dict(dt=0.01, max_steps=500)),
}
HIDDEN = {"omo": 0, "bob": 0, "qq-su": 0, "qcp-su": 1, "qbb": 2, "qq-st": 0, "qcp-st": 1, "pend": 0, "bob-d": 0}


def get_hidden(name, env):
    if name in ("qcp-su", "qcp-st"):
        return np.array([float(env._th_ddot)])
    if name == "qbb":
        return np.array(env.plate_angs, dtype=np.float64).copy()
    return np.zeros(0)


def set_hidden(name, env, h):
    if name in ("qcp-su", "qcp-st"):
        env._th_ddot = float(h[0])
    elif name == "qbb":
        env.plate_angs = np.array(h, dtype=np.float64).copy()


def param_names(env):
    return list(env.get_nominal_domain_param().keys())


def params_to_vec(env, dp):
    return np.array([float(dp[k]) for k in param_names(env)], dtype=np.float64)


def vec_to_params(env, vec):
    return {k: float(v) for k, v in zip(param_names(env), vec)}


def draw_params(name, env, rng, randomizer, mode):
    """mode 0: nominal, 1: default randomizer draw (fp32 values, Q13), 2: like 1 plus non-trivial dead-zone tholds"""
    dp = env.get_nominal_domain_param()
    if mode >= 1:
        randomizer.randomize(num_samples=1)
        for k, v in randomizer.get_params(fmt="dict", dtype="numpy").items():
            dp[k] = float(v)
    if mode == 2 and name in ("qq-su", "qcp-su", "qq-st", "qcp-st"):
        dp["voltage_thold_neg"] = float(np.float32(-rng.uniform(0.05, 0.6)))
        dp["voltage_thold_pos"] = float(np.float32(rng.uniform(0.05, 0.6)))
    return dp


def space_info(env):
    out = dict(state_lo=env.state_space.bound_lo, state_hi=env.state_space.bound_up,
               act_lo=env.act_space.bound_lo, act_hi=env.act_space.bound_up)
    c_max = getattr(env.task.rew_fcn, "c_max", None) if hasattr(env.task, "rew_fcn") else None
    out["c_max"] = np.array(np.nan if c_max is None else float(c_max))
    return {k: np.array(v, dtype=np.float64) for k, v in out.items()}


def gen_step_cases(name, m, seed):
    cls, kw = ENVS[name]
    env = cls(**kw)
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    randomizer = create_default_randomizer(env)
    rec = {k: [] for k in ("params", "state", "hidden", "act", "curr_step", "nstate", "nhidden", "obs", "rew", "done",
                           "state_lo", "state_hi", "act_lo", "act_hi", "c_max")}
    H = HIDDEN[name]
    for i in range(m):
        dp = draw_params(name, env, rng, randomizer, i % 3)
        env.reset(init_state=np.zeros(env.state_space.shape), domain_param=dp)  # applies params, constants, spaces
        info = space_info(env)
        lo, hi = info["state_lo"], info["state_hi"]
        # state: inside the box; every 6th case hugs one face (done flips), every 7th has one coord outside
        u = rng.uniform(-1, 1, size=lo.shape)
        if i % 6 == 0:
            j = rng.integers(lo.size)
            u[j] = np.sign(u[j]) * rng.uniform(0.9995, 1.0005)
        if i % 7 == 0:
            j = rng.integers(lo.size)
            u[j] = np.sign(u[j]) * rng.uniform(1.0, 1.05)
        state = 0.5 * (lo + hi) + 0.5 * (hi - lo) * u
        if H == 1:
            hidden = rng.uniform(-120, 120, size=1)
        elif H == 2:
            hidden = rng.uniform(-0.3, 0.3, size=2)
        else:
            hidden = np.zeros(0)
        amax = info["act_hi"]
        act = rng.uniform(-1.5, 1.5, size=amax.shape) * amax
        if i % 5 == 0:
            act = rng.uniform(-0.3, 0.3, size=amax.shape)  # small -> dead-zone candidates
        if i % 11 == 0:
            act = np.zeros_like(amax)
        max_steps = kw["max_steps"]
        curr = int(rng.integers(0, max_steps - 1))
        if i % 9 == 0:
            curr = max_steps - 1  # this step times out
        env.state = state.copy()
        env._curr_step = curr
        set_hidden(name, env, hidden)
        obs, rew, done, _ = env.step(act.copy())
        rec["params"].append(params_to_vec(env, env.domain_param))
        rec["state"].append(state)
        rec["hidden"].append(hidden)
        rec["act"].append(act)
        rec["curr_step"].append(curr)
        rec["nstate"].append(np.array(env.state, dtype=np.float64).copy())
        rec["nhidden"].append(get_hidden(name, env))
        rec["obs"].append(np.array(obs, dtype=np.float64))
        rec["rew"].append(float(rew))
        rec["done"].append(bool(done))
        for k in ("state_lo", "state_hi", "act_lo", "act_hi", "c_max"):
            rec[k].append(info[k])
    out = {k: np.array(v) for k, v in rec.items()}
    out["dt"] = np.array(kw["dt"])
    out["max_steps"] = np.array(kw["max_steps"])
    out["param_names"] = np.array(param_names(env))
    return out


def gen_traj(name, n_traj, t_max, seed, extra_after_done=3):
    """reset(init_state, domain_param) followed by up to t_max random-action steps; keeps stepping a few steps after
    done (rollout(stop_on_done=False) semantics) so the once-only final reward is covered."""
    cls, kw = ENVS[name]
    env = cls(**kw)
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    randomizer = create_default_randomizer(env)
    S = env.state_space.shape[0]
    A = env.act_space.shape[0]
    O = env.obs_space.shape[0]
    H = HIDDEN[name]
    P = len(param_names(env))
    z = lambda *s: np.full(s, np.nan)  # noqa: E731
    out = dict(params=z(n_traj, P), init=[], reset_obs=[], length=np.zeros(n_traj, dtype=np.int64),
               act=z(n_traj, t_max, A), state=z(n_traj, t_max + 1, S), hidden=z(n_traj, t_max + 1, H),
               obs=z(n_traj, t_max, O), rew=z(n_traj, t_max), done=np.zeros((n_traj, t_max), dtype=bool))
    for i in range(n_traj):
        dp = draw_params(name, env, rng, randomizer, 0 if i < n_traj // 2 else 1)
        env.reset(domain_param=dp)  # spaces for these params
        init = env.init_space.sample_uniform()
        if name in ("omo", "bob", "bob-d") and i % 2 == 1:
            # start close to a bound so the episode fails early (covers failure malus / done)
            init = np.array(init)
            init[0] = 0.97 * env.state_space.bound_up[0]
            init = np.concatenate([init]) if init.shape == env.state_space.shape else init
        robs = env.reset(init_state=np.array(init, dtype=np.float64), domain_param=dp)
        out["params"][i] = params_to_vec(env, env.domain_param)
        out["init"].append(np.array(init, dtype=np.float64))
        out["reset_obs"].append(np.array(robs, dtype=np.float64))
        out["state"][i, 0] = env.state
        out["hidden"][i, 0] = get_hidden(name, env)
        after = 0
        amax = env.act_space.bound_up
        for t in range(t_max):
            act = rng.uniform(-1.2, 1.2, size=amax.shape) * amax
            obs, rew, done, _ = env.step(act.copy())
            out["act"][i, t] = act
            out["obs"][i, t] = obs
            out["rew"][i, t] = rew
            out["done"][i, t] = done
            out["state"][i, t + 1] = env.state
            out["hidden"][i, t + 1] = get_hidden(name, env)
            out["length"][i] = t + 1
            if done:
                after += 1
                if after > extra_after_done:
                    break
    out["init"] = np.array(out["init"])
    out["reset_obs"] = np.array(out["reset_obs"])
    out["dt"] = np.array(kw["dt"])
    out["max_steps"] = np.array(kw["max_steps"])
    return out


def gen_reset(name, m, seed):
    """reset() with explicit init states of init-space shape and of full-state shape, nominal + randomised params"""
    cls, kw = ENVS[name]
    env = cls(**kw)
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    randomizer = create_default_randomizer(env)
    rec = {k: [] for k in ("params", "init", "full", "state", "hidden", "obs", "state_lo", "state_hi", "act_lo",
                           "act_hi", "c_max", "init_lo", "init_hi")}
    for i in range(m):
        dp = draw_params(name, env, rng, randomizer, i % 2)
        env.reset(domain_param=dp)
        full = bool(i % 4 >= 2)
        if full:
            lo, hi = env.state_space.bound_lo, env.state_space.bound_up
            init = rng.uniform(0.5 * lo, 0.5 * hi)
            pad = init
        else:
            init = np.array(env.init_space.sample_uniform(), dtype=np.float64)
            pad = np.concatenate([init, np.full(env.state_space.shape[0] - init.shape[0], np.nan)])
        obs = env.reset(init_state=init.copy(), domain_param=dp)
        info = space_info(env)
        rec["params"].append(params_to_vec(env, env.domain_param))
        rec["init"].append(pad)
        rec["full"].append(full)
        rec["state"].append(np.array(env.state, dtype=np.float64).copy())
        rec["hidden"].append(get_hidden(name, env))
        rec["obs"].append(np.array(obs, dtype=np.float64))
        for k in ("state_lo", "state_hi", "act_lo", "act_hi", "c_max"):
            rec[k].append(info[k])
        if name in ("bob", "bob-d"):
            sp = env.init_space.subspace(0)
            sp1 = env.init_space.subspace(1)
            rec["init_lo"].append(np.concatenate([sp.bound_lo, sp1.bound_lo]))
            rec["init_hi"].append(np.concatenate([sp.bound_up, sp1.bound_up]))
        else:
            rec["init_lo"].append(env.init_space.bound_lo)
            rec["init_hi"].append(env.init_space.bound_up)
    return {k: np.array(v) for k, v in rec.items()}


def gen_cfg1():
    """BASELINE.json configs[0] / SURVEY 8(d) config 1: OMO, 1 env, 500 forward-Euler steps, actions U(-30, 30)"""
    env = OneMassOscillatorSim(dt=0.02, max_steps=500)
    rng = np.random.default_rng(0)
    obs0 = env.reset(init_state=np.array([-0.7, 0.0]))
    acts, states, rews, dones = [], [np.array(env.state)], [], []
    for _ in range(500):
        a = rng.uniform(-30, 30, size=1)
        obs, rew, done, _ = env.step(a.copy())
        acts.append(a)
        states.append(np.array(env.state))
        rews.append(rew)
        dones.append(done)
    return dict(obs0=obs0, act=np.array(acts), state=np.array(states), rew=np.array(rews), done=np.array(dones))


def gen_randomizer_tables():
    tables = {}
    for name, (cls, kw) in ENVS.items():
        env = cls(**kw)
        rows = []
        for dp in create_default_randomizer(env).domain_params:
            kind = type(dp).__name__
            row = dict(name=dp.name, kind=kind, mean=float(dp.mean), clip_lo=float(dp.clip_lo),
                       clip_up=float(dp.clip_up))
            row["spread"] = float(dp.std) if kind == "NormalDomainParam" else float(dp.halfspan)
            rows.append(row)
        tables[name] = dict(nominal={k: float(v) for k, v in env.get_nominal_domain_param().items()}, randomizer=rows)
    return tables


def gen_ik(seed=3):
    """QBallBalancerKin on a sweep of servo angles and geometries (torch fp32 SGD, Q8)"""
    env = QBallBalancerSim(dt=0.01, max_steps=500)
    rng = np.random.default_rng(seed)
    th, r, l, ang = [], [], [], []  # noqa: E741
    for i in range(48):
        dp = env.get_nominal_domain_param()
        if i >= 16:
            dp["arm_radius"] = float(np.float32(rng.uniform(0.018, 0.033)))
            dp["plate_length"] = float(np.float32(rng.uniform(0.2, 0.35)))
        env.reset(domain_param=dp)
        t = float(rng.uniform(-0.6, 0.6)) if i % 4 else 0.0
        th.append(t)
        r.append(dp["arm_radius"])
        l.append(dp["plate_length"] / 2.0)
        ang.append(env._kin(t))
    return dict(th=np.array(th), r=np.array(r), l=np.array(l), ang=np.array(ang))


def gen_wrappers():
    """ActNormWrapper (action_normalization.py:63-89) single steps and the set order of a cyclic DomainRandWrapperBuffer
    (domain_randomization.py:151-261), straight from the reference classes"""
    from pyrado.environment_wrappers.action_normalization import ActNormWrapper
    from pyrado.environment_wrappers.domain_randomization import DomainRandWrapperBuffer

    out = {}
    for name in ("qq-su", "qbb"):
        cls, kw = ENVS[name]
        env = ActNormWrapper(cls(**kw))
        rng = np.random.default_rng(7)
        rec = {k: [] for k in ("state", "act", "nstate", "rew", "done", "obs")}
        for i in range(48):
            env.reset()
            s = rng.uniform(env.state_space.bound_lo, env.state_space.bound_up) * 0.8
            env.state = s.copy()
            if name == "qbb":
                env.wrapped_env.plate_angs = np.zeros(2)
            a = rng.uniform(-1.4, 1.4, size=env.act_space.shape)
            obs, rew, done, _ = env.step(a.copy())
            for k, v in zip(("state", "act", "nstate", "rew", "done", "obs"), (s, a, np.array(env.state), rew, done, obs)):
                rec[k].append(v)
        assert np.array_equal(env.act_space.bound_up, np.ones(env.act_space.shape))
        for k, v in rec.items():
            out[f"{name.replace('-', '_')}_{k}"] = np.array(v)
    cls, kw = ENVS["omo"]
    e = cls(**kw)
    torch.manual_seed(0)
    w = DomainRandWrapperBuffer(e, create_default_randomizer(e), selection="cyclic")
    w.fill_buffer(5)
    out["omo_buffer"] = np.array([[float(d[k]) for k in ("mass", "stiffness", "damping")] for d in w.buffer])
    seq = []
    for i in range(12):
        w.reset()
        seq.append([w.domain_param[k] for k in ("mass", "stiffness", "damping")])
    out["omo_buffer_seq"] = np.array(seq)
    return out


VARIANTS = {
    # ctor / task options of the reference classes that change the hot path (flags of vs_task_cfg on the device)
    "qcp_su_simple": ("qcp-su", dict(dt=0.002, max_steps=8000, simple_dynamics=True)),
    "qcp_su_long": ("qcp-su", dict(dt=0.002, max_steps=8000, long=True)),
    "qcp_su_tame_init": ("qcp-su", dict(dt=0.002, max_steps=8000, wild_init="False")),
    "qbb_simple": ("qbb", dict(dt=0.01, max_steps=500, simple_dynamics=True)),
    "qq_su_task_args": ("qq-su", dict(dt=0.004, max_steps=4000, task_args=dict(
        state_des=np.array([0.3, -np.pi, 0.5, -0.2]), Q=np.diag([2.0, 0.5, 1e-2, 1e-3]), R=np.diag([1e-2])))),
    "bob_task_args": ("bob", dict(dt=0.01, max_steps=500, task_args=dict(
        state_des=np.array([0.2, 0.0, 0.0, 0.0]), Q=np.diag([1e4, 1e2, 1e2, 1e1]), R=np.diag([0.5])))),
    "omo_inf_steps": ("omo", dict(dt=0.02)),  # max_steps = pyrado.inf: remaining_steps = 0, no time-out
    "qcp_st_short_pole": ("qcp-st", dict(dt=0.01, max_steps=300, long=False, simple_dynamics=False)),
}


def gen_variants(m=40, seed=900):
    out = {}
    for vi, (tag, (name, kw)) in enumerate(VARIANTS.items()):
        cls = ENVS[name][0]
        env = cls(**kw)
        rng = np.random.default_rng(seed + vi)
        rec = {k: [] for k in ("state", "hidden", "act", "curr_step", "nstate", "nhidden", "obs", "rew", "done")}
        dp = env.domain_param
        for i in range(m):
            env.reset(init_state=np.zeros(env.state_space.shape))
            lo, hi = env.state_space.bound_lo, env.state_space.bound_up
            u = rng.uniform(-1, 1, size=lo.shape)
            if i % 5 == 0:
                j = rng.integers(lo.size)
                u[j] = np.sign(u[j]) * rng.uniform(0.999, 1.02)
            state = 0.5 * (lo + hi) + 0.5 * (hi - lo) * u
            hidden = rng.uniform(-60, 60, size=1) if HIDDEN[name] == 1 else (rng.uniform(-0.2, 0.2, size=2) if HIDDEN[name] == 2 else np.zeros(0))
            act = rng.uniform(-1.3, 1.3, size=env.act_space.shape) * env.act_space.bound_up
            curr = int(rng.integers(0, 200))
            env.state = state.copy()
            env._curr_step = curr
            set_hidden(name, env, hidden)
            obs, rew, done, _ = env.step(act.copy())
            for k, v in zip(rec, (state, hidden, act, curr, np.array(env.state, dtype=np.float64), get_hidden(name, env),
                                  np.array(obs, dtype=np.float64), float(rew), bool(done))):
                rec[k].append(v)
        for k, v in rec.items():
            out[f"{tag}__{k}"] = np.array(v)
        out[f"{tag}__params"] = params_to_vec(env, dp)
        if hasattr(env.init_space, "bound_lo") and name != "qbb":  # Box init spaces only (bob: compound, qbb: polar)
            init = np.array([env.init_space.sample_uniform() for _ in range(64)])
            out[f"{tag}__init_lo"] = env.init_space.bound_lo
            out[f"{tag}__init_hi"] = env.init_space.bound_up
            assert ((init >= env.init_space.bound_lo) & (init <= env.init_space.bound_up)).all()
    return out


# wrapper stacks (outermost first); the same spec builds the reference objects here, the oracle's WrappedRef and the
# package's own wrapper classes in the tests
CHAINS = {
    "qq_delay3": ("qq-su", [dict(kind="act_delay", delay=3)]),
    "qq_norm_delay_bias": ("qq-su", [dict(kind="act_norm"), dict(kind="act_delay", delay=2),
                                     dict(kind="act_noise", mean=[0.2], std=[0.0])]),
    "qbb_noise_over_norm": ("qbb", [dict(kind="act_noise", mean=[0.1, -0.2], std=[0.05, 0.1]), dict(kind="act_norm")]),
    "qq_obsnorm_over_bias": ("qq-su", [
        dict(kind="obs_norm", lb={"theta_dot": -20.0, "alpha_dot": -25.0}, ub={"theta_dot": 20.0, "alpha_dot": 35.0}),
        dict(kind="obs_noise", mean=[0.01, -0.02, 0.03, 0.0, 0.5, -0.4], std=[0.0] * 6)]),
    "qbb_noise_norm_partial": ("qbb", [dict(kind="obs_noise", mean=[0.0] * 6, std=[0.01, 0.02, 0.03, 0.04, 0.05, 0.06]),
                                       dict(kind="obs_norm"), dict(kind="obs_partial", idcs=[0, 1])]),
    "bob_partial_norm_delay": ("bob", [dict(kind="obs_partial", mask=[0, 0, 0, 1]), dict(kind="obs_norm"),
                                       dict(kind="act_delay", delay=1), dict(kind="act_norm")]),
    # ObsNormWrapper cannot wrap QCartPole in the reference: its reset() returns the 4-D state (Q5) -> action side only
    "qcp_delay_over_noise": ("qcp-su", [dict(kind="act_delay", delay=2), dict(kind="act_noise", mean=[-0.3], std=[0.5])]),
    "omo_everything": ("omo", [
        dict(kind="obs_noise", mean=[0.0, 0.1], std=[0.02, 0.3]), dict(kind="act_delay", delay=2),
        dict(kind="obs_norm"), dict(kind="act_noise", mean=[0.5], std=[2.0]), dict(kind="act_norm"),
        dict(kind="obs_noise", mean=[0.05, 0.0], std=[0.01, 0.1])]),
}


def build_reference_chain(env, stages):
    from pyrado.environment_wrappers.action_delay import ActDelayWrapper
    from pyrado.environment_wrappers.action_noise import GaussianActNoiseWrapper
    from pyrado.environment_wrappers.action_normalization import ActNormWrapper
    from pyrado.environment_wrappers.observation_noise import GaussianObsNoiseWrapper
    from pyrado.environment_wrappers.observation_normalization import ObsNormWrapper
    from pyrado.environment_wrappers.observation_partial import ObsPartialWrapper

    for st in reversed(stages):  # innermost first
        k = st["kind"]
        if k == "act_norm":
            env = ActNormWrapper(env)
        elif k == "act_delay":
            env = ActDelayWrapper(env, delay=st["delay"])
        elif k == "act_noise":
            env = GaussianActNoiseWrapper(env, noise_mean=np.array(st["mean"]), noise_std=np.array(st["std"]))
        elif k == "obs_norm":
            env = ObsNormWrapper(env, explicit_lb=st.get("lb"), explicit_ub=st.get("ub"))
        elif k == "obs_noise":
            env = GaussianObsNoiseWrapper(env, noise_std=np.array(st["std"]), noise_mean=np.array(st["mean"]))
        elif k == "obs_partial":
            env = ObsPartialWrapper(env, mask=st.get("mask"), idcs=st.get("idcs"))
        else:
            raise ValueError(k)
    return env


def gen_chains(n_ep=3, T=30, seed=1200):
    """trajectories of wrapped reference envs.  Noise: np.random.seed(base + 1000 * episode + event) right before
    reset() (event 0) and before step t (event t + 1), so that a checker can replay the very same randn() calls."""
    from pyrado.environment_wrappers.utils import inner_env

    out = {"spec": np.array(json.dumps({k: dict(env=v[0], stages=v[1]) for k, v in CHAINS.items()}))}
    for ci, (tag, (name, stages)) in enumerate(CHAINS.items()):
        cls, kw = ENVS[name]
        env = build_reference_chain(cls(**kw), stages)
        base = inner_env(env)
        rng = np.random.default_rng(seed + ci)
        rec = {k: [] for k in ("s0", "h0", "obs0", "act", "obs", "rew", "done", "state", "hidden", "length")}
        for ep in range(n_ep):
            lo, hi = base.state_space.bound_lo, base.state_space.bound_up
            s0 = 0.3 * rng.uniform(lo, hi)
            np.random.seed(seed + 1000 * ep + 0)
            obs0 = env.reset(init_state=s0.copy())
            ep_rec = {k: [] for k in ("act", "obs", "rew", "done", "state", "hidden")}
            h0 = get_hidden(name, base)
            alo, ahi = env.act_space.bound_lo, env.act_space.bound_up
            length = T
            for t in range(T):
                a = rng.uniform(1.2 * alo, 1.2 * ahi)
                np.random.seed(seed + 1000 * ep + t + 1)
                obs, rew, done, _ = env.step(a.copy())
                for k, v in zip(ep_rec, (a, np.array(obs, dtype=np.float64), float(rew), bool(done),
                                          np.array(base.state, dtype=np.float64), get_hidden(name, base))):
                    ep_rec[k].append(v)
                if done and length == T:
                    length = t + 1
            rec["s0"].append(s0)
            rec["h0"].append(h0)
            rec["obs0"].append(np.array(obs0, dtype=np.float64))
            rec["length"].append(length)
            for k, v in ep_rec.items():
                rec[k].append(np.array(v))
        for k, v in rec.items():
            out[f"{tag}__{k}"] = np.array(v)
        out[f"{tag}__params"] = params_to_vec(base, base.domain_param)
        out[f"{tag}__obs_lo"] = np.array(env.obs_space.bound_lo, dtype=np.float64)
        out[f"{tag}__obs_hi"] = np.array(env.obs_space.bound_up, dtype=np.float64)
        out[f"{tag}__act_lo"] = np.array(env.act_space.bound_lo, dtype=np.float64)
        out[f"{tag}__act_hi"] = np.array(env.act_space.bound_up, dtype=np.float64)
    out["seed"] = np.array(seed)
    return out


DP_CASES = [
    ("BernoulliDomainParam", dict(name="mass", val_0=1.0, val_1=3.0, prob_1=0.3, clip_up=2.5)),
    ("BernoulliDomainParam", dict(name="stiffness", val_0=20.4, val_1=41.6, prob_1=0.7, roundint=True)),
    ("MultivariateNormalDomainParam", dict(name="damping", mean=[0.5], cov=[[0.04]], clip_lo=0.3)),
    ("NormalDomainParam", dict(name="stiffness", mean=30.0, std=4.0, roundint=True)),
    ("UniformDomainParam", dict(name="mass", mean=1.0, halfspan=0.5, clip_lo=0.8, clip_up=1.3)),
]


def gen_domain_param_samples(n=24):
    """draws of the reference's DomainParam classes under torch.manual_seed: the host classes of the package make the very
    same torch calls and must reproduce them bit for bit"""
    from pyrado.domain_randomization import domain_parameter as dpm

    out = []
    for i, (cls, kw) in enumerate(DP_CASES):
        torch.manual_seed(4000 + i)
        dp = getattr(dpm, cls)(**kw)
        smp = dp.sample(n)
        out.append(dict(cls=cls, kwargs=kw, seed=4000 + i, dtype=str(smp[0].dtype),
                        samples=[np.asarray(t.detach().numpy(), dtype=np.float64).reshape(-1).tolist() for t in smp],
                        mean=np.asarray(dp.mean, dtype=np.float64).reshape(-1).tolist(), fields=dp.get_field_names()))
    return out


def gen_jacobians(m=96, seed=1500):
    """Step Jacobians from the FORK'S OWN autograd path: QCartPoleSwingUpSim.step_diff_state
    (quanser_cartpole.py:257-278 -> _step_dynamics_diff 323-358 -> _dynamics_diff 360-431 through the tensor rk4 591-655) and
    torch.autograd.grad taken exactly as the fork's rollout does (P/sampling/rollout.py:832-837):
        obs_grad[i] = d next_state[i] / d (state, act),   rew_grad = d rew / d (state, act)
    The fork wrote this against a torch that still had torch.solve (removed in 1.13); the harness restores that one alias
    (torch.solve(B, A) -> (torch.linalg.solve(A, B), None)), like np.float / np.object.  Everything runs in float64 (the fork
    casts to whatever dtype its th_ddot tensor has).

    Where the fork's differentiable restatement is NOT the NumPy step it stands in for, the cases stay inside the region in
    which the two agree (recorded in DESIGN.md section 2):
      * rew_fn (quanser_cartpole.py:295-321) wraps only the pole-angle error and does so with `//`, which floored toward
        zero in the fork's torch and floors toward -inf in torch >= 1.13: cases keep err_theta = pi - theta in [0, 2 pi)
        (both give the same), and |err| <= pi in the other dimensions (RadiallySymmDesStateTask folds all of them, Q4);
      * next states stay inside the state space (the fork's is_done uses strict bounds, irrelevant for the Jacobians).
    Cases: nominal and randomised domain parameters, actions inside the action box and clipped (zero action gradient),
    the voltage dead zone off (thresholds 0) and on (+-0.5 V with actions inside it), hidden pole acceleration from a
    running episode."""
    # torch >= 1.13 keeps `torch.solve` only as a stub that raises and names this very replacement
    torch.solve = lambda B, A: (torch.linalg.solve(A, B), None)
    cls, kw = ENVS["qcp-su"]
    env = cls(**kw)
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    randomizer = create_default_randomizer(env)
    rec = {k: [] for k in ("params", "state", "hidden", "act", "nstate", "nhidden", "rew", "jac_state", "jac_rew")}
    for i in range(m):
        dp = draw_params("qcp-su", env, rng, randomizer, i % 3)
        if i % 4 == 3:  # dead zone on
            dp = dict(dp, voltage_thold_neg=-0.5, voltage_thold_pos=0.5)
        env.reset(init_state=np.zeros(4), domain_param=dp)
        env._th_ddot_tensor = torch.zeros(1, dtype=torch.float64)
        env._create_task(dict())  # refresh the task tensors (state_des, Q, R, bounds) in float64
        info = space_info(env)
        hi = info["state_hi"]
        state = np.array([rng.uniform(-0.8, 0.8) * hi[0], rng.uniform(-np.pi + 0.05, np.pi - 0.05),
                          rng.uniform(-0.8, 0.8) * min(hi[2], np.pi), rng.uniform(-0.9, 0.9) * np.pi])
        act = rng.uniform(-5.5, 5.5, size=1)
        if i % 4 == 3:
            act = rng.uniform(-0.45, 0.45, size=1)  # inside the dead zone
        if i % 5 == 4:
            act = np.sign(act) * rng.uniform(6.2, 9.0, size=1)  # clipped by the action box
        hidden = rng.uniform(-60, 60, size=1)
        st = torch.tensor(state[None, :], dtype=torch.float64, requires_grad=True)
        at = torch.tensor(act[None, :], dtype=torch.float64, requires_grad=True)
        thdd = torch.tensor(hidden, dtype=torch.float64)
        nxt, rew, done, infod = env.step_diff_state(st, at, at * 0, thdd)
        jac_s = torch.stack([torch.cat(torch.autograd.grad(nxt[:, k].sum(), [st, at], retain_graph=True), dim=1)
                             for k in range(nxt.shape[1])], dim=1)  # rollout.py:836
        jac_r = torch.cat(torch.autograd.grad(rew.sum(), [st, at]), dim=1)  # rollout.py:837
        # the NumPy step from the same inputs: the value the differentiable one must reproduce in this region
        env.state = state.copy()
        env._curr_step = 10
        env._th_ddot = float(hidden[0])
        _, rew_np, _, _ = env.step(act.copy())
        nstate_np = np.array(env.state, dtype=np.float64)
        if not (np.allclose(nstate_np, nxt.detach().numpy()[0], rtol=1e-9, atol=1e-11) and
                abs(rew_np - float(rew)) <= 1e-6 * max(1.0, abs(rew_np))):  # the fork keeps state_des / Q / R in float32 (pi!)
            raise RuntimeError(f"case {i}: the fork's differentiable step and its NumPy step disagree inside the agreed region: "
                               f"{nstate_np} vs {nxt.detach().numpy()[0]}, rew {rew_np} vs {float(rew)}")
        rec["params"].append(params_to_vec(env, env.domain_param))
        rec["state"].append(state)
        rec["hidden"].append(hidden)
        rec["act"].append(act)
        rec["nstate"].append(nstate_np)
        rec["nhidden"].append(np.array([float(env._th_ddot)]))
        rec["rew"].append(float(rew_np))
        rec["jac_state"].append(jac_s.detach().numpy()[0])  # [S, S + A]
        rec["jac_rew"].append(jac_r.detach().numpy()[0])    # [S + A]
    out = {k: np.array(v) for k, v in rec.items()}
    out["dt"] = np.array(kw["dt"])
    out["max_steps"] = np.array(kw["max_steps"])
    out["param_names"] = np.array(param_names(env))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    m_step = {"omo": 256, "bob": 256, "qq-su": 256, "qcp-su": 256, "qbb": 128, "qq-st": 192, "qcp-st": 192, "pend": 192,
              "bob-d": 192}
    t_traj = {"omo": 300, "bob": 200, "qq-su": 200, "qcp-su": 200, "qbb": 150, "qq-st": 150, "qcp-st": 150, "pend": 200,
              "bob-d": 150}
    force = "--force" in sys.argv
    for i, name in enumerate(ENVS):
        tag = name.replace("-", "_")
        if os.path.exists(os.path.join(OUT, f"step_{tag}.npz")) and not force:
            continue  # fixtures are committed; only new families are generated (pass --force to redo everything)
        np.savez_compressed(os.path.join(OUT, f"step_{tag}.npz"), **gen_step_cases(name, m_step[name], 100 + i))
        np.savez_compressed(os.path.join(OUT, f"traj_{tag}.npz"), **gen_traj(name, 8, t_traj[name], 200 + i))
        np.savez_compressed(os.path.join(OUT, f"reset_{tag}.npz"), **gen_reset(name, 16, 300 + i))
        print("wrote", name, flush=True)
    if force or not os.path.exists(os.path.join(OUT, "cfg1_omo_500.npz")):
        np.savez_compressed(os.path.join(OUT, "cfg1_omo_500.npz"), **gen_cfg1())
        np.savez_compressed(os.path.join(OUT, "qbb_ik.npz"), **gen_ik())
    if force or not os.path.exists(os.path.join(OUT, "variants.npz")):
        np.savez_compressed(os.path.join(OUT, "variants.npz"), **gen_variants())
    if force or not os.path.exists(os.path.join(OUT, "chains.npz")):
        np.savez_compressed(os.path.join(OUT, "chains.npz"), **gen_chains())
    if force or not os.path.exists(os.path.join(OUT, "jac_qcp_su.npz")):
        np.savez_compressed(os.path.join(OUT, "jac_qcp_su.npz"), **gen_jacobians())
        print("wrote jac_qcp_su", flush=True)
    if force or not os.path.exists(os.path.join(OUT, "wrappers.npz")):
        np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **gen_wrappers())
    with open(os.path.join(OUT, "domain_params.json"), "w") as fh:
        json.dump(gen_domain_param_samples(), fh, indent=1, sort_keys=True)
    with open(os.path.join(OUT, "randomizers.json"), "w") as fh:
        json.dump(gen_randomizer_tables(), fh, indent=1, sort_keys=True)
    # seed KAT straight from the reference function (table also in Pyrado/tests/test_set_seed.py:35-54)
    seeds = [[b, s, ss, pyrado.set_seed(b, s, ss)] for b in (0, 1, 7, 1001) for s in (None, 0, 1, 5) for ss in
             (None, 0, 1, 255)]
    with open(os.path.join(OUT, "set_seed.json"), "w") as fh:
        json.dump(seeds, fh)
    print("done")


if __name__ == "__main__":
    main()
