"""
Host-side mirror of the reference's environment interface for the five pysim families, backed by libvecsim.

Every class keeps the `Env` / `SimEnv` / `SimPyEnv` surface of Pyrado (P/environments/base.py:43-236,
P/environments/sim_base.py:39-123, P/environments/pysim/base.py:43-289) -- `reset / step / observe / limit_act`, the
`state`, `*_space`, `spec`, `dt`, `max_steps`, `curr_step`, `task`, `name`, `domain_param` properties,
`get_nominal_domain_param()` and `supported_domain_param` -- so that wrappers, samplers and algorithms written against
Pyrado consume them unchanged.  The computation happens in the HIP kernels: an env object owns one libvecsim handle
with `num_envs` lanes (1 by default = the reference's one-object-one-env semantics; the batched sampler asks for more
and talks to `env.vec` directly).  There is no CPU implementation behind these classes.

Pickling follows `Serializable`: constructor arguments + domain parameters (sim_base.py:115-123); the device handle is
re-created lazily on the other side.
"""
import math
from typing import Optional

import numpy as np

from . import _lib as L
from .exceptions import ShapeErr, TypeErr, ValueErr
from .spaces import BoxSpace, CompoundSpace, DiscreteSpace, EnvSpec, Polar2DPosVelSpace, SingularStateSpace
from .vec_env import VecSimEnv, nominal_params, param_names

inf = float("inf")
PI = np.pi


class RewFcnInfo:
    """Descriptor of the env's reward function (class name of the reference, diag Q / R, c_max where it exists)."""

    def __init__(self, kind, Q, R, c_max=None):
        self.kind, self.Q, self.R, self.c_max = kind, np.diag(Q), np.diag(R), c_max

    def __repr__(self):
        return f"{self.kind}(Q=diag{list(np.diag(self.Q))}, R=diag{list(np.diag(self.R))}, c_max={self.c_max})"


class TaskInfo:
    """What `env.task` exposes of the reference's Task objects: desired state, reward function, spec.
    The reward itself is evaluated inside the step kernel (vecsim.hip: step_reward)."""

    def __init__(self, kind, env_spec, state_des, rew_fcn, final_rew_factor=0.0):
        self.kind, self.env_spec, self.state_des, self.rew_fcn = kind, env_spec, state_des, rew_fcn
        self.final_rew_factor = final_rew_factor

    def __repr__(self):
        return f"{self.kind}(state_des={self.state_des}, rew_fcn={self.rew_fcn})"


class SimEnv:
    """Marker base (P/environments/sim_base.py:39). `isinstance(inner_env(env), SimEnv)` holds for all classes below."""


class VecSimPyEnv(SimEnv):
    """Counterpart of SimPyEnv (P/environments/pysim/base.py:43)."""

    name: str = None
    _REW_KIND = {"omo": "QuadrErrRewFcn", "bob": "ScaledExpQuadrErrRewFcn", "qq-su": "ExpQuadrErrRewFcn",
                 "qcp-su": "ExpQuadrErrRewFcn", "qbb": "ScaledExpQuadrErrRewFcn", "qq-st": "ExpQuadrErrRewFcn",
                 "qcp-st": "QuadrErrRewFcn", "pend": "ExpQuadrErrRewFcn", "bob-d": "ScaledExpQuadrErrRewFcn"}
    _TASK_KIND = {"omo": "FinalRewTask(DesStateTask)", "bob": "DesStateTask", "qq-su": "RadiallySymmDesStateTask",
                  "qcp-su": "RadiallySymmDesStateTask", "qbb": "DesStateTask", "qq-st": "RadiallySymmDesStateTask",
                  "qcp-st": "FinalRewTask(RadiallySymmDesStateTask)", "pend": "RadiallySymmDesStateTask",
                  "bob-d": "DesStateTask"}

    def __init__(self, dt: float, max_steps: int = inf, task_args: Optional[dict] = None, num_envs: int = 1,
                 device: int = 0, **flags):
        if not isinstance(dt, (int, float)):
            raise TypeErr(given=dt, expected_type=(int, float))
        if dt < 0:
            raise ValueErr(given=dt, ge_constraint="0")
        if max_steps < 1:
            raise ValueErr(given=max_steps, ge_constraint="1")
        if not (isinstance(task_args, dict) or task_args is None):
            raise TypeErr(given=task_args, expected_type=dict)
        self._ctor = dict(dt=dt, max_steps=max_steps, task_args=task_args, num_envs=num_envs, device=device, **flags)
        self._dt = float(dt)
        self._max_steps = max_steps
        self._task_args = dict() if task_args is None else task_args
        self._num_envs = int(num_envs)
        self._device = int(device)
        self._flags = flags
        self._domain_param = self.get_nominal_domain_param(**self._nominal_kwargs())
        self._vec = None
        self._curr_rew = -inf
        self._init_space_override = None

    # ------------------------------------------------------------------------------------------ device handle
    def _nominal_kwargs(self):
        return {}

    @property
    def vec(self) -> VecSimEnv:
        """The libvecsim handle (created on first use; fails loudly without a GPU)."""
        if self._vec is None:
            v = VecSimEnv(self.name, self._num_envs, self._dt, self._max_steps, task_args=self._task_args or None,
                          device=self._device, **self._flags)
            v.set_params_uniform(self._domain_param)
            v.use_stream(0)  # an env object is stepped from host code next to torch's default stream: launch on it
            self._vec = v
        return self._vec

    @property
    def num_envs(self) -> int:
        return self._num_envs

    def close(self):
        pass  # SimEnv.close (sim_base.py:111-113)

    def __getstate__(self):
        return dict(ctor=self._ctor, domain_param=self._domain_param)

    def __setstate__(self, st):
        self.__init__(**st["ctor"])
        self._domain_param.update(st["domain_param"])

    # ------------------------------------------------------------------------------------------ static description
    @classmethod
    def get_nominal_domain_param(cls, **kw) -> dict:
        vals = nominal_params(cls.name, long=bool(kw.get("long", False)))
        out = {k: float(np.float64(v)) for k, v in zip(param_names(cls.name), vals)}
        out.update(cls._NOMINAL_F64)
        if kw.get("long"):
            out.update(pole_mass=0.23, pole_length=0.641 / 2)
        if kw.get("mass") is not None:
            out["pole_mass"] = kw["mass"]
        return out

    @property
    def supported_domain_param(self):
        return self.get_nominal_domain_param(**self._nominal_kwargs()).keys()

    @property
    def dt(self) -> float:
        return self._dt

    @dt.setter
    def dt(self, dt):
        if not dt > 0:
            raise ValueErr(given=dt, g_constraint="0")
        if not isinstance(dt, (float, int)):
            raise TypeErr(given=dt, expected_type=[float, int])
        self._dt = float(dt)
        self._ctor["dt"] = dt
        if self._vec is not None:
            self._vec.set_dt(self._dt)  # like the reference's attribute assignment: the state stays

    @property
    def max_steps(self):
        return self._max_steps

    @max_steps.setter
    def max_steps(self, num_steps):
        if not (isinstance(num_steps, int) or num_steps == inf):
            raise TypeErr(msg=f"Number of steps needs to be an integer of infinite, but is {num_steps}")
        if not num_steps > 0:
            raise ValueErr(given=num_steps, g_constraint="0")
        if num_steps != self._max_steps:
            self._max_steps = num_steps
            self._ctor["max_steps"] = num_steps
            if self._vec is not None:
                self._vec.set_max_steps(num_steps)  # like the reference's attribute assignment: the state stays

    def _drop_handle(self):
        if self._vec is not None:
            self._vec.close()
            self._vec = None

    @property
    def curr_step(self) -> int:
        if self._vec is None:
            return 0
        c = self.vec.get(L.VS_STEPCOUNT)
        return int(c[0]) if self._num_envs == 1 else c

    # spaces (host-side description; the kernels carry the same bounds as derived constants)
    def _spaces(self):
        raise NotImplementedError

    @property
    def state_space(self):
        return self._spaces()[0]

    @property
    def obs_space(self):
        return self._spaces()[1]

    @property
    def act_space(self):
        return self._spaces()[2]

    @property
    def init_space(self):
        return self._init_space_override if self._init_space_override is not None else self._spaces()[3]

    @init_space.setter
    def init_space(self, space):
        from .spaces import Space

        if not isinstance(space, Space):
            raise TypeErr(given=space, expected_type=Space)
        self._init_space_override = space

    @property
    def spec(self) -> EnvSpec:
        ss, os_, as_, _ = self._spaces()
        return EnvSpec(os_, as_, ss)

    @property
    def task(self) -> TaskInfo:
        des, qd, rd = VecSimEnv.default_task(self.name)
        ta = self._task_args
        des = np.asarray(ta.get("state_des", des), dtype=np.float64)
        Q = np.asarray(ta.get("Q", np.diag(qd)))
        R = np.asarray(ta.get("R", np.diag(rd)))
        qd = np.diag(Q) if Q.ndim == 2 else Q
        rd = np.diag(R) if R.ndim == 2 else R
        c_max = None
        if self._REW_KIND[self.name].startswith("Scaled"):
            ss, _, as_, _ = self._spaces()
            smax, amax = ss.bound_abs_up, as_.bound_abs_up
            c_max = -1.0 * np.log(1e-4) / (smax.dot(qd * smax) + amax.dot(rd * amax))  # reward_functions.py:284-297
        return TaskInfo(self._TASK_KIND[self.name], self.spec, des, RewFcnInfo(self._REW_KIND[self.name], qd, rd, c_max),
                        1e3 if self.name == "omo" else 0.0)

    # ------------------------------------------------------------------------------------------ domain parameters
    @property
    def domain_param(self) -> dict:
        # a copy, like the reference's getter (pysim/base.py:108-110); keys the env does not know (the wrappers park
        # theirs here: act_delay, obs_noise_std, ...) are kept as in the reference's plain dict.update
        return {**self._domain_param, **getattr(self, "_foreign_domain_param", {})}

    @domain_param.setter
    def domain_param(self, domain_param: dict):
        if not isinstance(domain_param, dict):
            raise TypeErr(given=domain_param, expected_type=dict)
        known = {k: v for k, v in domain_param.items() if k in self._domain_param}
        if len(known) < len(domain_param):
            foreign = dict(getattr(self, "_foreign_domain_param", {}))
            foreign.update({k: v for k, v in domain_param.items() if k not in known})
            self._foreign_domain_param = foreign
        self._domain_param.update({k: float(np.asarray(v).reshape(-1)[0]) for k, v in known.items()})
        if self._vec is not None:
            self._vec.set_params_uniform(self._domain_param)  # _calc_constants + spaces + task.reset on the device

    # ------------------------------------------------------------------------------------------ state
    @property
    def state(self) -> np.ndarray:
        s = self.vec.get(L.VS_STATE).astype(np.float64)
        return s[0] if self._num_envs == 1 else s

    @state.setter
    def state(self, state):
        if not isinstance(state, np.ndarray):
            raise TypeErr(given=state, expected_type=np.ndarray)
        S = self.vec.dims["S"]
        st = state.reshape(1, -1) if self._num_envs == 1 else state
        if st.shape != (self._num_envs, S):
            raise ShapeErr(given=state, expected_match=(S,))
        self.vec.put(L.VS_STATE, st.astype(np.float32))

    # ------------------------------------------------------------------------------------------ reset / step
    def _first_obs(self, obs, state):
        return obs

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        """SimPyEnv.reset (pysim/base.py:166-203)"""
        v = self.vec
        if domain_param is not None:
            self.domain_param = domain_param
        if init_state is None:
            init_state = np.stack([np.asarray(self.init_space.sample_uniform()) for _ in range(self._num_envs)])
        elif not isinstance(init_state, np.ndarray):
            try:
                init_state = np.asarray(init_state)
            except Exception:
                raise TypeErr(given=init_state, expected_type=np.ndarray)
        init = init_state.reshape(1, -1) if init_state.ndim == 1 else init_state
        if init.shape[0] != self._num_envs or init.shape[1] not in (v.dims["I"], v.dims["S"]):
            raise ShapeErr(given=init_state, expected_match=self.init_space)
        if init.shape[1] != v.dims["S"]:
            # an element of the init space (a state-shaped one is copied verbatim, unchecked: pysim/base.py:184-188):
            # non-fatal containment check -- the reference only prints (pysim/base.py:189-193)
            for row in init:
                try:
                    inside = self.init_space.contains(np.asarray(row, dtype=np.float64))
                except Exception:  # (NaN / shape errors surface in the step, as in the reference)
                    inside = True
                if not inside:
                    print("The  init state is not within init state space.")
                    break
        v.reset(init_state=init.astype(np.float32))
        obs = v.get(L.VS_OBS).astype(np.float64)
        state = v.get(L.VS_STATE).astype(np.float64)
        obs = self._first_obs(obs, state)
        return obs[0] if self._num_envs == 1 else obs

    def step(self, act: np.ndarray) -> tuple:
        """SimPyEnv.step (pysim/base.py:217-241), one kernel launch. NaN in act/state raises ValueErr like
        BoxSpace.contains does in the reference."""
        import torch

        v = self.vec
        a = np.asarray(act, dtype=np.float32).reshape(self._num_envs, -1)
        if a.shape[1] != v.dims["A"]:
            raise ShapeErr(given=act, expected_match=self.act_space)
        v.step(torch.from_numpy(a).to(f"cuda:{self._device}"))
        obs = v.get(L.VS_OBS).astype(np.float64)
        rew = v.get(L.VS_REW).astype(np.float64)
        done = v.get(L.VS_DONE).astype(bool)
        v.raise_on_error()
        if self._num_envs == 1:
            self._curr_rew = float(rew[0])
            return obs[0], float(rew[0]), bool(done[0]), dict()
        return obs, rew, done, dict()

    def observe(self, state: np.ndarray) -> np.ndarray:
        return np.asarray(state).copy()

    def limit_act(self, act: np.ndarray) -> np.ndarray:
        return self.act_space.project_to(act)

    def render(self, mode=None, render_step: int = 1):
        pass


# ---------------------------------------------------------------------------------------------------------------- envs
class OneMassOscillatorSim(VecSimPyEnv):
    """P/environments/pysim/one_mass_oscillator.py:49-121"""

    name = "omo"
    _NOMINAL_F64 = dict(mass=1.0, stiffness=30.0, damping=0.5)

    def _spaces(self):
        k = self._domain_param["stiffness"]
        max_state = np.array([1.0, 10.0])
        ss = BoxSpace(-max_state, max_state, labels=["x", "x_dot"])
        init = BoxSpace(np.array([-0.75 * max_state[0], -0.01 * max_state[1]]),
                        np.array([-0.65 * max_state[0], +0.01 * max_state[1]]), labels=["x", "x_dot"])
        max_act = np.array([max_state[0] * k])
        return ss, ss, BoxSpace(-max_act, max_act, labels=["F"]), init


class BallOnBeamSim(VecSimPyEnv):
    """P/environments/pysim/ball_on_beam.py:41-136"""

    name = "bob"
    _NOMINAL_F64 = dict(gravity_const=9.81, ball_mass=0.5, ball_radius=0.1, beam_mass=3.0, beam_length=2.0,
                        beam_thickness=0.1, friction_coeff=0.05, ang_offset=0.0)

    def _spaces(self):
        l_beam, g = self._domain_param["beam_length"], self._domain_param["gravity_const"]
        max_state = np.array([l_beam / 2.0, PI / 4.0, 10.0, PI])
        max_act = np.array([l_beam / 2.0 * g * 3.0])
        labels = ["x", "alpha", "x_dot", "alpha_dot"]
        ss = BoxSpace(-max_state, max_state, labels=labels)
        a, v, w = 5 / 180.0 * PI, 0.02 * max_state[2], 0.02 * max_state[3]
        init = CompoundSpace([
            BoxSpace(np.array([-0.8 * l_beam / 2.0, -a, -v, -w]), np.array([-0.7 * l_beam / 2.0, a, v, w]), labels=labels),
            BoxSpace(np.array([0.7 * l_beam / 2.0, -a, -v, -w]), np.array([0.8 * l_beam / 2.0, a, v, w]), labels=labels)])
        return ss, ss, BoxSpace(-max_act, max_act, labels=["tau"]), init


class QQubeSwingUpSim(VecSimPyEnv):
    """P/environments/pysim/quanser_qube.py:152-188"""

    name = "qq-su"
    _NOMINAL_F64 = dict(gravity_const=9.81, motor_resistance=8.4, motor_back_emf=0.042, mass_rot_pole=0.095,
                        length_rot_pole=0.085, damping_rot_pole=5e-6, mass_pend_pole=0.024, length_pend_pole=0.129,
                        damping_pend_pole=1e-6, voltage_thold_neg=0, voltage_thold_pos=0)

    def _spaces(self):
        max_state = np.array([115.0 / 180 * PI, 4 * PI, 20 * PI, 20 * PI])
        max_init = np.array([2.0, 1.0, 0.5, 0.5]) / 180 * PI
        max_obs = np.array([1.0, 1.0, 1.0, 1.0, 20 * PI, 20 * PI])
        lab = ["theta", "alpha", "theta_dot", "alpha_dot"]
        return (BoxSpace(-max_state, max_state, labels=lab),
                BoxSpace(-max_obs, max_obs, labels=["sin_theta", "cos_theta", "sin_alpha", "cos_alpha", "theta_dot", "alpha_dot"]),
                BoxSpace(-4.5, 4.5, shape=(1,), labels=["V"]),  # MAX_ACT_QQ
                BoxSpace(-max_init, max_init, labels=lab))

    def observe(self, state):  # quanser_qube.py:148-149
        s = np.asarray(state)
        return np.array([np.sin(s[0]), np.cos(s[0]), np.sin(s[1]), np.cos(s[1]), s[2], s[3]])


class QCartPoleSwingUpSim(VecSimPyEnv):
    """P/environments/pysim/quanser_cartpole.py:507-587 (ctor defaults :515-524)"""

    name = "qcp-su"
    _NOMINAL_F64 = dict(gravity_const=9.81, cart_mass=0.58, rail_length=0.814, motor_efficiency=0.9, gear_efficiency=0.9,
                        gear_ratio=3.71, motor_inertia=3.9e-7, pinion_radius=6.35e-3, motor_resistance=2.6,
                        motor_back_emf=7.67e-3, pole_damping=0.0024, combined_damping=5.4, pole_mass=0.127,
                        pole_length=0.3365 / 2, cart_friction_coeff=0.02, voltage_thold_neg=0, voltage_thold_pos=0)

    def __init__(self, dt: float, max_steps: int = inf, task_args: Optional[dict] = None, long: bool = False,
                 simple_dynamics: bool = False, wild_init: str = "True", mass=None, num_envs: int = 1, device: int = 0):
        self._long, self._mass, self._wild_init = long, mass, wild_init
        super().__init__(dt, max_steps, task_args, num_envs=num_envs, device=device, long=long,
                         simple_dynamics=simple_dynamics, wild_init=wild_init)
        self._ctor = dict(dt=dt, max_steps=max_steps, task_args=task_args, long=long, simple_dynamics=simple_dynamics,
                          wild_init=wild_init, mass=mass, num_envs=num_envs, device=device)

    def _nominal_kwargs(self):
        return dict(long=self._long, mass=self._mass)

    def _spaces(self):
        l_rail = self._domain_param["rail_length"]
        max_state = np.array([l_rail / 2.0 - 0.15, 4 * PI, l_rail, 20 * PI])
        if self._wild_init == "True":
            max_init = np.array([0.25, PI, 0.8, PI])
        elif self._wild_init == "False":
            max_init = np.array([0.02, 2 / 180.0 * PI, 0.0, 1 / 180.0 * PI])
        else:
            max_init = np.array([0.02, PI, 0.0, 1 / 180.0 * PI])
        max_obs = np.array([l_rail / 2.0, 1.0, 1.0, np.inf, np.inf])
        lab = ["x", "theta", "x_dot", "theta_dot"]
        return (BoxSpace(-max_state, max_state, labels=lab),
                BoxSpace(-max_obs, max_obs, labels=["x", "sin_theta", "cos_theta", "x_dot", "theta_dot"]),
                BoxSpace(-6.0, 6.0, shape=(1,), labels=["V"]),  # MAX_ACT_QCP
                BoxSpace(-max_init, max_init, labels=lab))

    def _first_obs(self, obs, state):
        return state  # QCartPoleSim.reset returns the 4-D state, not the observation (quirk Q5, :101-105)

    def observe(self, state):  # :107-108
        s = np.asarray(state)
        return np.array([s[0], np.sin(s[1]), np.cos(s[1]), s[2], s[3]])

    @property
    def th_ddot(self):
        h = self.vec.get(L.VS_HIDDEN)
        return float(h[0, 0]) if self._num_envs == 1 else h[:, 0]


class QBallBalancerSim(VecSimPyEnv):
    """P/environments/pysim/quanser_ball_balancer.py:49-337"""

    name = "qbb"
    _NOMINAL_F64 = dict(gravity_const=9.81, ball_mass=0.003, ball_radius=0.019625, plate_length=0.275, arm_radius=0.0254,
                        gear_ratio=70.0, gear_efficiency=0.9, load_inertia=5.2822e-5, motor_inertia=4.6063e-7,
                        motor_back_emf=0.0077, motor_resistance=2.6, motor_efficiency=0.69, combined_damping=0.015,
                        ball_damping=0.05, voltage_thold_x_pos=0.28, voltage_thold_x_neg=-0.10, voltage_thold_y_pos=0.28,
                        voltage_thold_y_neg=-0.074, offset_th_x=0.0, offset_th_y=0.0)

    def __init__(self, dt: float, max_steps: int = inf, task_args: Optional[dict] = None, simple_dynamics: bool = False,
                 load_experimental_tholds: bool = True, num_envs: int = 1, device: int = 0):
        super().__init__(dt, max_steps, task_args, num_envs=num_envs, device=device, simple_dynamics=simple_dynamics)
        self._ctor = dict(dt=dt, max_steps=max_steps, task_args=task_args, simple_dynamics=simple_dynamics,
                          load_experimental_tholds=load_experimental_tholds, num_envs=num_envs, device=device)

    def _spaces(self):
        l_plate = self._domain_param["plate_length"]
        max_state = np.array([PI / 4.0, PI / 4.0, l_plate / 2.0, l_plate / 2.0, 5 * PI, 5 * PI, 0.5, 0.5])
        lab = ["theta_x", "theta_y", "x", "y", "theta_x_dot", "theta_y_dot", "x_dot", "y_dot"]
        ss = BoxSpace(-max_state, max_state, labels=lab)
        init = Polar2DPosVelSpace(np.array([0.75 * l_plate / 2, -PI, -0.05 * max_state[6], -0.05 * max_state[7]]),
                                  np.array([0.8 * l_plate / 2, PI, 0.05 * max_state[6], 0.05 * max_state[7]]),
                                  labels=["r", "phi", "x_dot", "y_dot"])
        return ss, ss.copy(), BoxSpace(-3.0, 3.0, shape=(2,), labels=["V_x", "V_y"]), init  # MAX_ACT_QBB

    @property
    def plate_angs(self):
        h = self.vec.get(L.VS_HIDDEN).astype(np.float64)
        return h[0] if self._num_envs == 1 else h


# ------------------------------------------------------------------------------------- remaining pysim families
class QQubeStabSim(QQubeSwingUpSim):
    """P/environments/pysim/quanser_qube.py:191-222"""

    name = "qq-st"

    def _spaces(self):
        ss, os_, as_, _ = super()._spaces()
        lab = ["theta", "alpha", "theta_dot", "alpha_dot"]
        init = BoxSpace(np.array([-5.0 / 180 * PI, 175.0 / 180 * PI, 0, 0]), np.array([5.0 / 180 * PI, 185.0 / 180 * PI, 0, 0]),
                        labels=lab)
        return ss, os_, as_, init


class QCartPoleStabSim(QCartPoleSwingUpSim):
    """P/environments/pysim/quanser_cartpole.py:441-504 (ctor defaults long=True, simple_dynamics=True)"""

    name = "qcp-st"
    stab_thold = 15 / 180.0 * PI
    max_init_th_offset = 8 / 180.0 * PI

    def __init__(self, dt: float, max_steps: int = inf, task_args: Optional[dict] = None, long: bool = True,
                 simple_dynamics: bool = True, num_envs: int = 1, device: int = 0):
        super().__init__(dt, max_steps, task_args, long=long, simple_dynamics=simple_dynamics, wild_init="False",
                         num_envs=num_envs, device=device)
        self._ctor = dict(dt=dt, max_steps=max_steps, task_args=task_args, long=long, simple_dynamics=simple_dynamics,
                          num_envs=num_envs, device=device)

    def _spaces(self):
        _, os_, as_, _ = super()._spaces()
        l_rail = self._domain_param["rail_length"]
        lab = ["x", "theta", "x_dot", "theta_dot"]
        ss = BoxSpace(np.array([-l_rail / 2.0 + 0.15, PI - self.stab_thold, -l_rail, -2 * PI]),
                      np.array([+l_rail / 2.0 - 0.15, PI + self.stab_thold, +l_rail, +2 * PI]), labels=lab)
        init = BoxSpace(np.array([-0.02, PI - self.max_init_th_offset, -0.02, -5 / 180 * PI]),
                        np.array([+0.02, PI + self.max_init_th_offset, +0.02, +5 / 180 * PI]), labels=lab)
        return ss, os_, as_, init


class PendulumSim(VecSimPyEnv):
    """P/environments/pysim/pendulum.py:43-117"""

    name = "pend"
    _NOMINAL_F64 = dict(gravity_const=9.81, pole_mass=1.0, pole_length=1.0, pole_damping=0.05, torque_thold=3.5)

    def __init__(self, dt: float, max_steps: int = inf, task_args: Optional[dict] = None,
                 init_state: Optional[np.ndarray] = None, num_envs: int = 1, device: int = 0):
        self._init_state = np.zeros(2) if init_state is None else np.asarray(init_state, dtype=np.float64)
        if self._init_state.size != 2:
            raise ShapeErr(given=self._init_state, expected_match=(2,))
        super().__init__(dt, max_steps, task_args, num_envs=num_envs, device=device, init_state=self._init_state)
        self._ctor = dict(dt=dt, max_steps=max_steps, task_args=task_args, init_state=self._init_state, num_envs=num_envs,
                          device=device)

    def _spaces(self):
        max_state = np.array([4 * PI, 4 * PI])
        max_obs = np.array([1.0, 1.0, np.inf])
        tau_max = self._domain_param["torque_thold"]
        return (BoxSpace(-max_state, max_state, labels=["theta", "theta_dot"]),
                BoxSpace(-max_obs, max_obs, labels=["sin_theta", "cos_theta", "theta_dot"]),
                BoxSpace(-tau_max, tau_max, shape=(1,), labels=["tau"]),
                SingularStateSpace(self._init_state, labels=["theta", "theta_dot"]))

    def observe(self, state):  # :91-92
        s = np.asarray(state)
        return np.array([np.sin(s[0]), np.cos(s[0]), s[1]])


class BallOnBeamDiscSim(BallOnBeamSim):
    """P/environments/pysim/ball_on_beam.py:139-161: three discrete torques {-max, 0, +max}"""

    name = "bob-d"

    def _spaces(self):
        ss, os_, as_, init = super()._spaces()
        lo, hi = as_.bounds
        return ss, os_, DiscreteSpace(np.linspace(lo, hi, num=3, endpoint=True), labels=["tau"]), init


ENV_CLASSES = {c.name: c for c in (OneMassOscillatorSim, BallOnBeamSim, QQubeSwingUpSim, QCartPoleSwingUpSim,
                                   QBallBalancerSim, QQubeStabSim, QCartPoleStabSim, PendulumSim, BallOnBeamDiscSim)}
