"""
oracle/cpu_ref.py -- CPU restatement of Pyrado's SimPyEnv hot path (TEST INFRASTRUCTURE ONLY).

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  ``simurlacra_amd`` never imports anything from ``oracle/``.

Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py) against
  * golden vectors produced by running the reference itself in the build container
    (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``), and
  * the two known-answer tests the reference holds for this path
    (``Pyrado/tests/test_tasks.py:82-99`` radial fold, ``Pyrado/tests/test_set_seed.py:33-59`` seed table).

Everything is vectorised over a leading env axis N and written op-for-op after the reference
(``P/`` = ``/root/reference/Pyrado/pyrado/``), including its quirks:

  step orchestration     P/environments/pysim/base.py:217-241
  reset                  P/environments/pysim/base.py:166-215
  OneMassOscillatorSim   P/environments/pysim/one_mass_oscillator.py:54-114
  BallOnBeamSim          P/environments/pysim/ball_on_beam.py:49-129
  QQubeSwingUpSim        P/environments/pysim/quanser_qube.py:54-188
  QCartPoleSwingUpSim    P/environments/pysim/quanser_cartpole.py:88-230, 545-587, 591-655
  QBallBalancerSim       P/environments/pysim/quanser_ball_balancer.py:92-330, 375-444
  tasks / rewards        P/tasks/desired_state.py:107-155, P/tasks/reward_functions.py:202-297,
                         P/tasks/final_reward.py:111-174, P/tasks/base.py:159-180
  spaces                 P/spaces/box.py:138-184, P/spaces/base.py:66-69, P/spaces/polar.py:108-113
  seeding                P/__init__.py:135-183
  wrappers (WrappedRef)  P/environment_wrappers/base.py:288-381, action_normalization.py:66-72, action_noise.py:71-76,
                         action_delay.py:87-112, observation_normalization.py:117-120, observation_noise.py:67-72,
                         observation_partial.py:70-71

``dtype=np.float64`` reproduces the reference arithmetic (it computes in float64); ``dtype=np.float32`` runs the same
expression tree in single precision and is used only to reason about done-mask flips next to a bound.
"""
import hashlib
from collections import OrderedDict

import numpy as np

PI = np.pi

# --------------------------------------------------------------------------------------------------------------------
# seeding -- P/__init__.py:135-183
# --------------------------------------------------------------------------------------------------------------------


def derive_seed(base_seed, sub_seed=None, sub_sub_seed=None):
    """MD5 of "{base}-{sub}-{subsub}" crushed to 32 bit (P/__init__.py:160-168). Returns None for a non-int base."""
    if sub_seed is None:
        sub_seed = 0
    if sub_sub_seed is None:
        sub_sub_seed = 0
    if not isinstance(base_seed, int):
        return None
    return int(hashlib.md5(f"{base_seed}-{sub_seed}-{sub_sub_seed}".encode()).hexdigest(), 16) % (2 ** 32)


# --------------------------------------------------------------------------------------------------------------------
# reward functions and tasks
# --------------------------------------------------------------------------------------------------------------------


def weighted_quadr_cost(err_s, err_a, Qd, Rd):
    """err_s.dot(Q.dot(err_s)) + err_a.dot(R.dot(err_a)) for diagonal Q, R (P/tasks/reward_functions.py:212-221)."""
    cs = np.zeros(err_s.shape[0], dtype=err_s.dtype)
    for j in range(err_s.shape[1]):
        cs = cs + err_s[:, j] * (Qd[j] * err_s[:, j])
    ca = np.zeros(err_a.shape[0], dtype=err_a.dtype)
    for j in range(err_a.shape[1]):
        ca = ca + err_a[:, j] * (Rd[j] * err_a[:, j])
    return cs + ca


def radial_fold(err, idcs, mod=2 * PI):
    """RadiallySymmDesStateTask.step_rew error treatment (P/tasks/desired_state.py:146-153), quirk Q4:
    fmod only on ``idcs``; the two +-pi folds are applied sequentially to ALL state dims."""
    err = err.copy()
    mod = np.broadcast_to(np.asarray(mod, dtype=err.dtype), (len(idcs),))
    for k, j in enumerate(idcs):
        err[:, j] = np.fmod(err[:, j], mod[k])
    dt = err.dtype.type
    m = err > dt(PI)
    err[m] = dt(2 * PI) - err[m]
    m = err < dt(-PI)
    err[m] = dt(-2 * PI) - err[m]
    return err


def c_max_scaled(state_abs_max, act_abs_max, Qd, Rd, min_rew=1e-4):
    """ScaledExpQuadrErrRewFcn.reset (P/tasks/reward_functions.py:284-297); inputs [N, S], [N, A]."""
    max_cost = weighted_quadr_cost(state_abs_max, act_abs_max, Qd, Rd)
    return state_abs_max.dtype.type(-1.0) * np.log(state_abs_max.dtype.type(min_rew)) / max_cost


# --------------------------------------------------------------------------------------------------------------------
# environment specifications
# --------------------------------------------------------------------------------------------------------------------

REW_QUADR, REW_EXP, REW_SCALED_EXP = 0, 1, 2


class EnvRef:
    """Base of the per-env restatements. Sub-classes fill in the class attributes and the static hooks."""

    name = None
    S = A = O = H = I = 0  # widths: state, act, obs, hidden, init-space element
    param_names = ()
    nominal = ()
    state_des = None
    Qd = None
    Rd = None
    rew_kind = None
    radial_idcs = ()  # RadiallySymmDesStateTask idcs (empty -> plain DesStateTask)
    final_rew_factor = 0.0  # FinalRewTask(always_negative) malus, 0 -> no FinalRewTask
    final_state_time_dependent = False  # FinalRewTask(FinalRewMode(state_dependent=True, time_dependent=True))

    def __init__(self, dt, max_steps, dtype=np.float64, task_args=None, **flags):
        self.dt = float(dt)
        self.max_steps = max_steps  # int or float('inf')
        self.dtype = np.dtype(dtype)
        self.flags = flags
        task_args = task_args or {}
        f = self.dtype.type
        self._state_des = np.asarray(task_args.get("state_des", self.state_des), dtype=self.dtype)
        Q = np.asarray(task_args.get("Q", np.diag(self.Qd)), dtype=np.float64)
        R = np.asarray(task_args.get("R", np.diag(self.Rd)), dtype=np.float64)
        assert np.count_nonzero(Q - np.diag(np.diag(Q))) == 0 and np.count_nonzero(R - np.diag(np.diag(R))) == 0
        self._Qd = np.diag(Q).astype(self.dtype)
        self._Rd = np.diag(R).astype(self.dtype)
        self._f = f

    # ---- hooks ----
    @classmethod
    def nominal_params(cls, n=1, dtype=np.float64, **flags):
        return np.tile(np.asarray(cls.nominal, dtype=dtype), (n, 1))

    def p(self, params, name):
        return params[:, self.param_names.index(name)]

    def bounds(self, params):
        """-> (state_lo, state_hi, act_lo, act_hi), each [N, dim]"""
        raise NotImplementedError

    def init_bounds(self, params):
        raise NotImplementedError

    def dynamics(self, state, hidden, act, params):
        """-> (next_state, next_hidden); ``act`` is already clipped"""
        raise NotImplementedError

    def observe(self, state):
        return state.copy()  # P/environments/base.py:203-213

    def state_from_init(self, init):
        return init.copy()  # P/environments/pysim/base.py:205-215

    def init_hidden(self, state, params):
        return np.zeros((state.shape[0], self.H), dtype=self.dtype)

    def limit_act(self, act_raw, alo, ahi):
        return np.clip(act_raw, alo, ahi)  # BoxSpace.project_to (P/spaces/box.py:180-184)

    # ---- generic pieces ----
    def c_max(self, params):
        slo, shi, alo, ahi = self.bounds(params)
        # Space.bound_abs_up (P/spaces/base.py:66-69)
        smax = np.maximum(np.abs(slo), np.abs(shi))
        amax = np.maximum(np.abs(alo), np.abs(ahi))
        return c_max_scaled(smax, amax, self._Qd, self._Rd)

    def step_rew(self, state, act_raw, params):
        err = self._state_des[None, :] - state
        if len(self.radial_idcs):
            err = radial_fold(err, list(self.radial_idcs))
        cost = weighted_quadr_cost(err, -act_raw, self._Qd, self._Rd)
        if self.rew_kind == REW_QUADR:
            return -cost
        if self.rew_kind == REW_EXP:
            return np.exp(-cost)
        return np.exp(-self.c_max(params) * cost)

    def reset(self, params, init_state, init_is_full_state=None):
        """SimPyEnv.reset with an explicit init_state (P/environments/pysim/base.py:166-203).
        -> dict(state, hidden, obs, curr_step)"""
        init_state = np.asarray(init_state, dtype=self.dtype)
        params = np.asarray(params, dtype=self.dtype)
        if init_is_full_state is None:
            init_is_full_state = init_state.shape[1] == self.S
        state = init_state.copy() if init_is_full_state else self.state_from_init(init_state)
        hidden = self.init_hidden(state, params)
        return dict(state=state, hidden=hidden, obs=self.reset_obs(state),
                    curr_step=np.zeros(state.shape[0], dtype=np.int64))

    def reset_obs(self, state):
        return self.observe(state)

    def step(self, state, hidden, act_raw, params, curr_step, yielded=None):
        """One SimPyEnv.step (P/environments/pysim/base.py:217-241) for N envs.
        -> dict(state, hidden, obs, rew, done, curr_step, err, failed, yielded)"""
        f = self._f
        state = np.asarray(state, dtype=self.dtype)
        hidden = np.asarray(hidden, dtype=self.dtype).reshape(state.shape[0], self.H)
        act_raw = np.asarray(act_raw, dtype=self.dtype)
        params = np.asarray(params, dtype=self.dtype)
        curr_step = np.asarray(curr_step, dtype=np.int64)
        slo, shi, alo, ahi = self.bounds(params)
        if self.flags.get("act_norm", False):
            # ActNormWrapper._process_act (P/environment_wrappers/action_normalization.py:66-72): the inner env only ever
            # sees the de-normalised action
            act_raw = alo + (act_raw + f(1)) * (ahi - alo) / f(2)

        # reward on the pre-step state and the unclipped action (Q3)
        rew = self.step_rew(state, act_raw, params)
        # limit_act -> BoxSpace.project_to (P/spaces/box.py:180-184); NaN raises in the reference -> err flag here
        err = np.isnan(act_raw).any(axis=1)
        act = self.limit_act(act_raw, alo, ahi)
        nstate, nhidden = self.dynamics(state, hidden, act, params)
        nstep = curr_step + 1
        err = err | np.isnan(nstate).any(axis=1)
        with np.errstate(invalid="ignore"):
            failed = ((nstate < slo) | (nstate > shi)).any(axis=1)  # not contains(), inclusive bounds (Q9)
        done = failed.copy()
        if self.max_steps != float("inf"):
            done = done | (nstep >= int(self.max_steps))
        if yielded is None:
            yielded = np.zeros(state.shape[0], dtype=bool)
        nyielded = yielded.copy()
        if self.final_rew_factor != 0.0:
            # FinalRewTask(always_negative).compute_final_rew, once per episode (P/tasks/final_reward.py:130-135,165-174)
            pay = done & ~yielded
            rew = rew + np.where(pay & failed, f(-1.0 * abs(self.final_rew_factor)), f(0.0))
            nyielded = yielded | pay
        elif self.final_state_time_dependent:
            # state- and time-dependent mode (final_reward.py:215-226): -remaining_steps * |step_rew(s', act=0)| on failure;
            # remaining_steps is the value computed before the step (pysim/base.py:219)
            pay = done & ~yielded
            remaining = (int(self.max_steps) - nstep) if self.max_steps != float("inf") else np.zeros_like(nstep)
            zero_act = np.zeros_like(act_raw)
            mal = f(-1.0) * remaining.astype(self.dtype) * np.abs(self.step_rew(nstate, zero_act, params))
            rew = rew + np.where(pay & failed, mal, f(0.0))
            nyielded = yielded | pay
        return dict(state=nstate, hidden=nhidden, obs=self.observe(nstate), rew=rew, done=done, curr_step=nstep,
                    err=err, failed=failed, yielded=nyielded)


# ---------------------------------------------------------------------------------------------------------------- OMO
class OneMassOscillatorRef(EnvRef):
    """P/environments/pysim/one_mass_oscillator.py:49-121"""

    name = "omo"
    S, A, O, H, I = 2, 1, 2, 0, 2
    param_names = ("mass", "stiffness", "damping")
    nominal = (1.0, 30.0, 0.5)
    state_des = np.zeros(2)
    Qd = (1e1, 1e-2)
    Rd = (1e-6,)
    rew_kind = REW_QUADR
    final_rew_factor = 1e3

    def bounds(self, params):
        n = params.shape[0]
        f = self._f
        k = self.p(params, "stiffness")
        smax = np.tile(np.array([1.0, 10.0], dtype=self.dtype), (n, 1))
        amax = (f(1.0) * k)[:, None]
        return -smax, smax, -amax, amax

    def init_bounds(self, params):
        n = params.shape[0]
        lo = np.tile(np.array([-0.75 * 1.0, -0.01 * 10.0], dtype=self.dtype), (n, 1))
        hi = np.tile(np.array([-0.65 * 1.0, +0.01 * 10.0], dtype=self.dtype), (n, 1))
        return lo, hi

    def dynamics(self, state, hidden, act, params):
        f = self._f
        m, k, d = (self.p(params, n) for n in self.param_names)
        omega = np.sqrt(k / m)
        zeta = d / (f(2.0) * np.sqrt(m * k))
        a10 = -(omega ** 2)
        a11 = f(-2.0) * zeta * omega
        # A.dot(state) + B.dot(act)
        sd0 = f(0.0) * state[:, 0] + f(1.0) * state[:, 1] + f(0.0) * act[:, 0]
        sd1 = a10 * state[:, 0] + a11 * state[:, 1] + (f(1.0) / m) * act[:, 0]
        ns = np.stack([state[:, 0] + sd0 * f(self.dt), state[:, 1] + sd1 * f(self.dt)], axis=1)  # forward Euler
        return ns, hidden


# ---------------------------------------------------------------------------------------------------------------- BoB
class BallOnBeamRef(EnvRef):
    """P/environments/pysim/ball_on_beam.py:41-136"""

    name = "bob"
    S, A, O, H, I = 4, 1, 4, 0, 4
    param_names = ("gravity_const", "ball_mass", "ball_radius", "beam_mass", "beam_length", "beam_thickness",
                   "friction_coeff", "ang_offset")
    nominal = (9.81, 0.5, 0.1, 3.0, 2.0, 0.1, 0.05, 0.0)
    state_des = np.zeros(4)
    Qd = (1e5, 1e3, 1e3, 1e2)
    Rd = (1.0,)
    rew_kind = REW_SCALED_EXP

    def bounds(self, params):
        f = self._f
        l_beam = self.p(params, "beam_length")
        g = self.p(params, "gravity_const")
        one = np.ones_like(l_beam)
        smax = np.stack([l_beam / f(2.0), one * f(PI / 4.0), one * f(10.0), one * f(PI)], axis=1)
        amax = (l_beam / f(2.0) * g * f(3.0))[:, None]
        return -smax, smax, -amax, amax

    def init_bounds(self, params, box=0):
        """Two boxes of the CompoundSpace (ball_on_beam.py:60-73); ``box`` in {0, 1}"""
        f = self._f
        l_beam = self.p(params, "beam_length")
        one = np.ones_like(l_beam)
        a = one * f(5 / 180.0 * PI)
        v = one * f(0.02 * 10.0)
        w = one * f(0.02 * PI)
        if box == 0:
            lo = np.stack([f(-0.8) * l_beam / f(2.0), -a, -v, -w], axis=1)
            hi = np.stack([f(-0.7) * l_beam / f(2.0), a, v, w], axis=1)
        else:
            lo = np.stack([f(0.7) * l_beam / f(2.0), -a, -v, -w], axis=1)
            hi = np.stack([f(0.8) * l_beam / f(2.0), a, v, w], axis=1)
        return lo, hi

    def constants(self, params):
        f = self._f
        m_ball, r_ball = self.p(params, "ball_mass"), self.p(params, "ball_radius")
        m_beam, l_beam, d_beam = (self.p(params, n) for n in ("beam_mass", "beam_length", "beam_thickness"))
        J_ball = f(2.0 / 5) * m_ball * r_ball ** 2
        J_beam = f(1.0 / 12) * m_beam * (l_beam ** 2 + d_beam ** 2)
        zeta_ball = m_ball + J_ball / r_ball ** 2
        return J_ball, J_beam, zeta_ball

    def dynamics(self, state, hidden, act, params):
        f = self._f
        g, m_ball = self.p(params, "gravity_const"), self.p(params, "ball_mass")
        c_frict, ang_offset = self.p(params, "friction_coeff"), self.p(params, "ang_offset")
        _, J_beam, zeta_ball = self.constants(params)
        x = state[:, 0]
        a = state[:, 1] + ang_offset
        x_dot = state[:, 2]
        a_dot = state[:, 3]
        zeta_beam = m_ball * x ** 2 + J_beam
        x_ddot = (-c_frict * x_dot + m_ball * x * a_dot ** 2 - m_ball * g * np.sin(a)) / zeta_ball
        a_ddot = (act[:, 0] - f(2.0) * m_ball * x * x_dot * a_dot - m_ball * g * np.cos(a) * x) / zeta_beam
        # symplectic Euler
        v0 = state[:, 2] + x_ddot * f(self.dt)
        v1 = state[:, 3] + a_ddot * f(self.dt)
        p0 = state[:, 0] + v0 * f(self.dt)
        p1 = state[:, 1] + v1 * f(self.dt)
        return np.stack([p0, p1, v0, v1], axis=1), hidden


# -------------------------------------------------------------------------------------------------------------- QQube
class QQubeSwingUpRef(EnvRef):
    """P/environments/pysim/quanser_qube.py:41-188"""

    name = "qq-su"
    S, A, O, H, I = 4, 1, 6, 0, 4
    param_names = ("gravity_const", "motor_resistance", "motor_back_emf", "mass_rot_pole", "length_rot_pole",
                   "damping_rot_pole", "mass_pend_pole", "length_pend_pole", "damping_pend_pole",
                   "voltage_thold_neg", "voltage_thold_pos")
    nominal = (9.81, 8.4, 0.042, 0.095, 0.085, 5e-6, 0.024, 0.129, 1e-6, 0.0, 0.0)
    state_des = np.array([0.0, PI, 0.0, 0.0])
    Qd = (1.0, 1.0, 2e-2, 5e-3)
    Rd = (4e-3,)
    rew_kind = REW_EXP
    radial_idcs = (1,)
    MAX_ACT = 4.5  # P/environments/quanser/__init__.py:34

    def bounds(self, params):
        n = params.shape[0]
        smax = np.tile(np.array([115.0 / 180 * PI, 4 * PI, 20 * PI, 20 * PI], dtype=self.dtype), (n, 1))
        amax = np.full((n, 1), self.MAX_ACT, dtype=self.dtype)
        return -smax, smax, -amax, amax

    def init_bounds(self, params):
        n = params.shape[0]
        hi = np.tile((np.array([2.0, 1.0, 0.5, 0.5]) / 180 * PI).astype(self.dtype), (n, 1))
        return -hi, hi

    def constants(self, params):
        f = self._f
        mr, Lr = self.p(params, "mass_rot_pole"), self.p(params, "length_rot_pole")
        mp, Lp = self.p(params, "mass_pend_pole"), self.p(params, "length_pend_pole")
        g = self.p(params, "gravity_const")
        Jr = mr * Lr ** 2 / f(12)
        Jp = mp * Lp ** 2 / f(12)
        c0 = Jr + mp * Lr ** 2
        c1 = f(0.25) * mp * Lp ** 2
        c2 = f(0.5) * mp * Lp * Lr
        c3 = Jp + c1
        c4 = f(0.5) * mp * Lp * g
        return c0, c1, c2, c3, c4

    def dyn(self, x, u, params):
        """QQubeSim._dyn (quanser_qube.py:89-125)"""
        f = self._f
        km, Rm = self.p(params, "motor_back_emf"), self.p(params, "motor_resistance")
        Dr, Dp = self.p(params, "damping_rot_pole"), self.p(params, "damping_pend_pole")
        c0, c1, c2, c3, c4 = self.constants(params)
        th, al, thd, ald = x[:, 0], x[:, 1], x[:, 2], x[:, 3]
        sin_al = np.sin(al)
        sin_2al = np.sin(f(2) * al)
        a = c0 + c1 * sin_al ** 2
        b = c2 * np.cos(al)
        c = c3
        det = a * c - b * b
        trq = km * (u - km * thd) / Rm
        cc0 = c1 * sin_2al * thd * ald - c2 * sin_al * ald * ald
        cc1 = f(-0.5) * c1 * sin_2al * thd * thd + c4 * sin_al
        xx = trq - Dr * thd - cc0
        yy = -Dp * ald - cc1
        thdd = (c * xx - b * yy) / det
        aldd = (a * yy - b * xx) / det
        return thd, ald, thdd, aldd

    def dynamics(self, state, hidden, act, params):
        f = self._f
        dt = f(self.dt)
        u = act[:, 0].copy()
        dz = (self.p(params, "voltage_thold_neg") <= u) & (u <= self.p(params, "voltage_thold_pos"))
        u[dz] = f(0)
        thd, ald, thdd, aldd = self.dyn(state, u, params)
        # pseudo-RK4 of the reference: _dyn is always evaluated at self.state (quirk Q1, quanser_qube.py:134-146)
        k = [np.stack([thd, ald, thdd, aldd], axis=1)]
        for j in range(1, 4):
            if j <= 2:
                s = state + dt / f(2.0) * k[j - 1]
            else:
                s = state + dt * k[j - 1]
            k.append(np.stack([s[:, 2], s[:, 3], thdd, aldd], axis=1))
        ns = state + dt / f(6) * (k[0] + f(2) * k[1] + f(2) * k[2] + k[3])
        return ns, hidden

    def observe(self, state):
        return np.stack([np.sin(state[:, 0]), np.cos(state[:, 0]), np.sin(state[:, 1]), np.cos(state[:, 1]),
                         state[:, 2], state[:, 3]], axis=1)


# ---------------------------------------------------------------------------------------------------------- QCartPole
class QCartPoleSwingUpRef(EnvRef):
    """P/environments/pysim/quanser_cartpole.py:45-230, 507-587, 591-655.
    flags: long=False, simple_dynamics=False, wild_init='True' (ctor defaults, quanser_cartpole.py:515-524)"""

    name = "qcp-su"
    S, A, O, H, I = 4, 1, 5, 1, 4
    param_names = ("gravity_const", "cart_mass", "rail_length", "motor_efficiency", "gear_efficiency", "gear_ratio",
                   "motor_inertia", "pinion_radius", "motor_resistance", "motor_back_emf", "pole_damping",
                   "combined_damping", "pole_mass", "pole_length", "cart_friction_coeff", "voltage_thold_neg",
                   "voltage_thold_pos")
    nominal = (9.81, 0.58, 0.814, 0.9, 0.9, 3.71, 3.9e-7, 6.35e-3, 2.6, 7.67e-3, 0.0024, 5.4, 0.127, 0.3365 / 2,
               0.02, 0.0, 0.0)
    state_des = np.array([0.0, PI, 0.0, 0.0])
    Qd = (3e-1, 5e-1, 5e-3, 1e-3)
    Rd = (1e-3,)
    rew_kind = REW_EXP
    radial_idcs = (1,)
    MAX_ACT = 6.0  # P/environments/quanser/__init__.py:33
    X_BUFFER = 0.15

    @classmethod
    def nominal_params(cls, n=1, dtype=np.float64, long=False, mass=None, **flags):
        p = list(cls.nominal)
        if long:
            p[cls.param_names.index("pole_mass")] = 0.23
            p[cls.param_names.index("pole_length")] = 0.641 / 2
        if mass is not None:
            p[cls.param_names.index("pole_mass")] = mass
        return np.tile(np.asarray(p, dtype=dtype), (n, 1))

    def bounds(self, params):
        f = self._f
        l_rail = self.p(params, "rail_length")
        one = np.ones_like(l_rail)
        shi = np.stack([l_rail / f(2.0) - f(self.X_BUFFER), one * f(4 * PI), f(1) * l_rail, one * f(20 * PI)], axis=1)
        slo = np.stack([-l_rail / f(2.0) + f(self.X_BUFFER), one * f(-4 * PI), f(-1) * l_rail, one * f(-20 * PI)],
                       axis=1)
        amax = np.full((params.shape[0], 1), self.MAX_ACT, dtype=self.dtype)
        return slo, shi, -amax, amax

    def init_bounds(self, params):
        n = params.shape[0]
        wild = self.flags.get("wild_init", "True")
        if wild == "True":
            hi = np.array([0.25, PI, 0.8, PI])
        elif wild == "False":
            hi = np.array([0.02, 2 / 180.0 * PI, 0.0, 1 / 180.0 * PI])
        else:
            hi = np.array([0.02, PI, 0.0, 1 / 180.0 * PI])
        hi = np.tile(hi.astype(self.dtype), (n, 1))
        return -hi, hi

    def constants(self, params):
        f = self._f
        l_pole, m_pole, m_cart = (self.p(params, n) for n in ("pole_length", "pole_mass", "cart_mass"))
        eta_g, K_g, J_m, r_mp = (self.p(params, n) for n in ("gear_efficiency", "gear_ratio", "motor_inertia",
                                                             "pinion_radius"))
        J_pole = l_pole ** 2 * m_pole / f(3.0)
        J_eq = m_cart + (eta_g * K_g ** 2 * J_m) / r_mp ** 2
        return J_pole, J_eq

    def _f_dyn(self, s_aug, th_ddot_prev, params):
        """QCartPoleSim._dynamics (quanser_cartpole.py:166-230), quirk Q6"""
        f = self._f
        P = lambda n: self.p(params, n)  # noqa: E731
        g, l_p, m_p, m_c = P("gravity_const"), P("pole_length"), P("pole_mass"), P("cart_mass")
        eta_m, eta_g, K_g, R_m, k_m = (P("motor_efficiency"), P("gear_efficiency"), P("gear_ratio"),
                                       P("motor_resistance"), P("motor_back_emf"))
        r_mp, B_eq, B_p, mu_c = P("pinion_radius"), P("combined_damping"), P("pole_damping"), P("cart_friction_coeff")
        J_pole, J_eq = self.constants(params)
        simple = bool(self.flags.get("simple_dynamics", False))

        x, th, x_dot, th_dot, u_in = (s_aug[:, j] for j in range(5))
        sin_th = np.sin(th)
        cos_th = np.cos(th)
        m_tot = m_c + m_p
        u = u_in.copy()
        if not simple:
            dz = (P("voltage_thold_neg") <= u) & (u <= P("voltage_thold_pos"))
            u[dz] = f(0.0)
        f_act = (eta_g * K_g * eta_m * k_m) / (R_m * r_mp) * (eta_m * u - K_g * k_m * x_dot / r_mp)
        if simple:
            f_tot = f_act
        else:
            f_normal = m_tot * g - m_p * l_p / f(2) * (sin_th * th_ddot_prev + cos_th * th_dot ** 2)
            f_c = np.where(f_normal < 0, f(0.0), mu_c * f_normal * np.sign(x_dot))
            f_tot = f_act - f_c
        M00 = m_p + J_eq
        M01 = m_p * l_p * cos_th
        M11 = J_pole + m_p * l_p ** 2
        r0 = f_tot - B_eq * x_dot - m_p * l_p * sin_th * th_dot ** 2
        r1 = -B_p * th_dot - m_p * l_p * g * sin_th
        # np.linalg.solve on the symmetric 2x2 = LAPACK gesv: LU with partial pivoting (M10 == M01)
        swap = np.abs(M01) > np.abs(M00)
        a00 = np.where(swap, M01, M00)
        a01 = np.where(swap, M11, M01)
        a10 = np.where(swap, M00, M01)
        a11 = np.where(swap, M01, M11)
        b0 = np.where(swap, r1, r0)
        b1 = np.where(swap, r0, r1)
        l10 = a10 / a00
        u11 = a11 - l10 * a01
        y1 = b1 - l10 * b0
        th_ddot = y1 / u11
        x_ddot = (b0 - a01 * th_ddot) / a00
        # the returned "position derivative" is the already Euler-advanced velocity; uses self._dt (Q6)
        th_dot_n = th_dot + th_ddot * f(self.dt)
        x_dot_n = x_dot + x_ddot * f(self.dt)
        return np.stack([x_dot_n, th_dot_n, x_ddot, th_ddot, u * f(0)], axis=1), th_ddot

    def dynamics(self, state, hidden, act, params):
        f = self._f
        dt = f(self.dt) - f(0)  # t = [0, dt]
        dt2 = dt / f(2.0)
        y0 = np.concatenate([state, act[:, :1]], axis=1)
        h0 = hidden[:, 0]
        k1, a1 = self._f_dyn(y0, h0, params)
        k2, a2 = self._f_dyn(y0 + dt2 * k1, a1, params)
        k3, a3 = self._f_dyn(y0 + dt2 * k2, a2, params)
        k4, a4 = self._f_dyn(y0 + dt * k3, a3, params)
        y1 = y0 + dt / f(6.0) * (k1 + f(2) * k2 + f(2) * k3 + k4)
        h1 = (a1 + a2 + a3 + a4) / f(4)
        return y1[:, :4], h1[:, None]

    def observe(self, state):
        return np.stack([state[:, 0], np.sin(state[:, 1]), np.cos(state[:, 1]), state[:, 2], state[:, 3]], axis=1)

    def reset_obs(self, state):
        return state.copy()  # QCartPoleSim.reset returns the state, not the observation (Q5, quanser_cartpole.py:109)


# ---------------------------------------------------------------------------------------------------------------- QBB
def qbb_ik_fp32(th, r, l, d=0.10, num_iter=100, lr=0.01, momentum=0.9):
    """QBallBalancerKin.__call__ (quanser_ball_balancer.py:375-444): torch-fp32 SGD(momentum) on the rod-tip position,
    hand-differentiated the way torch autograd evaluates it (pow_backward: g*(2*x); sqrt backward: g/(2*result)).
    Inputs are [N] float64 arrays, the arithmetic is float32; returns float64 [N] (``float(ang)``)."""
    f = np.float32
    th = np.asarray(th, dtype=np.float64).astype(f)
    r = np.asarray(r, dtype=np.float64).astype(f)
    l = np.asarray(l, dtype=np.float64).astype(f)  # noqa: E741
    d = f(d)
    lr, momentum = f(lr), f(momentum)
    t0, t1 = r.copy(), l.copy()  # tip_init = [r, l]
    rc, rs = r * np.cos(th), r * np.sin(th)
    b0 = b1 = None
    for _ in range(num_iter):
        dx, dy = t0 - rc, t1 - rs
        rod = np.sqrt(dx * dx + dy * dy)
        ex, ey = t0 - r - l, t1 - d
        half = np.sqrt(ex * ex + ey * ey)
        g1 = f(2) * (rod - d)
        gu1 = g1 / (f(2) * rod)
        g2 = f(2) * (half - l)
        gu2 = g2 / (f(2) * half)
        gr0 = gu1 * (f(2) * dx) + gu2 * (f(2) * ex)
        gr1 = gu1 * (f(2) * dy) + gu2 * (f(2) * ey)
        if b0 is None:
            b0, b1 = gr0.copy(), gr1.copy()
        else:
            b0, b1 = b0 * momentum + gr0, b1 * momentum + gr1
        t0, t1 = t0 - lr * b0, t1 - lr * b1
    ang = f(PI / 2.0) - np.arctan2(r + l - t0, t1 - d)
    return ang.astype(np.float64)


class QBallBalancerRef(EnvRef):
    """P/environments/pysim/quanser_ball_balancer.py:49-337; flags: simple_dynamics=False"""

    name = "qbb"
    S, A, O, H, I = 8, 2, 8, 2, 4
    param_names = ("gravity_const", "ball_mass", "ball_radius", "plate_length", "arm_radius", "gear_ratio",
                   "gear_efficiency", "load_inertia", "motor_inertia", "motor_back_emf", "motor_resistance",
                   "motor_efficiency", "combined_damping", "ball_damping", "voltage_thold_x_pos",
                   "voltage_thold_x_neg", "voltage_thold_y_pos", "voltage_thold_y_neg", "offset_th_x", "offset_th_y")
    # measured thresholds are never found (dir name mismatch, Q8) -> hard-coded defaults (141-143)
    nominal = (9.81, 0.003, 0.019625, 0.275, 0.0254, 70.0, 0.9, 5.2822e-5, 4.6063e-7, 0.0077, 2.6, 0.69, 0.015, 0.05,
               0.28, -0.10, 0.28, -0.074, 0.0, 0.0)
    state_des = np.zeros(8)
    Qd = (1e0, 1e0, 5e3, 5e3, 1e-2, 1e-2, 5e-1, 5e-1)
    Rd = (1e-2, 1e-2)
    rew_kind = REW_SCALED_EXP
    MAX_ACT = 3.0  # P/environments/quanser/__init__.py:32

    def bounds(self, params):
        f = self._f
        l_plate = self.p(params, "plate_length")
        one = np.ones_like(l_plate)
        smax = np.stack([one * f(PI / 4.0), one * f(PI / 4.0), l_plate / f(2.0), l_plate / f(2.0),
                         one * f(5 * PI), one * f(5 * PI), one * f(0.5), one * f(0.5)], axis=1)
        amax = np.full((params.shape[0], 2), self.MAX_ACT, dtype=self.dtype)
        return -smax, smax, -amax, amax

    def init_bounds(self, params):
        """polar init space [r, phi, x_dot, y_dot] (quanser_ball_balancer.py:108-117)"""
        f = self._f
        l_plate = self.p(params, "plate_length")
        one = np.ones_like(l_plate)
        lo = np.stack([f(0.75) * l_plate / f(2), one * f(-PI), one * f(-0.05 * 0.5), one * f(-0.05 * 0.5)], axis=1)
        hi = np.stack([f(0.8) * l_plate / f(2), one * f(PI), one * f(0.05 * 0.5), one * f(0.05 * 0.5)], axis=1)
        return lo, hi

    @staticmethod
    def polar_to_init(sample):
        """Polar2DPosVelSpace.sample_uniform transform (P/spaces/polar.py:108-113)"""
        out = sample.copy()
        out[:, 0] = sample[:, 0] * np.cos(sample[:, 1])
        out[:, 1] = sample[:, 0] * np.sin(sample[:, 1])
        return out

    def state_from_init(self, init):
        state = np.zeros((init.shape[0], 8), dtype=self.dtype)  # quanser_ball_balancer.py:225-229
        state[:, 2:4] = init[:, :2]
        state[:, 6:8] = init[:, 2:]
        return state

    def init_hidden(self, state, params):
        if self.flags.get("simple_dynamics", False):
            return np.zeros((state.shape[0], 2), dtype=self.dtype)
        r = self.p(params, "arm_radius").astype(np.float64)
        l = (self.p(params, "plate_length") / self._f(2.0)).astype(np.float64)  # noqa: E741
        ax = qbb_ik_fp32(state[:, 0].astype(np.float64) + self.p(params, "offset_th_x"), r, l)
        ay = qbb_ik_fp32(state[:, 1].astype(np.float64) + self.p(params, "offset_th_y"), r, l)
        return np.stack([ax, ay], axis=1).astype(self.dtype)

    def constants(self, params):
        f = self._f
        P = lambda n: self.p(params, n)  # noqa: E731
        l_plate, m_ball, r_ball = P("plate_length"), P("ball_mass"), P("ball_radius")
        eta_g, eta_m, K_g = P("gear_efficiency"), P("motor_efficiency"), P("gear_ratio")
        J_m, J_l, r_arm = P("motor_inertia"), P("load_inertia"), P("arm_radius")
        k_m, R_m, B_eq = P("motor_back_emf"), P("motor_resistance"), P("combined_damping")
        J_ball = f(2.0 / 5) * m_ball * r_ball ** 2
        J_eq = eta_g * K_g ** 2 * J_m + J_l
        c_kin = f(2.0) * r_arm / l_plate
        A_m = eta_g * K_g * eta_m * k_m / R_m
        B_eq_v = eta_g * K_g ** 2 * eta_m * k_m ** 2 / R_m + B_eq
        zeta = m_ball * r_ball ** 2 + J_ball
        return J_ball, J_eq, c_kin, A_m, B_eq_v, zeta

    def dynamics(self, state, hidden, act, params):
        f = self._f
        dt = f(self.dt)
        P = lambda n: self.p(params, n)  # noqa: E731
        g, m_ball, r_ball, ball_damping = P("gravity_const"), P("ball_mass"), P("ball_radius"), P("ball_damping")
        simple = bool(self.flags.get("simple_dynamics", False))
        J_ball, J_eq, c_kin, A_m, B_eq_v, zeta = self.constants(params)
        a0, a1 = act[:, 0].copy(), act[:, 1].copy()
        if not simple:
            a0[(P("voltage_thold_x_neg") <= a0) & (a0 <= P("voltage_thold_x_pos"))] = f(0)
            a1[(P("voltage_thold_y_neg") <= a1) & (a1 <= P("voltage_thold_y_pos"))] = f(0)
        th_x = state[:, 0] + P("offset_th_x")
        th_y = state[:, 1] + P("offset_th_y")
        x, y = state[:, 2], state[:, 3]
        th_x_dot, th_y_dot, x_dot, y_dot = state[:, 4], state[:, 5], state[:, 6], state[:, 7]
        th_x_ddot = (A_m * a0 - B_eq_v * th_x_dot) / J_eq
        th_y_ddot = (A_m * a1 - B_eq_v * th_y_dot) / J_eq
        a, b = hidden[:, 0], hidden[:, 1]
        a_dot = c_kin * th_x_dot * np.cos(th_x) / np.cos(a)
        b_dot = c_kin * -th_y_dot * np.cos(-th_y) / np.cos(b)
        a_ddot = (f(1.0) / np.cos(a)
                  * (c_kin * (th_x_ddot * np.cos(th_x) - th_x_dot ** 2 * np.sin(th_x)) + a_dot ** 2 * np.sin(a)))
        b_ddot = (f(1.0) / np.cos(b)
                  * (c_kin * (-th_y_ddot * np.cos(th_y) - (-th_y_dot) ** 2 * np.sin(-th_y)) + b_dot ** 2 * np.sin(b)))
        if simple:
            x_ddot = c_kin * m_ball * g * r_ball ** 2 * np.sin(th_x) / zeta
            y_ddot = c_kin * m_ball * g * r_ball ** 2 * np.sin(th_y) / zeta
        else:
            x_ddot = (-ball_damping * x_dot * r_ball ** 2 - J_ball * r_ball * a_ddot
                      + m_ball * x * a_dot ** 2 * r_ball ** 2
                      + c_kin * m_ball * g * r_ball ** 2 * np.sin(th_x)) / zeta
            y_ddot = (-ball_damping * y_dot * r_ball ** 2 - J_ball * r_ball * b_ddot
                      + m_ball * y * (-b_dot) ** 2 * r_ball ** 2
                      + c_kin * m_ball * g * r_ball ** 2 * np.sin(th_y)) / zeta
        acc = np.stack([th_x_ddot, th_y_ddot, x_ddot, y_ddot], axis=1)
        vel = state[:, 4:] + acc * dt  # symplectic Euler
        pos = state[:, :4] + vel * dt
        nh = hidden + np.stack([a_dot, b_dot], axis=1) * dt  # forward Euler on the plate angles
        return np.concatenate([pos, vel], axis=1), nh



# ------------------------------------------------------------------------------------------------- further pysim families
class QQubeStabRef(QQubeSwingUpRef):
    """QQubeStabSim, P/environments/pysim/quanser_qube.py:191-222: same dynamics, other init space and weights"""

    name = "qq-st"
    Qd = (3.0, 4.0, 2.0, 2.0)
    Rd = (5e-2,)

    def init_bounds(self, params):
        n = params.shape[0]
        lo = np.tile(np.array([-5.0 / 180 * PI, 175.0 / 180 * PI, 0, 0], dtype=self.dtype), (n, 1))
        hi = np.tile(np.array([5.0 / 180 * PI, 185.0 / 180 * PI, 0, 0], dtype=self.dtype), (n, 1))
        return lo, hi


class QCartPoleStabRef(QCartPoleSwingUpRef):
    """QCartPoleStabSim, P/environments/pysim/quanser_cartpole.py:441-504 (ctor defaults long=True, simple_dynamics=True)"""

    name = "qcp-st"
    Qd = (5e-0, 1e1, 1e-2, 1e-2)
    Rd = (1e-3,)
    rew_kind = REW_QUADR
    final_state_time_dependent = True
    STAB_THOLD = 15 / 180.0 * PI
    MAX_INIT_TH_OFFSET = 8 / 180.0 * PI

    def __init__(self, dt, max_steps, dtype=np.float64, task_args=None, **flags):
        flags.setdefault("long", True)
        flags.setdefault("simple_dynamics", True)
        super().__init__(dt, max_steps, dtype=dtype, task_args=task_args, **flags)

    @classmethod
    def nominal_params(cls, n=1, dtype=np.float64, long=True, mass=None, **flags):
        return super().nominal_params(n, dtype, long=long, mass=mass)

    def bounds(self, params):
        f = self._f
        l_rail = self.p(params, "rail_length")
        one = np.ones_like(l_rail)
        slo = np.stack([-l_rail / f(2.0) + f(self.X_BUFFER), one * f(PI - self.STAB_THOLD), -l_rail, one * f(-2 * PI)], axis=1)
        shi = np.stack([+l_rail / f(2.0) - f(self.X_BUFFER), one * f(PI + self.STAB_THOLD), +l_rail, one * f(+2 * PI)], axis=1)
        amax = np.full((params.shape[0], 1), self.MAX_ACT, dtype=self.dtype)
        return slo, shi, -amax, amax

    def init_bounds(self, params):
        n = params.shape[0]
        hi = np.tile(np.array([+0.02, PI + self.MAX_INIT_TH_OFFSET, +0.02, +5 / 180 * PI], dtype=self.dtype), (n, 1))
        lo = np.tile(np.array([-0.02, PI - self.MAX_INIT_TH_OFFSET, -0.02, -5 / 180 * PI], dtype=self.dtype), (n, 1))
        return lo, hi


class PendulumRef(EnvRef):
    """PendulumSim, P/environments/pysim/pendulum.py:43-117"""

    name = "pend"
    S, A, O, H, I = 2, 1, 3, 0, 2
    param_names = ("gravity_const", "pole_mass", "pole_length", "pole_damping", "torque_thold")
    nominal = (9.81, 1.0, 1.0, 0.05, 3.5)
    state_des = np.array([PI, 0.0])
    Qd = (1e-0, 1e-3)
    Rd = (1e-2,)
    rew_kind = REW_EXP
    radial_idcs = (1,)  # idcs=[1] in the reference: the modulo hits theta_dot (pendulum.py:87)

    def bounds(self, params):
        n = params.shape[0]
        smax = np.tile(np.array([4 * PI, 4 * PI], dtype=self.dtype), (n, 1))
        amax = self.p(params, "torque_thold")[:, None].copy()
        return -smax, smax, -amax, amax

    def init_bounds(self, params):
        n = params.shape[0]
        fixed = np.tile(np.asarray(self.flags.get("init_state", np.zeros(2)), dtype=self.dtype), (n, 1))
        return fixed, fixed.copy()  # SingularStateSpace

    def dynamics(self, state, hidden, act, params):
        f = self._f
        g, m, l, dmp = (self.p(params, k) for k in ("gravity_const", "pole_mass", "pole_length", "pole_damping"))  # noqa: E741
        th, th_dot = state[:, 0], state[:, 1]
        th_ddot = (act[:, 0] - m * g * l / f(2.0) * np.sin(th) - dmp * th_dot) / (m * l ** 2 / f(3.0))
        v = th_dot + th_ddot * f(self.dt)  # symplectic Euler
        p = th + v * f(self.dt)
        return np.stack([p, v], axis=1), hidden

    def observe(self, state):
        return np.stack([np.sin(state[:, 0]), np.cos(state[:, 0]), state[:, 1]], axis=1)


class BallOnBeamDiscRef(BallOnBeamRef):
    """BallOnBeamDiscSim, P/environments/pysim/ball_on_beam.py:139-161: actions are snapped to {-max, 0, +max}
    (DiscreteSpace.project_to, P/spaces/discrete.py:104-131)"""

    name = "bob-d"

    def limit_act(self, act_raw, alo, ahi):
        eles = np.stack([alo[:, 0], (alo[:, 0] + ahi[:, 0]) * self._f(0.5), ahi[:, 0]], axis=1)  # linspace(min, max, 3)
        a = act_raw[:, 0:1]
        close = np.isclose(eles, a).any(axis=1)  # contains(): approximately equal to one of the elements
        idx = np.argmin(np.abs(a - eles), axis=1)
        snapped = eles[np.arange(len(idx)), idx]
        out = np.where(close, a[:, 0], snapped)
        out = np.where(np.isnan(a[:, 0]), a[:, 0], out)
        return out[:, None]


ENV_REFS = OrderedDict((c.name, c) for c in (OneMassOscillatorRef, BallOnBeamRef, QQubeSwingUpRef,
                                              QCartPoleSwingUpRef, QBallBalancerRef, QQubeStabRef, QCartPoleStabRef,
                                              PendulumRef, BallOnBeamDiscRef))


# --------------------------------------------------------------------------------------------------------------------
# wrappers -- P/environment_wrappers/{action_normalization,action_noise,action_delay,observation_normalization,
#             observation_noise,observation_partial}.py, stacked as EnvWrapperAct.step / EnvWrapperObs.step|reset do
#             (P/environment_wrappers/base.py:288-381)
# --------------------------------------------------------------------------------------------------------------------
class WrappedRef:
    """A stack of Pyrado wrappers around an EnvRef, restated sequentially (one stage after the other, as the Python
    objects call each other), for N envs at once.  `stages` lists the wrappers OUTERMOST FIRST:

        ("act_norm",)                      ActNormWrapper._process_act               action_normalization.py:66-72
        ("act_noise", mean[A], std[A])     GaussianActNoiseWrapper._process_act      action_noise.py:71-76
        ("act_delay", delay)               ActDelayWrapper.reset / _process_act      action_delay.py:87-112
        ("obs_norm", lb[O'], ub[O'])       ObsNormWrapper._process_obs               observation_normalization.py:117-120
        ("obs_noise", mean[O'], std[O'])   GaussianObsNoiseWrapper._process_obs      observation_noise.py:67-72
        ("obs_partial", keep_mask[O'])     ObsPartialWrapper._process_obs            observation_partial.py:70-71

    `randn(n, width)` supplies the standard-normal draws in the order the reference would make them: per step the action
    noise wrappers outermost first, then the observation noise wrappers innermost first; per reset the observation noise
    wrappers innermost first."""

    def __init__(self, ref, stages, randn):
        self.ref = ref
        self.stages = list(stages)
        self.randn = randn
        self.queue = None

    def _process_obs(self, obs):
        for st in reversed(self.stages):  # innermost wrapper processes first
            if st[0] == "obs_norm":
                lb, ub = np.asarray(st[1], dtype=float), np.asarray(st[2], dtype=float)
                obs = (obs - lb) / (ub - lb) * 2 - 1
            elif st[0] == "obs_noise":
                mean, std = np.asarray(st[1], dtype=float), np.asarray(st[2], dtype=float)
                obs = obs + (self.randn(obs.shape[0], obs.shape[1]) * std + mean)
            elif st[0] == "obs_partial":
                obs = obs[:, np.asarray(st[1], dtype=bool)]
        return obs

    def reset(self, state):
        """-> the observation the outermost env returns from reset(); `state`: [N, S] the inner env was reset to"""
        n = state.shape[0]
        self.queue = None
        for st in self.stages:
            if st[0] == "act_delay":
                self.queue = [np.zeros((n, self.ref.A)) for _ in range(int(round(st[1])))]
        return self._process_obs(self.ref.reset_obs(np.asarray(state, dtype=self.ref.dtype)))

    def step(self, state, hidden, act, params, curr_step, yielded=None):
        act = np.asarray(act, dtype=float).reshape(state.shape[0], self.ref.A)
        _, _, alo, ahi = self.ref.bounds(np.asarray(params, dtype=self.ref.dtype))
        for st in self.stages:  # outermost wrapper processes first
            if st[0] == "act_norm":
                act = alo + (act + 1) * (ahi - alo) / 2
            elif st[0] == "act_noise":
                act = act + (self.randn(act.shape[0], act.shape[1]) * np.asarray(st[2], dtype=float) + np.asarray(st[1], dtype=float))
            elif st[0] == "act_delay" and int(round(st[1])) != 0:
                self.queue.append(act)
                act = self.queue.pop(0)
        out = self.ref.step(state, hidden, act, params, curr_step, yielded)
        out["act_applied"] = act
        out["obs_inner"] = out["obs"]
        out["obs"] = self._process_obs(out["obs"])
        return out


def make_ref(name, dt, max_steps, dtype=np.float64, task_args=None, **flags):
    return ENV_REFS[name](dt, max_steps, dtype=dtype, task_args=task_args, **flags)
