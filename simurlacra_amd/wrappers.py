"""
Environment wrappers on the named path: EnvWrapper (P/environment_wrappers/base.py:45-285),
DomainRandWrapper / DomainRandWrapperLive (P/environment_wrappers/domain_randomization.py:43-148) and the chain helpers
of P/environment_wrappers/utils.py.  Pure delegation, as in the reference.
"""
from random import randint
from typing import List, Optional, Union

import numpy as np

from .domain_randomization import DomainRandomizer
from .envs import SimEnv
from .exceptions import ShapeErr, TypeErr, ValueErr


class EnvWrapper:
    def __init__(self, wrapped_env):
        if not isinstance(wrapped_env, (SimEnv, EnvWrapper)):
            raise TypeErr(given=wrapped_env, expected_type=(SimEnv, EnvWrapper))
        self._wrapped_env = wrapped_env

    @property
    def name(self) -> str:
        return self._wrapped_env.name

    @property
    def wrapped_env(self):
        return self._wrapped_env

    @property
    def state_space(self):
        return self._wrapped_env.state_space

    @property
    def obs_space(self):
        return self._wrapped_env.obs_space

    @property
    def act_space(self):
        return self._wrapped_env.act_space

    @property
    def init_space(self):
        return self._wrapped_env.init_space

    @init_space.setter
    def init_space(self, space):
        self._wrapped_env.init_space = space

    @property
    def spec(self):
        from .spaces import EnvSpec

        return EnvSpec(self.obs_space, self.act_space, self.state_space)

    @property
    def dt(self):
        return self._wrapped_env.dt

    @dt.setter
    def dt(self, dt):
        self._wrapped_env.dt = dt

    @property
    def curr_step(self):
        return self._wrapped_env.curr_step

    @property
    def max_steps(self):
        return self._wrapped_env.max_steps

    @max_steps.setter
    def max_steps(self, num_steps):
        self._wrapped_env.max_steps = num_steps

    @property
    def state(self) -> np.ndarray:
        return self._wrapped_env.state.copy()

    @state.setter
    def state(self, state: np.ndarray):
        if not isinstance(state, np.ndarray):
            raise TypeErr(given=state, expected_type=np.ndarray)
        if not state.shape == self._wrapped_env.state.shape:
            raise ShapeErr(given=state, expected_match=self._wrapped_env.state)
        self._wrapped_env.state = state

    @property
    def task(self):
        return self._wrapped_env.task

    @property
    def domain_param(self) -> dict:
        param = self._wrapped_env.domain_param
        self._set_wrapper_domain_param(param)
        return param

    @domain_param.setter
    def domain_param(self, domain_param: dict):
        self._get_wrapper_domain_param(domain_param)
        self._wrapped_env.domain_param = domain_param

    def get_nominal_domain_param(self) -> dict:
        return self._wrapped_env.get_nominal_domain_param()

    @property
    def supported_domain_param(self):
        return self._wrapped_env.supported_domain_param

    @property
    def randomizer(self) -> Optional[DomainRandomizer]:
        return getattr(self._wrapped_env, "randomizer", None)

    @property
    def vec(self):
        """The libvecsim handle of the innermost env (batched access for the GPU sampler)."""
        return inner_env(self).vec

    @property
    def num_envs(self):
        return inner_env(self).num_envs

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        if domain_param is not None:
            self._get_wrapper_domain_param(domain_param)
        return self._wrapped_env.reset(init_state=init_state, domain_param=domain_param)

    def step(self, act: np.ndarray) -> tuple:
        return self._wrapped_env.step(act)

    def observe(self, state):
        return self._wrapped_env.observe(state)

    def limit_act(self, act):
        # Env.limit_act of the wrapper itself (P/environments/base.py:215-222): the projection onto ITS act space --
        # [-1, 1] for ActNormWrapper -- which is what rollout() records as the applied action (rollout.py:244)
        return self.act_space.project_to(act)

    def render(self, mode=None, render_step: int = 1):
        self._wrapped_env.render(mode, render_step)

    def close(self):
        return self._wrapped_env.close()

    def _get_wrapper_domain_param(self, param: dict):
        pass

    def _set_wrapper_domain_param(self, param: dict):
        pass


def all_envs(env):
    yield env
    while isinstance(env, EnvWrapper):
        env = env.wrapped_env
        yield env


def inner_env(env):
    while isinstance(env, EnvWrapper):
        env = env.wrapped_env
    return env


def typed_env(env, tp):
    for e in all_envs(env):
        if isinstance(e, tp):
            return e
    return None


class DomainRandWrapper(EnvWrapper):
    def __init__(self, wrapped_env, randomizer: Optional[DomainRandomizer]):
        if not isinstance(inner_env(wrapped_env), SimEnv):
            raise TypeErr(given=wrapped_env, expected_type=SimEnv)
        if not isinstance(randomizer, DomainRandomizer) and randomizer is not None:
            raise TypeErr(given=randomizer, expected_type=DomainRandomizer)
        super().__init__(wrapped_env)
        self._randomizer = randomizer

    @property
    def randomizer(self) -> DomainRandomizer:
        return self._randomizer

    @randomizer.setter
    def randomizer(self, randomizer: DomainRandomizer):
        if not isinstance(randomizer, DomainRandomizer):
            raise TypeErr(given=randomizer, expected_type=DomainRandomizer)
        self._randomizer = randomizer


class DomainRandWrapperLive(DomainRandWrapper):
    """Randomises the wrapped env at every reset (domain_randomization.py:135-148).

    Single-env use is the reference's: one draw from the torch RNG per reset().  For a batched env
    (`num_envs > 1`, or the GPU sampler) `device_randomization()` hands the same distributions to the kernels, which
    redraw the parameters of a lane at each of its resets (vs_set_randomizer)."""

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        if domain_param is None:
            self._randomizer.randomize(num_samples=1)
            domain_param = self._randomizer.get_params(fmt="dict", dtype="numpy")
        return super().reset(init_state=init_state, domain_param=domain_param)

    def device_randomization(self, on: bool = True):
        self.vec.set_randomizer(self._randomizer.device_specs() if on else [])


class DomainRandWrapperBuffer(DomainRandWrapper):
    """Cycles through (or draws from) a buffer of domain-parameter sets at every reset
    (P/environment_wrappers/domain_randomization.py:151-261)."""

    def __init__(self, wrapped_env, randomizer: Optional[DomainRandomizer], selection: Optional[str] = "cyclic"):
        if selection not in ["cyclic", "random"]:
            raise ValueErr(given=selection, eq_constraint="cyclic or random")
        super().__init__(wrapped_env, randomizer)
        self._ring_idx = None
        self._buffer = None
        self.selection = selection

    @property
    def ring_idx(self) -> int:
        return self._ring_idx

    @ring_idx.setter
    def ring_idx(self, idx: int):
        if not isinstance(idx, int) or not 0 <= idx < len(self._buffer):
            raise ValueErr(given=idx, ge_constraint="0 (int)", l_constraint=len(self._buffer))
        self._ring_idx = idx

    @property
    def selection(self) -> str:
        return self._selection

    @selection.setter
    def selection(self, selection: str):
        if selection not in ["cyclic", "random"]:
            raise ValueErr(given=selection, eq_constraint="cyclic or random")
        self._selection = selection

    def fill_buffer(self, num_domains: int):
        if self._randomizer is None:
            raise TypeErr(msg="The randomizer must not be None to call fill_buffer()!")
        if not isinstance(num_domains, int) or num_domains < 0:
            raise ValueErr(given=num_domains, g_constraint="0 (int)")
        self._randomizer.randomize(num_domains)
        self._buffer = self._randomizer.get_params(-1, fmt="list", dtype="numpy")
        self._ring_idx = 0

    @property
    def buffer(self):
        return self._buffer

    @buffer.setter
    def buffer(self, buffer: Union[List[dict], dict]):
        if not (isinstance(buffer, list) or isinstance(buffer, dict)):
            raise TypeErr(given=buffer, expected_type=[list, dict])
        self._buffer = buffer

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        if domain_param is None:
            if isinstance(self._buffer, dict):
                domain_param = self._buffer
            elif isinstance(self._buffer, list):
                domain_param = self._buffer[self._ring_idx]
                if self._selection == "cyclic":
                    self._ring_idx = (self._ring_idx + 1) % len(self._buffer)
                elif self._selection == "random":
                    self._ring_idx = randint(0, len(self._buffer) - 1)
            else:
                raise TypeErr(given=self._buffer, expected_type=[dict, list])
        return super().reset(init_state=init_state, domain_param=domain_param)

    def device_buffer(self, on: bool = True):
        """Hand the buffer to the kernels (vs_set_param_buffer): every lane walks the ring at its own resets."""
        buf = [self._buffer] if isinstance(self._buffer, dict) else self._buffer
        self.vec.set_param_buffer(buf if on else None, self._selection)


class EnvWrapperAct(EnvWrapper):
    """Base of the wrappers that modify the action (P/environment_wrappers/base.py:288-330)."""

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    def _process_act_space(self, space):
        return space

    def step(self, act: np.ndarray) -> tuple:
        return self._wrapped_env.step(self._process_act(act))

    @property
    def act_space(self):
        return self._process_act_space(self._wrapped_env.act_space)


class ActNormWrapper(EnvWrapperAct):
    """Normalises the action space to [-1, 1] (P/environment_wrappers/action_normalization.py:63-89).

    On an env object the de-normalisation happens here on the host, exactly as in the reference; `device_fuse()` moves it
    into the step kernels (VS_FLAG_ACT_NORM) for batched use, after which `vec.step` / `vec.step_random` take and draw
    normalised actions."""

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        lb, ub = self.wrapped_env.act_space.bounds
        return lb + (act + 1) * (ub - lb) / 2

    def _process_act_space(self, space):
        from .spaces import BoxSpace

        if not isinstance(space, BoxSpace):
            raise NotImplementedError("Only implemented ActNormWrapper._process_act_space() for BoxSpace!")
        return BoxSpace(-np.ones(space.shape), np.ones(space.shape), labels=space.labels)

    def device_fuse(self, on: bool = True):
        self.vec.set_act_norm(on)


class EnvWrapperObs(EnvWrapper):
    """Base of the wrappers that modify the observation (P/environment_wrappers/base.py:333-381)."""

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    def _process_obs_space(self, space):
        return space

    @property
    def obs_space(self):
        return self._process_obs_space(self._wrapped_env.obs_space)

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        return self._process_obs(super().reset(init_state=init_state, domain_param=domain_param))

    def step(self, act: np.ndarray) -> tuple:
        obs, rew, done, info = super().step(act)
        return self._process_obs(obs), rew, done, info


class GaussianActNoiseWrapper(EnvWrapperAct):
    """act + randn * std + mean (P/environment_wrappers/action_noise.py:38-79); NumPy global RNG on an env object,
    Philox on the device once `fuse_wrappers` has moved it into the kernels."""

    def __init__(self, wrapped_env, noise_mean=None, noise_std=None):
        super().__init__(wrapped_env)
        shape = self.act_space.shape
        self._mean = np.zeros(shape) if noise_mean is None else np.array(noise_mean)
        self._std = np.zeros(shape) if noise_std is None else np.array(noise_std)
        for v in (self._mean, self._std):
            if not v.shape == shape:
                raise ShapeErr(given=v, expected_match=self.act_space)

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        noise = np.random.randn(*self.act_space.shape) * self._std + self._mean
        return act + noise

    def _set_wrapper_domain_param(self, domain_param: dict):
        domain_param["act_noise_mean"] = self._mean
        domain_param["act_noise_std"] = self._std

    def _get_wrapper_domain_param(self, domain_param: dict):
        # the reference stores these under other attribute names than _process_act reads (action_noise.py:92-101), so a
        # domain-param update never changes the noise that is applied; kept
        if "act_noise_mean" in domain_param:
            self._noise_mean = np.array(domain_param["act_noise_mean"])
            if not self._noise_mean.shape == self.act_space.shape:
                raise ShapeErr(given=self._noise_mean, expected_match=self.act_space)
        if "act_noise_std" in domain_param:
            self._noise_std = np.array(domain_param["act_noise_std"])
            if not self._noise_std.shape == self.act_space.shape:
                raise ShapeErr(given=self._noise_std, expected_match=self.act_space)


class ActDelayWrapper(EnvWrapperAct):
    """Delays the actions by a fixed number of steps; the queue starts as `delay` zero actions at every reset
    (P/environment_wrappers/action_delay.py:37-112)."""

    def __init__(self, wrapped_env, delay: int = 0):
        super().__init__(wrapped_env)
        self._delay = delay
        self._act_queue = []

    @property
    def delay(self) -> int:
        if isinstance(self._delay, np.ndarray):
            return np.round(self._delay)
        return round(self._delay)

    @delay.setter
    def delay(self, delay: int):
        if not delay >= 0:
            raise ValueErr(given=delay, ge_constraint="0")
        self._delay = round(delay)

    def _set_wrapper_domain_param(self, domain_param: dict):
        domain_param["act_delay"] = self._delay

    def _get_wrapper_domain_param(self, domain_param: dict):
        self._delay = domain_param.get("act_delay", self._delay)

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        init_obs = super().reset(init_state=init_state, domain_param=domain_param)
        self._act_queue = [np.zeros(self.act_space.shape)] * int(self.delay)
        return init_obs

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        if self.delay != 0:
            self._act_queue.append(act)
            act = self._act_queue.pop(0)
        return act


class GaussianObsNoiseWrapper(EnvWrapperObs):
    """obs + randn * std + mean (P/environment_wrappers/observation_noise.py:38-95)"""

    def __init__(self, wrapped_env, noise_std, noise_mean=None):
        super().__init__(wrapped_env)
        self._std = np.array(noise_std)
        if not self._std.shape == self.obs_space.shape:
            raise ShapeErr(given=self._std, expected_match=self.obs_space)
        if noise_mean is not None:
            self._mean = np.array(noise_mean)
            if not self._mean.shape == self.obs_space.shape:
                raise ShapeErr(given=self._mean, expected_match=self.obs_space)
        else:
            self._mean = np.zeros(self.obs_space.shape)

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        noise = np.random.randn(*self.obs_space.shape) * self._std + self._mean
        return obs + noise

    def _set_wrapper_domain_param(self, domain_param: dict):
        domain_param["obs_noise_mean"] = self._mean
        domain_param["obs_noise_std"] = self._std

    def _get_wrapper_domain_param(self, domain_param: dict):
        if "obs_noise_mean" in domain_param:
            self._mean = np.array(domain_param["obs_noise_mean"])
            assert self._mean.shape == self.obs_space.shape
        if "obs_noise_std" in domain_param:
            self._std = np.array(domain_param["obs_noise_std"])
            assert self._std.shape == self.obs_space.shape


class ObsNormWrapper(EnvWrapperObs):
    """Maps the observation box to [-1, 1]: (obs - lb) / (ub - lb) * 2 - 1, where infinite bounds of the wrapped space must
    be replaced through `explicit_lb` / `explicit_ub` ({label: bound}) (P/environment_wrappers/observation_normalization.py:
    41-126).  Error behaviour as there: an override dict that leaves an infinite entry untouched, or a bound that is still
    infinite afterwards, raises ValueErr."""

    def __init__(self, wrapped_env, explicit_lb=None, explicit_ub=None):
        super().__init__(wrapped_env)
        self.explicit_lb, self.explicit_ub = explicit_lb, explicit_ub
        space = self.wrapped_env.obs_space
        lo, up = space.bounds
        self.ov_lb = self.override_bounds(lo, explicit_lb, space.labels)
        self.ov_ub = self.override_bounds(up, explicit_ub, space.labels)
        for which, vec, bad in (("lower bounds is (negative)", self.ov_lb, -np.inf), ("upper bound is (positive)", self.ov_ub, np.inf)):
            if np.any(vec == bad):
                raise ValueErr(msg=f"At least one element of the {which} infinite:\n(overwritten) bound: {vec}\n"
                                   f"names: {space.labels}")

    @staticmethod
    def override_bounds(bounds: np.ndarray, override, names: np.ndarray) -> np.ndarray:
        """a copy of `bounds` with the labelled entries of `override` put in; without an override dict the array itself"""
        if not override:
            return bounds
        out = np.array(bounds, dtype=float, copy=True)
        given = np.array([override.get(lab) is not None for lab in names.ravel()]).reshape(names.shape)
        loose = np.isinf(out) & ~given
        if loose.any():
            raise ValueErr(msg=f"The entry {names[loose].ravel()[0]} of a bound is infinite and not overwritten. "
                               f"Cannot apply normalization!")
        out[given] = [override[lab] for lab in names[given].ravel()]
        return out

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        span = self.ov_ub - self.ov_lb
        return (obs - self.ov_lb) / span * 2 - 1

    def _process_obs_space(self, space):
        from .spaces import BoxSpace

        if not isinstance(space, BoxSpace):
            raise NotImplementedError("Only implemented ObsNormWrapper._process_obs_space() for BoxSpace!")
        one = np.ones(space.shape)
        return BoxSpace(-one, one, labels=space.labels)


class ObsPartialWrapper(EnvWrapperObs):
    """Drops (or keeps) selected observation entries (P/environment_wrappers/observation_partial.py:36-75)"""

    def __init__(self, wrapped_env, mask: list = None, idcs: list = None, keep_selected: bool = False):
        super().__init__(wrapped_env)
        if mask is not None:
            mask = np.array(mask, dtype=bool)
            if not mask.shape == wrapped_env.obs_space.shape:
                raise ShapeErr(given=mask, expected_match=wrapped_env.obs_space)
        else:
            assert idcs is not None, "Either mask or indices must be specified"
            mask = wrapped_env.obs_space.create_mask(idcs)
        self.keep_mask = mask if keep_selected else np.logical_not(mask)

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        return obs[self.keep_mask]

    def _process_obs_space(self, space):
        return space.subspace(self.keep_mask)


# ---------------------------------------------------------------------------------------------- chain -> kernel pipeline
class FusedChain:
    """What a stack of wrappers around a pysim env amounts to, in the terms of vs_set_act_norm / vs_set_act_pipeline /
    vs_set_obs_pipeline.  `keep` selects the rows of VS_OBS the outermost env reports (ObsPartialWrapper)."""

    def __init__(self, O, A):
        self.act_norm = False
        self.delay = 0
        self.noise_mean, self.noise_std = np.zeros(A), np.zeros(A)
        self.noise_normed = self.noise_after_delay = False
        self.scale, self.shift, self.var = np.ones(O), np.zeros(O), np.zeros(O)
        self.keep = np.ones(O, dtype=bool)

    @property
    def obs_std(self):
        return np.sqrt(self.var)

    def apply(self, vec, seed=0):
        vec.set_act_norm(self.act_norm)
        vec.set_act_pipeline(self.delay, self.noise_mean, self.noise_std, self.noise_normed, self.noise_after_delay,
                             seed=seed ^ 0x2545F4914F6CDD1D)
        vec.set_obs_pipeline(self.scale, self.shift, self.obs_std, seed=seed ^ 0x9E3779B97F4A7C15)


def fuse_wrappers(env) -> FusedChain:
    """Walk the wrapper chain of `env` and express it as the kernels' fixed pipeline.

    Action side (outermost wrapper acts first):  ActNormWrapper, GaussianActNoiseWrapper, ActDelayWrapper in any order, at
    most one of each.  A delay wrapper commutes with the normalisation (the queue's zero action is the centre of the
    symmetric action boxes of these envs); a noise wrapper outside the normalisation draws in [-1, 1] units
    (`noise_normed`), one inside the delay adds to the delayed action (`noise_after_delay`).
    Observation side (innermost acts first): any stack of ObsNormWrapper / GaussianObsNoiseWrapper / ObsPartialWrapper;
    affine stages compose, independent Gaussian stages add their variances.
    Raises NotImplementedError for wrappers the kernels do not express."""
    base = inner_env(env)
    O, A = base.obs_space.flat_dim, base.act_space.flat_dim
    fc = FusedChain(O, A)
    chain = [e for e in all_envs(env) if isinstance(e, EnvWrapper)]  # outermost first
    seen_norm = seen_delay = seen_noise = False
    for w in chain:  # action side: outermost first
        if isinstance(w, ActNormWrapper):
            if seen_norm:
                raise NotImplementedError("two ActNormWrappers in one chain")
            seen_norm = fc.act_norm = True
        elif isinstance(w, ActDelayWrapper):
            if seen_delay:
                raise NotImplementedError("two ActDelayWrappers in one chain")
            seen_delay = True
            fc.delay = int(w.delay)
        elif isinstance(w, GaussianActNoiseWrapper):
            if seen_noise:
                raise NotImplementedError("two GaussianActNoiseWrappers in one chain")
            seen_noise = True
            fc.noise_mean, fc.noise_std = w._mean.astype(np.float64), w._std.astype(np.float64)
            fc.noise_normed = not seen_norm and any(isinstance(x, ActNormWrapper) for x in all_envs(w.wrapped_env))
            fc.noise_after_delay = seen_delay
        elif isinstance(w, EnvWrapperAct):
            raise NotImplementedError(f"{type(w).__name__} is not fused into the kernels")
    idx = np.arange(O)  # rows of VS_OBS that are still visible
    for w in reversed(chain):  # observation side: innermost first
        if isinstance(w, ObsPartialWrapper):
            idx = idx[w.keep_mask]
        elif isinstance(w, ObsNormWrapper):
            k = 2.0 / (w.ov_ub - w.ov_lb)
            b = -w.ov_lb * k - 1.0
            fc.scale[idx] *= k
            fc.shift[idx] = fc.shift[idx] * k + b
            fc.var[idx] *= k * k
        elif isinstance(w, GaussianObsNoiseWrapper):
            fc.shift[idx] += w._mean
            fc.var[idx] += w._std ** 2
        elif isinstance(w, EnvWrapperObs):
            raise NotImplementedError(f"{type(w).__name__} is not fused into the kernels")
    fc.keep = np.zeros(O, dtype=bool)
    fc.keep[idx] = True
    return fc
