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
        return self._wrapped_env.limit_act(act)

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
