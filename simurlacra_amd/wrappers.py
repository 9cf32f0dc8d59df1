"""
Environment wrappers on the named path: EnvWrapper (P/environment_wrappers/base.py:45-285),
DomainRandWrapper / DomainRandWrapperLive (P/environment_wrappers/domain_randomization.py:43-148) and the chain helpers
of P/environment_wrappers/utils.py.  Pure delegation, as in the reference.
"""
from typing import Optional

import numpy as np

from .domain_randomization import DomainRandomizer
from .envs import SimEnv
from .exceptions import ShapeErr, TypeErr


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
