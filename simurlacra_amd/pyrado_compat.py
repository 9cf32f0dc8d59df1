"""
Optional glue for running next to a real Pyrado installation: registers the classes of this package as virtual
subclasses of Pyrado's ABCs so that `isinstance(inner_env(env), pyrado.environments.sim_base.SimEnv)` checks in Pyrado's
own wrappers and rollout() (environment_wrappers/domain_randomization.py:56-57, sampling/rollout.py:148) accept them.
Pyrado does not travel to the GPU box; nothing else in this package imports it.
"""


def register_with_pyrado() -> bool:
    """Returns True when Pyrado was importable and the registration happened."""
    try:
        from pyrado.environment_wrappers.base import EnvWrapper as PyradoEnvWrapper
        from pyrado.environments.base import Env as PyradoEnv
        from pyrado.environments.sim_base import SimEnv as PyradoSimEnv
    except Exception:
        return False
    from .envs import VecSimPyEnv
    from .wrappers import EnvWrapper

    PyradoSimEnv.register(VecSimPyEnv)
    PyradoEnv.register(VecSimPyEnv)
    PyradoEnv.register(EnvWrapper)
    PyradoEnvWrapper.register(EnvWrapper)
    return True


def to_pyrado_step_sequence(ro):
    """One rollout of this package's sampler as Pyrado's own `StepSequence` (P/sampling/step_sequence.py:223-362), so that
    Pyrado's algorithms (`StepSequence.concat`, `gae_returns`, `split_shuffled_batches`, ...) consume GPU rollouts
    unchanged.  Needs Pyrado importable; raises ImportError otherwise."""
    from pyrado.sampling.step_sequence import StepSequence as PyradoStepSequence

    extra = {}
    if getattr(ro, "states", None) is not None:
        extra["states"] = ro.states
    # the fields rollout() adds besides observations / actions / rewards (P/sampling/rollout.py:305-325): the applied
    # actions and, in the fork, the cartpole's hidden pole acceleration
    for field in ("actions_applied", "th_ddot"):
        if getattr(ro, field, None) is not None:
            extra[field] = getattr(ro, field)
    if getattr(ro, "time", None) is not None:
        extra["time"] = ro.time
    return PyradoStepSequence(observations=ro.observations, actions=ro.actions, rewards=ro.rewards,
                              rollout_info=ro.rollout_info, complete=ro.complete, **extra)


def to_pyrado_step_sequences(ros) -> list:
    return [to_pyrado_step_sequence(ro) for ro in ros]
