"""
simurlacra_amd -- MI355X-native vectorised stepper for Pyrado's pure-Python simulated robots.

The hot path (SimPyEnv.step / reset / observe, DomainRandWrapperLive, random-policy rollouts) runs as hand-written HIP
kernels in ``csrc/libvecsim.so`` behind the C-ABI of ``include/vecsim.h``; this package is the host-side mirror of the
reference's Python interface for that path (same class names, arguments and error behaviour).  Nothing in here falls
back to a CPU implementation: without the built library or without a GPU the constructors raise.
"""
from . import _lib  # noqa: F401
from .domain_randomization import (BernoulliDomainParam, DomainParam, DomainRandomizer,  # noqa: F401
                                   MultivariateNormalDomainParam, NormalDomainParam, UniformDomainParam,
                                   create_conservative_randomizer, create_default_randomizer,
                                   create_zero_var_randomizer)
from .envs import (ENV_CLASSES, BallOnBeamDiscSim, BallOnBeamSim, OneMassOscillatorSim, PendulumSim,  # noqa: F401
                   QBallBalancerSim, QCartPoleStabSim, QCartPoleSwingUpSim, QQubeStabSim, QQubeSwingUpSim, SimEnv,
                   VecSimPyEnv)
from .exceptions import KeyErr, ShapeErr, TypeErr, ValueErr  # noqa: F401
from .seeding import derive_seed, get_base_seed, set_seed  # noqa: F401
from .spaces import BoxSpace, CompoundSpace, DiscreteSpace, EnvSpec, Polar2DPosVelSpace, SingularStateSpace  # noqa: F401
from .vec_env import MixedVecSimEnv, VecSimEnv, env_dims, nominal_params, param_names  # noqa: F401
from .wrappers import (ActDelayWrapper, ActNormWrapper, DomainRandWrapper, DomainRandWrapperBuffer,  # noqa: F401
                       DomainRandWrapperLive, EnvWrapper, EnvWrapperAct, EnvWrapperObs, FusedChain,
                       GaussianActNoiseWrapper, GaussianObsNoiseWrapper, ObsNormWrapper, ObsPartialWrapper, all_envs,
                       fuse_wrappers, inner_env, typed_env)

inf = float("inf")


def __getattr__(name):
    # torch-dependent pieces are imported on first use
    if name in ("ParallelRolloutSampler", "StepSequence", "PackedRollouts", "rollout", "CVaRSampler", "select_cvar"):
        from . import sampling

        return getattr(sampling, name)
    if name in ("DummyPolicy", "IdlePolicy", "Policy", "FNN", "FNNPolicy", "NormalActNoiseExplStrat", "fnn_kernel_spec"):
        from . import policies

        return getattr(policies, name)
    raise AttributeError(name)
