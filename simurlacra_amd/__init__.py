"""
simurlacra_amd -- MI355X-native vectorised stepper for Pyrado's pure-Python simulated robots.

The hot path (SimPyEnv.step / reset / observe, DomainRandWrapperLive, random-policy rollouts) runs as hand-written HIP
kernels in ``csrc/libvecsim.so`` behind the C-ABI of ``include/vecsim.h``; this package is the host-side mirror of the
reference's Python interface for that path.  Nothing in here falls back to a CPU implementation.
"""
from . import _lib  # noqa: F401
from .exceptions import KeyErr, ShapeErr, TypeErr, ValueErr  # noqa: F401
from .vec_env import VecSimEnv, env_dims, nominal_params, param_names  # noqa: F401

inf = float("inf")
