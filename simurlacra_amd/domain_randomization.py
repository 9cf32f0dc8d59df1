"""
Host-side mirror of Pyrado's domain randomisation objects for the five pysim envs:
DomainParam / NormalDomainParam / UniformDomainParam (P/domain_randomization/domain_parameter.py:43-203),
DomainRandomizer (P/domain_randomization/domain_randomizer.py:48-290) and the default randomizers
(P/domain_randomization/default_randomizers.py:163-416).

Two ways to draw: `DomainRandomizer.randomize(n)` draws on the host from torch.distributions with the torch global RNG,
exactly like the reference (fp32 values, quirk Q13) -- used by the single-env `DomainRandWrapperLive.reset()`;
`device_specs()` hands the same distributions to libvecsim (vs_sample_params / vs_set_randomizer), which draws per lane
on the GPU with Philox.
"""
from copy import deepcopy
from typing import List, Optional, Union

import numpy as np

from .exceptions import KeyErr, ShapeErr, TypeErr, ValueErr

inf = float("inf")


class DomainParam:
    def __init__(self, name: str, clip_lo: float = -inf, clip_up: float = inf, roundint: bool = False):
        if not isinstance(name, (str, list)):
            raise TypeErr(given=name, expected_type=(str, list))
        self.name = name
        self.mean = None
        self.clip_lo = clip_lo
        self.clip_up = clip_up
        self.roundint = roundint
        self.distr = None

    def get_field_names(self) -> List[str]:
        return ["name", "clip_lo", "clip_up", "roundint"]

    def __eq__(self, other):
        if not isinstance(other, DomainParam):
            raise TypeErr(given=other, expected_type=DomainParam)
        return all(getattr(self, fn) == getattr(other, fn, None) for fn in self.get_field_names())

    def adapt(self, domain_distr_param: str, domain_distr_param_value):
        if domain_distr_param not in self.get_field_names():
            raise KeyErr(msg=f"The domain parameter {self.name} does not have a domain distribution parameter "
                             f"called {domain_distr_param}!")
        setattr(self, domain_distr_param, domain_distr_param_value)
        self._make_distr()

    def _make_distr(self):
        pass

    def sample(self, num_samples: int = 1) -> list:
        """distr.sample -> clamp -> (round); list of 0-d torch tensors (domain_parameter.py:104-132)"""
        import torch

        if not isinstance(num_samples, int):
            raise TypeErr(given=num_samples, expected_type=int)
        if num_samples <= 0:
            raise ValueErr(given=num_samples, g_constraint="0")
        if self.distr is None:
            raise RuntimeError("Trying to sample a domain parameter without a specified distribution!")
        t = self.distr.sample(sample_shape=torch.Size([num_samples]))
        t = torch.clamp(t, self.clip_lo, self.clip_up)
        if self.roundint:
            t = torch.round(t).to(torch.int32)
        return list(t)


class UniformDomainParam(DomainParam):
    def __init__(self, mean, halfspan, **kwargs):
        super().__init__(**kwargs)
        self.mean = mean
        self.halfspan = halfspan
        self._make_distr()

    def _make_distr(self):
        from torch.distributions.uniform import Uniform

        self.distr = Uniform(self.mean - self.halfspan, self.mean + self.halfspan, validate_args=True)

    def get_field_names(self):
        return ["name", "mean", "halfspan", "clip_lo", "clip_up", "roundint"]


class NormalDomainParam(DomainParam):
    def __init__(self, mean, std, **kwargs):
        super().__init__(**kwargs)
        self.mean = mean
        self.std = std
        self._make_distr()

    def _make_distr(self):
        from torch.distributions.normal import Normal

        self.distr = Normal(self.mean, self.std, validate_args=True)

    def get_field_names(self):
        return ["name", "mean", "std", "clip_lo", "clip_up", "roundint"]


class MultivariateNormalDomainParam(DomainParam):
    """domain_parameter.py:206-245.  One name, a vector-valued sample: for the scalar parameters of the pysim envs only
    dimension 1 is meaningful (and is what `device_specs` accepts)."""

    def __init__(self, mean, cov, **kwargs):
        import torch

        super().__init__(**kwargs)
        self.mean = torch.as_tensor(mean, dtype=torch.get_default_dtype()).view(-1)
        self.cov = torch.as_tensor(cov, dtype=torch.get_default_dtype())
        if not self.cov.ndim == 2:
            raise ShapeErr(msg="The covariance needs to be given as a matrix!")
        self._make_distr()

    def _make_distr(self):
        from torch.distributions.multivariate_normal import MultivariateNormal

        self.distr = MultivariateNormal(self.mean, self.cov, validate_args=True)

    def get_field_names(self):
        return ["name", "mean", "cov", "clip_lo", "clip_up", "roundint"]

    def adapt(self, domain_distr_param: str, domain_distr_param_value):
        if domain_distr_param == "cov" and domain_distr_param_value < 0:
            raise ValueErr(given_name="cov", ge_constraint="0")
        super().adapt(domain_distr_param, domain_distr_param_value)


class BernoulliDomainParam(DomainParam):
    """val_1 with probability prob_1, else val_0 (domain_parameter.py:248-311)"""

    def __init__(self, val_0, val_1, prob_1: float, **kwargs):
        super().__init__(**kwargs)
        self.val_0 = val_0
        self.val_1 = val_1
        self.prob_1 = prob_1
        self.mean = val_0 * (1 - prob_1) + val_1 * prob_1
        self._make_distr()

    def _make_distr(self):
        from torch.distributions.bernoulli import Bernoulli

        self.distr = Bernoulli(self.prob_1, validate_args=True)

    def get_field_names(self):
        return ["name", "mean", "val_0", "val_1", "prob_1", "clip_lo", "clip_up", "roundint"]

    def sample(self, num_samples: int = 1) -> list:
        import torch

        if not isinstance(num_samples, int):
            raise TypeErr(given=num_samples, expected_type=int)
        if num_samples <= 0:
            raise ValueErr(given=num_samples, g_constraint="0")
        if self.distr is None:
            raise RuntimeError("Trying to sample a domain parameter without a specified distribution!")
        t = self.distr.sample(sample_shape=torch.Size([num_samples]))
        t = (torch.ones_like(t) - t) * self.val_0 + t * self.val_1
        t = torch.clamp(t, self.clip_lo, self.clip_up)
        if self.roundint:
            t = torch.round(t).to(torch.int32)
        return list(t)


class DomainRandomizer:
    def __init__(self, *domain_params: DomainParam):
        self.domain_params = []
        self.add_domain_params(*domain_params)
        self._params_pert_dict = None
        self._params_pert_list = None

    def add_domain_params(self, *domain_params: DomainParam):
        for dp in domain_params:
            if not isinstance(dp, DomainParam):
                raise TypeErr(given=dp, expected_type=DomainParam)
            self.domain_params.append(dp)

    def randomize(self, num_samples: int):
        """domain_randomizer.py:123-157"""
        if not isinstance(num_samples, int):
            raise TypeErr(given=num_samples, expected_type=int)
        if num_samples <= 0:
            raise ValueErr(given=num_samples, g_constraint="0")
        keys = [dp.name for dp in self.domain_params]
        values = [dp.sample(num_samples) for dp in self.domain_params]
        self._params_pert_dict = dict(zip(keys, values))
        self._params_pert_list = [{k: v[i] for k, v in zip(keys, values)} for i in range(num_samples)]

    def get_params(self, num_samples: int = -1, fmt: str = "list", dtype: str = "numpy") -> Union[list, dict]:
        """domain_randomizer.py:159-227: list of dicts / dict of lists; 'numpy' gives 0-d float32 arrays (Q13)"""
        if not isinstance(num_samples, int):
            raise TypeErr(given=num_samples, expected_type=int)
        if num_samples <= -2 or num_samples == 0:
            raise ValueErr(msg="The number of samples needs to be -1 or a positive integer!")
        if fmt.lower() not in ("list", "dict"):
            raise ValueErr(given=fmt, eq_constraint="list or dict")
        if dtype.lower() not in ("numpy", "torch"):
            raise ValueErr(given=dtype, eq_constraint="numpy or torch")
        if self._params_pert_list is None:
            raise RuntimeError("randomize() must be called before get_params()")
        conv = (lambda t: t.detach().numpy()) if dtype == "numpy" else (lambda t: t.clone())
        have = len(self._params_pert_list)
        if num_samples == 1 or have == 1:
            out = {k: conv(v) for k, v in self._params_pert_list[0].items()}
            return [out] if fmt == "list" else out
        n = have if num_samples == -1 else min(num_samples, have)
        if fmt == "list":
            return [{k: conv(v) for k, v in d.items()} for d in self._params_pert_list[:n]]
        return {k: [conv(x) for x in v[:n]] for k, v in self._params_pert_dict.items()}

    def adapt_one_distr_param(self, domain_param_name: str, domain_distr_param: str, value):
        for dp in self.domain_params:
            if dp.name == domain_param_name:
                dp.adapt(domain_distr_param, value)
                return
        raise KeyErr(msg=f"No domain parameter called {domain_param_name}")

    def rescale_distr_param(self, param: str, scale: float):
        if not scale >= 0:
            raise ValueErr(given=scale, ge_constraint="0")
        for dp in self.domain_params:
            if param in dp.get_field_names():
                dp.adapt(param, scale * getattr(dp, param))

    def get_subset(self, names: List[str]) -> "DomainRandomizer":
        return DomainRandomizer(*[deepcopy(dp) for dp in self.domain_params if dp.name in names])

    def device_specs(self) -> list:
        """[(name, kind, mean, spread, clip_lo, clip_up, aux, roundint)] for vs_sample_params / vs_set_randomizer
        (kinds and field meanings: include/vecsim.h, vs_dp_spec)"""
        out = []
        for dp in self.domain_params:
            tail = (float(dp.clip_lo), float(dp.clip_up))
            if isinstance(dp, NormalDomainParam):
                out.append((dp.name, "normal", float(dp.mean), float(dp.std)) + tail + (0.0, bool(dp.roundint)))
            elif isinstance(dp, UniformDomainParam):
                out.append((dp.name, "uniform", float(dp.mean), float(dp.halfspan)) + tail + (0.0, bool(dp.roundint)))
            elif isinstance(dp, BernoulliDomainParam):
                out.append((dp.name, "bernoulli", float(dp.val_0), float(dp.val_1)) + tail + (float(dp.prob_1), bool(dp.roundint)))
            elif isinstance(dp, MultivariateNormalDomainParam):
                if dp.mean.numel() != 1:
                    raise NotImplementedError("a MultivariateNormalDomainParam of dimension > 1 puts a vector under one "
                                              "parameter name; the pysim envs only have scalar parameters")
                out.append((dp.name, "normal", float(dp.mean[0]), float(dp.cov[0, 0]) ** 0.5) + tail + (0.0, bool(dp.roundint)))
            else:
                raise TypeErr(given=dp, expected_type=(NormalDomainParam, UniformDomainParam, BernoulliDomainParam,
                                                       MultivariateNormalDomainParam))
        return out


# ---------------------------------------------------------------------------------------------------- default randomizers
# (name, kind, spread as a divisor of |nominal| or an absolute number, clip_lo, clip_up); nominal = the env's
# get_nominal_domain_param().  N = Normal(mean=nominal, std=|nominal|/div), U = Uniform(mean=nominal, halfspan=...)
_DEG = np.pi / 180
_DEFAULTS = {
    # default_randomizers.py:163-189
    "bob": [("gravity_const", "N", 10, 1e-4, inf), ("ball_mass", "N", 5, 1e-4, inf), ("ball_radius", "N", 5, 1e-4, inf),
            ("beam_mass", "N", 5, 1e-3, inf), ("beam_length", "N", 5, 1e-3, inf), ("beam_thickness", "N", 5, 1e-3, inf),
            ("friction_coeff", "U", 1, 0, inf), ("ang_offset", "Uabs", 0.1 * _DEG, -inf, inf)],
    # :192-206
    "omo": [("mass", "N", 3, 1e-3, inf), ("stiffness", "N", 3, 1e-3, inf), ("damping", "N", 3, 1e-3, inf)],
    # :234-300
    "qbb": [("gravity_const", "N", 10, 1e-4, inf), ("ball_mass", "N", 5, 1e-4, inf), ("ball_radius", "N", 5, 1e-3, inf),
            ("plate_length", "N", 5, 5e-2, inf), ("arm_radius", "N", 5, 1e-4, inf), ("gear_ratio", "N", 4, 1e-2, inf),
            ("load_inertia", "N", 4, 1e-6, inf), ("motor_inertia", "N", 4, 1e-9, inf),
            ("motor_back_emf", "N", 4, 1e-4, inf), ("motor_resistance", "N", 4, 1e-4, inf),
            ("gear_efficiency", "U", 4, 1e-4, 1), ("motor_efficiency", "U", 4, 1e-4, 1),
            ("combined_damping", "U", 4, 1e-4, inf), ("ball_damping", "U", 4, 1e-4, inf),
            ("voltage_thold_x_pos", "U", 3, -inf, inf), ("voltage_thold_x_neg", "U", 3, -inf, inf),
            ("voltage_thold_y_pos", "U", 3, -inf, inf), ("voltage_thold_y_neg", "U", 3, -inf, inf),
            ("offset_th_x", "Uabs", 6.0 * _DEG, -inf, inf), ("offset_th_y", "Uabs", 6.0 * _DEG, -inf, inf)],
    # :303-372
    "qcp-su": [("gravity_const", "N", 10, 1e-4, inf), ("cart_mass", "N", 5, 1e-4, inf), ("pole_mass", "N", 5, 1e-4, inf),
               ("rail_length", "N", 5, 1e-2, inf), ("pole_length", "N", 5, 1e-2, inf),
               ("motor_efficiency", "U", 4, 1e-4, 1), ("gear_efficiency", "U", 4, 1e-4, 1),
               ("gear_ratio", "N", 4, 1e-4, inf), ("motor_inertia", "N", 4, 1e-9, inf),
               ("pinion_radius", "N", 5, 1e-4, inf), ("motor_resistance", "N", 4, 1e-4, inf),
               ("motor_back_emf", "N", 4, 1e-4, inf), ("combined_damping", "U", 4, 1e-4, inf),
               ("pole_damping", "U", 4, 1e-4, inf), ("cart_friction_coeff", "U", 2, 0, inf)],
    # :375-416
    "qq-su": [("gravity_const", "N", 10, 1e-3, inf), ("motor_resistance", "N", 5, 1e-3, inf),
              ("motor_back_emf", "N", 5, 1e-4, inf), ("mass_rot_pole", "N", 5, 1e-4, inf),
              ("length_rot_pole", "N", 5, 1e-4, inf), ("damping_rot_pole", "N", 4, 1e-9, inf),
              ("mass_pend_pole", "N", 5, 1e-4, inf), ("length_pend_pole", "N", 5, 1e-4, inf),
              ("damping_pend_pole", "N", 4, 1e-9, inf)],
}


# QQubeStabSim / QCartPoleStabSim / BallOnBeamDiscSim share their base family's table (registered on the base classes,
# default_randomizers.py:303-304, 375-376 and the MRO walk of create_default_randomizer :79-86); PendulumSim :208-229
_DEFAULTS["qq-st"] = _DEFAULTS["qq-su"]
_DEFAULTS["qcp-st"] = _DEFAULTS["qcp-su"]
_DEFAULTS["bob-d"] = _DEFAULTS["bob"]
_DEFAULTS["pend"] = [("gravity_const", "N", 10, 1e-3, inf), ("pole_mass", "N", 10, 1e-3, inf),
                     ("pole_length", "N", 10, 1e-3, inf), ("pole_damping", "N", 10, 1e-3, inf),
                     ("torque_thold", "N", 10, 1e-3, inf)]


def default_randomizer_for(name: str, nominal: dict) -> DomainRandomizer:
    """The default randomizer of env family `name` built around the nominal parameter dict."""
    if name not in _DEFAULTS:
        raise ValueErr(msg=f"No default randomizer settings for env of type {name}!")
    dps = []
    for pn, kind, arg, lo, hi in _DEFAULTS[name]:
        nom = nominal[pn]
        if kind == "N":
            dps.append(NormalDomainParam(name=pn, mean=nom, std=nom / arg, clip_lo=lo, clip_up=hi))
        elif kind == "U":
            dps.append(UniformDomainParam(name=pn, mean=nom, halfspan=abs(nom) / arg, clip_lo=lo, clip_up=hi))
        else:  # absolute half span around a nominal of zero
            dps.append(UniformDomainParam(name=pn, mean=nom, halfspan=arg, clip_lo=lo, clip_up=hi))
    return DomainRandomizer(*dps)


def create_default_randomizer(env) -> DomainRandomizer:
    """default_randomizers.py:71-89: looks at the innermost env of a wrapper chain"""
    from .wrappers import inner_env

    e = inner_env(env)
    nominal = e.get_nominal_domain_param()
    if e.name in ("qcp-su", "qcp-st"):
        nominal = type(e).get_nominal_domain_param(long=False)  # the reference builds the table for the short pole (:312)
    return default_randomizer_for(e.name, nominal)


def create_zero_var_randomizer(env, eps: float = 1e-8) -> DomainRandomizer:
    r = create_default_randomizer(env)
    r.rescale_distr_param("std", np.sqrt(eps))
    r.rescale_distr_param("halfspan", np.sqrt(eps))
    return r


def create_conservative_randomizer(env) -> DomainRandomizer:
    r = create_default_randomizer(env)
    r.rescale_distr_param("std", 0.5)
    r.rescale_distr_param("halfspan", 0.5)
    return r
