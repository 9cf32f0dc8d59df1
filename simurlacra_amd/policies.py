"""The two trivial policies the reference's tests drive the envs with (P/policies/feed_forward/dummy.py:40-84)."""
import torch


class Policy(torch.nn.Module):
    is_recurrent = False

    def __init__(self, spec):
        super().__init__()
        self.env_spec = spec

    def reset(self, **kwargs):
        pass


class IdlePolicy(Policy):
    """always zero (dummy.py:40-57)"""

    def forward(self, obs: torch.Tensor = None) -> torch.Tensor:
        shape = tuple(self.env_spec.act_space.shape)
        if obs is not None and obs.dim() == 2:
            shape = (obs.shape[0],) + shape
        return torch.zeros(shape, device=obs.device if obs is not None else None)


class DummyPolicy(Policy):
    """uniform random action in the action space, cast to fp32 (dummy.py:60-84); batched when obs is [N, O]"""

    def forward(self, obs: torch.Tensor = None) -> torch.Tensor:
        lo = torch.as_tensor(self.env_spec.act_space.bound_lo, dtype=torch.float32)
        hi = torch.as_tensor(self.env_spec.act_space.bound_up, dtype=torch.float32)
        if obs is not None and obs.dim() == 2:
            lo, hi = lo.to(obs.device), hi.to(obs.device)
            return lo + (hi - lo) * torch.rand(obs.shape[0], lo.numel(), device=obs.device)
        return torch.from_numpy(self.env_spec.act_space.sample_uniform()).to(torch.float32)


# ------------------------------------------------------------------------------------------------- feed-forward network
def _init_linear(m):
    """init_param for nn.Linear (P/policies/initialization.py:64-70): PyTorch's default initialisation"""
    from math import sqrt

    torch.nn.init.kaiming_uniform_(m.weight, a=sqrt(5))
    if m.bias is not None:
        fan_in = m.weight.shape[1]
        bound = 1 / sqrt(fan_in) if fan_in > 0 else 0
        torch.nn.init.uniform_(m.bias, -bound, bound)


class FNN(torch.nn.Module):
    """Feed-forward neural network (P/policies/feed_back/fnn.py:43-160): hidden Linear layers with a nonlinearity each, a
    Linear output layer with an optional one.  Parameters in the reference's order: hidden_layers.i.weight / .bias ...,
    output_layer.weight / .bias."""

    def __init__(self, input_size, output_size, hidden_sizes, hidden_nonlin, dropout=0.0, output_nonlin=None,
                 init_param_kwargs=None, use_cuda=False):
        super().__init__()
        self._device = "cuda" if use_cuda and torch.cuda.is_available() else "cpu"
        hidden_sizes = list(hidden_sizes)
        self.hidden_nonlin = list(hidden_nonlin) if isinstance(hidden_nonlin, (list, tuple)) else len(hidden_sizes) * [hidden_nonlin]
        self.dropout = dropout
        self.output_nonlin = output_nonlin
        self.hidden_layers = torch.nn.ModuleList()
        last = input_size
        for hs in hidden_sizes:
            self.hidden_layers.append(torch.nn.Linear(last, hs))
            last = hs
            if self.dropout > 0:
                self.hidden_layers.append(torch.nn.Dropout(p=self.dropout))
        self.output_layer = torch.nn.Linear(last, output_size)
        self.init_param(None, **(init_param_kwargs or {}))
        self.to(self._device)

    @property
    def device(self):
        return self._device

    @property
    def param_values(self):
        return torch.nn.utils.parameters_to_vector(self.parameters())

    @param_values.setter
    def param_values(self, param):
        torch.nn.utils.vector_to_parameters(param, self.parameters())

    def init_param(self, init_values=None, **kwargs):
        if init_values is None:
            for layer in list(self.hidden_layers) + [self.output_layer]:
                if isinstance(layer, torch.nn.Linear):
                    _init_linear(layer)
        else:
            self.param_values = init_values

    def forward(self, obs):
        p0 = next(self.parameters(), None)
        x = obs if p0 is None else obs.to(p0.device)  # (fnn.py: obs.to(self.device); a no-op on the same device, also under graph capture)
        for i, layer in enumerate(self.hidden_layers):
            x = layer(x)
            if self.dropout == 0:
                if self.hidden_nonlin[i] is not None:
                    x = self.hidden_nonlin[i](x)
            elif i % 2 == 0 and self.hidden_nonlin[i // 2] is not None:
                x = self.hidden_nonlin[i // 2](x)
        out = self.output_layer(x)
        return self.output_nonlin(out) if self.output_nonlin is not None else out


class FNNPolicy(Policy):
    """Feed-forward neural network policy (P/policies/feed_back/fnn.py:163-222).  The fork's forward() feeds the network
    [o_0, sin o_1, cos o_1, o_2 ..] -- its cartpole observes the state (quanser_cartpole.py:107-109 in the fork returns it
    unchanged), so row 1 is the pole angle -- and sizes the input layer obs_dim + 1 accordingly: `featurize=True` (default) is
    that behaviour, `featurize=False` the plain net(obs) of upstream Pyrado."""

    name = "fnn"

    def __init__(self, spec, hidden_sizes, hidden_nonlin, dropout=0.0, output_nonlin=None, init_param_kwargs=None,
                 use_cuda=False, featurize=True):
        super().__init__(spec)
        self.featurize = bool(featurize)
        # the reference's order (fnn.py:187-201): the net initialises once WITHOUT the kwargs, then the policy calls
        # init_param(None, **init_param_kwargs) -- the same draws from torch's RNG as the reference under the same seed
        self.net = FNN(spec.obs_space.flat_dim + (1 if self.featurize else 0), spec.act_space.flat_dim, hidden_sizes,
                       hidden_nonlin, dropout, output_nonlin, None, use_cuda)
        self.init_param(None, **(init_param_kwargs or {}))

    @property
    def param_values(self):
        return torch.nn.utils.parameters_to_vector(self.parameters())

    @param_values.setter
    def param_values(self, param):
        torch.nn.utils.vector_to_parameters(param, self.parameters())

    def init_param(self, init_values=None, **kwargs):
        if init_values is None:
            self.net.init_param(None, **kwargs)
        else:
            self.param_values = init_values

    def forward(self, obs):
        if self.featurize:
            obs = torch.cat([obs[..., 0:1], torch.sin(obs[..., 1:2]), torch.cos(obs[..., 1:2]), obs[..., 2:]], dim=-1)
        return self.net(obs)


class NormalActNoiseExplStrat(Policy):
    """Gaussian noise on the actions of a wrapped policy (P/exploration/stochastic_action.py:121-180, shallow form: a fixed
    or externally updated diagonal std)"""

    def __init__(self, policy, std_init, std_min=1e-3):
        super().__init__(policy.env_spec)
        self.policy = policy
        n = policy.env_spec.act_space.flat_dim
        std = torch.as_tensor(std_init, dtype=torch.float32).reshape(-1)
        # a buffer: policy.to(device) moves it (a pageable CPU tensor copied inside forward() is a synchronous host-to-device
        # copy, illegal during the stream capture of ParallelRolloutSampler(graph_policy=True))
        self.register_buffer("std", torch.clamp(std.expand(n).clone(), min=float(std_min)))

    def reset(self, **kwargs):
        self.policy.reset(**kwargs)

    def forward(self, obs):
        act = self.policy(obs)
        std = self.std if self.std.device == act.device else self.std.to(act.device)
        return act + std.to(act.dtype) * torch.randn_like(act)


_NONLIN_NAMES = {torch.tanh: "tanh", torch.nn.functional.tanh: "tanh", torch.relu: "relu", torch.nn.functional.relu: "relu",
                 torch.sigmoid: "sigmoid", torch.nn.functional.sigmoid: "sigmoid", None: None}


def _nonlin_name(f):
    if isinstance(f, torch.nn.Tanh):
        return "tanh"
    if isinstance(f, torch.nn.ReLU):
        return "relu"
    if isinstance(f, torch.nn.Sigmoid):
        return "sigmoid"
    if isinstance(f, torch.nn.Identity):
        return None
    return _NONLIN_NAMES[f]  # KeyError: not a nonlinearity the kernel has


def fnn_kernel_spec(policy):
    """The arguments of VecSimEnv.set_policy_fnn for a policy the fused kernel can evaluate itself -- an FNN / FNNPolicy of at
    most 4 hidden layers of at most 64 units, tanh / relu / sigmoid / no nonlinearities, no dropout, optionally inside a
    NormalActNoiseExplStrat -- or None (the sampler then keeps the policy in torch, one vs_step_record per step)."""
    noise_std = None
    if isinstance(policy, NormalActNoiseExplStrat):
        noise_std = policy.std.detach().cpu().numpy()
        policy = policy.policy
    feat = False
    if isinstance(policy, FNNPolicy):
        feat, net = policy.featurize, policy.net
    elif isinstance(policy, FNN):
        net = policy
    else:
        return None
    if net.dropout > 0:
        return None
    sizes = [layer.out_features for layer in net.hidden_layers]
    if not 1 <= len(sizes) <= 4 or max(sizes) > 64:
        return None
    try:
        hidden_nonlin = [_nonlin_name(f) for f in net.hidden_nonlin[:len(sizes)]]
        output_nonlin = _nonlin_name(net.output_nonlin)
    except (KeyError, TypeError):
        return None
    return dict(params=torch.nn.utils.parameters_to_vector(net.parameters()).detach().to(torch.float32),
                hidden_sizes=sizes, hidden_nonlin=hidden_nonlin, output_nonlin=output_nonlin, feat=feat, noise_std=noise_std)
