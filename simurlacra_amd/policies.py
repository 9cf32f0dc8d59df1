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
