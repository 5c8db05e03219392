"""ActorVCritic over ONE flat fp32 parameter vector on the GPU.

Mirror of the reference surface (safepo/common/model.py:51-170): `Actor`, `VCritic`,
`ActorVCritic(obs_dim, act_dim, hidden_sizes)` with `.actor(obs) -> Normal`, `.reward_critic(obs)`,
`.cost_critic(obs)`, `.step(obs, deterministic) -> (act, logp, v_r, v_c)` and identical
state_dict keys (checkpoint contract: logger.torch_save saves policy.actor.state_dict(),
reference safepo/common/logger.py:255-271).

MI355X design: every nn.Parameter is a VIEW into `self.theta`, a single contiguous fp32 vector in
`policy.parameters()` order (layout in include/safepo_hip.h).  The HIP kernels read and update
`theta` in place, so nn.Module views, optimiser state (flat m/v) and the data-parallel gradient
all-reduce (one 99 KB message) share one address range.  `step()` runs the fused HIP kernel
(spo_policy_step); it raises on CPU tensors -- there is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal

from safepo import _abi

HIDDEN = 64  # SPO_HIDDEN


def build_mlp_network(sizes):
    """tanh MLP, identity output; weights kaiming_uniform_(a=sqrt(5)) drawn AFTER nn.Linear's own
    init so the RNG stream matches the reference construction (model.py:30-48)."""
    layers = []
    for j in range(len(sizes) - 1):
        lin = nn.Linear(sizes[j], sizes[j + 1])
        nn.init.kaiming_uniform_(lin.weight, a=np.sqrt(5))
        layers += [lin, nn.Tanh() if j < len(sizes) - 2 else nn.Identity()]
    return nn.Sequential(*layers)


class Actor(nn.Module):
    """Gaussian policy with state-independent log_std (model.py:51-81)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden_sizes=(64, 64)):
        super().__init__()
        self.mean = build_mlp_network([obs_dim] + list(hidden_sizes) + [act_dim])
        self.log_std = nn.Parameter(torch.zeros(act_dim), requires_grad=True)

    def forward(self, obs: torch.Tensor):
        return Normal(self.mean(obs), torch.exp(self.log_std))


class VCritic(nn.Module):
    """State-value critic (model.py:84-108)."""

    def __init__(self, obs_dim: int, hidden_sizes=(64, 64)):
        super().__init__()
        self.critic = build_mlp_network([obs_dim] + list(hidden_sizes) + [1])

    def forward(self, obs):
        return torch.squeeze(self.critic(obs), -1)


class ActorVCritic(nn.Module):
    """reward critic, cost critic, actor -- registered in the reference order (model.py:131-135)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden_sizes=(64, 64)):
        super().__init__()
        hidden_sizes = list(hidden_sizes)
        self.obs_dim, self.act_dim, self.hidden_sizes = int(obs_dim), int(act_dim), hidden_sizes
        self.reward_critic = VCritic(obs_dim, hidden_sizes)
        self.cost_critic = VCritic(obs_dim, hidden_sizes)
        self.actor = Actor(obs_dim, act_dim, hidden_sizes)
        self.theta = None
        self._flatten()

    # ------------------------------------------------------------------ flat storage
    def _flatten(self):
        params = list(self.parameters())
        flat = torch.cat([p.detach().reshape(-1) for p in params]).to(torch.float32).contiguous()
        off = 0
        for p in params:
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self.theta = flat

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def kernels_supported(self) -> bool:
        return self.hidden_sizes == [HIDDEN, HIDDEN]

    def _require_kernels(self):
        if not self.kernels_supported():
            raise NotImplementedError(
                f"HIP kernels are specialised for hidden_sizes=[64, 64] (got {self.hidden_sizes}); "
                "isaac_gym_specific_cfg shapes are not built yet")
        n = _abi.load().spo_param_count(self.obs_dim, self.act_dim)
        assert n == self.theta.numel(), (n, self.theta.numel())

    @property
    def log_std_offset(self) -> int:
        return int(_abi.load().spo_param_offset(self.obs_dim, self.act_dim, 2))

    # ------------------------------------------------------------------ reference API
    def step(self, obs, deterministic: bool = False, eps: torch.Tensor | None = None):
        """ActorVCritic.step (model.py:149-170) through spo_policy_step.  `eps` may carry the
        rsample noise (parity tests); otherwise it is drawn from torch's device generator."""
        self._require_kernels()
        single = obs.dim() == 1
        obs2 = obs.reshape(1, -1) if single else obs
        obs2 = _abi.require_gpu_tensor(obs2.contiguous(), "obs", torch.float32)
        n = obs2.shape[0]
        dev = obs2.device
        if deterministic:
            eps = None
        elif eps is None:
            eps = torch.randn((n, self.act_dim), device=dev, dtype=torch.float32)
        else:
            eps = _abi.require_gpu_tensor(eps.reshape(n, self.act_dim).contiguous(), "eps", torch.float32)
        act = torch.empty((n, self.act_dim), device=dev, dtype=torch.float32)
        logp = torch.empty(n, device=dev, dtype=torch.float32)
        v_r = torch.empty(n, device=dev, dtype=torch.float32)
        v_c = torch.empty(n, device=dev, dtype=torch.float32)
        lib = _abi.load()
        _abi.check(lib.spo_policy_step(_abi.ptr(self.theta), _abi.ptr(obs2), _abi.ptr(eps), _abi.ptr(act),
                                       _abi.ptr(logp), _abi.ptr(v_r), _abi.ptr(v_c), None, None, None, None, None,
                                       n, 1, 0, self.obs_dim, self.act_dim, _abi.stream_ptr()), "spo_policy_step")
        if single:
            return act[0], logp[0], v_r[0], v_c[0]
        return act, logp, v_r, v_c

    def values(self, obs):
        """(v_r, v_c) for a batch of observations (bootstrap calls, ppo_lag.py:201-215)."""
        self._require_kernels()
        obs2 = _abi.require_gpu_tensor(obs.reshape(-1, self.obs_dim).contiguous(), "obs", torch.float32)
        n = obs2.shape[0]
        v_r = torch.empty(n, device=obs2.device, dtype=torch.float32)
        v_c = torch.empty(n, device=obs2.device, dtype=torch.float32)
        _abi.check(_abi.load().spo_values(_abi.ptr(self.theta), _abi.ptr(obs2), _abi.ptr(v_r), _abi.ptr(v_c), n,
                                          self.obs_dim, self.act_dim, _abi.stream_ptr()), "spo_values")
        return v_r, v_c
