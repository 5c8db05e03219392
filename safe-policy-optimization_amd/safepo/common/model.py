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

    def kernels_supported(self, family: str = "ppo") -> bool:
        """True when this (obs_dim, act_dim, hidden_sizes) lies inside the envelope of the persistent LDS-resident kernels:
        hidden_sizes [64, 64] (the default_cfg of every reference script), act_dim <= 16 and obs_dim <= 128 for the
        PPO-family kernels (`family="ppo"`: policy step, minibatch update, KL), obs_dim <= 64 for the full-batch CPO kernels
        (`family="cpo"`: surrogate gradient, Fisher-vector product, line search).  Everything else -- the reference takes ANY
        dims (model.py:131; its default sweep includes 72-88-dim Car / Doggo / Racecar observations and HumanoidVelocity's
        376 / 17, single_agent/benchmark.py:5-22) -- runs on the wide-network kernels (safepo.common.wide)."""
        if self.hidden_sizes != [HIDDEN, HIDDEN] or self.act_dim > _abi.MAX_ACT:
            return False
        return self.obs_dim <= (_abi.CPO_MAX_OBS if family == "cpo" else _abi.MAX_OBS)

    def _require_kernels(self, family: str = "ppo"):
        if not self.kernels_supported(family):
            raise NotImplementedError(
                f"this entry point uses the kernels specialised for hidden_sizes=[64, 64], act_dim <= 16, obs_dim <= "
                f"{_abi.CPO_MAX_OBS if family == 'cpo' else _abi.MAX_OBS} (got obs {self.obs_dim}, act {self.act_dim}, hidden "
                f"{self.hidden_sizes}); other shapes run through safepo.common.wide (Wide*Engine)")
        n = _abi.load().spo_param_count(self.obs_dim, self.act_dim)
        assert n == self.theta.numel(), (n, self.theta.numel())

    @property
    def wide(self):
        """Host side of the wide-network kernels for this policy (created on first use; hidden_sizes != [64, 64])."""
        w = self.__dict__.get("_wide")
        if w is None or w.theta_key != (self.theta.data_ptr(), self.theta.device):      # .to(device) / _flatten() re-made theta
            from safepo.common.wide import WideNets
            w = WideNets(self)
            self.__dict__["_wide"] = w
        return w

    @property
    def log_std_offset(self) -> int:
        n_critic = sum(p.numel() for p in self.reward_critic.parameters())
        return 2 * n_critic             # policy.parameters() order: reward critic, cost critic, actor.log_std, actor.mean.*

    # ------------------------------------------------------------------ reference API
    def step(self, obs, deterministic: bool = False, eps: torch.Tensor | None = None):
        """ActorVCritic.step (model.py:149-170) through spo_policy_step.  `eps` may carry the
        rsample noise (parity tests); otherwise it is drawn from torch's device generator."""
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
        if not self.kernels_supported():
            act, logp, v_r, v_c = self.wide.step(obs2, eps)
            return (act[0], logp[0], v_r[0], v_c[0]) if single else (act, logp, v_r, v_c)
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
        obs2 = _abi.require_gpu_tensor(obs.reshape(-1, self.obs_dim).contiguous(), "obs", torch.float32)
        if not self.kernels_supported():
            return self.wide.values(obs2)
        n = obs2.shape[0]
        v_r = torch.empty(n, device=obs2.device, dtype=torch.float32)
        v_c = torch.empty(n, device=obs2.device, dtype=torch.float32)
        _abi.check(_abi.load().spo_values(_abi.ptr(self.theta), _abi.ptr(obs2), _abi.ptr(v_r), _abi.ptr(v_c), n,
                                          self.obs_dim, self.act_dim, _abi.stream_ptr()), "spo_values")
        return v_r, v_c


# ====================================================================== multi-agent networks
def _ma_init(module, gain, orthogonal=True):
    """safepo/utils/util.py init(): weight by orthogonal_/xavier_uniform_ with `gain`, bias 0."""
    (nn.init.orthogonal_ if orthogonal else nn.init.xavier_uniform_)(module.weight.data, gain=gain)
    nn.init.constant_(module.bias.data, 0)
    return module


class _MATrunk(nn.Module):
    """Parameters of MLPBase (safepo/utils/mlp.py) under the reference's state_dict names:
    base.feature_norm.{weight,bias}, base.mlp.fc1.{0,2}.*, base.mlp.fc2.<i>.{0,2}.* -- Linear at index 0, LayerNorm at
    index 2 of each Sequential (index 1 is the ELU)."""

    def __init__(self, config, in_dim):
        super().__init__()
        H, layer_N = config["hidden_size"], config["layer_N"]
        gain = nn.init.calculate_gain(["tanh", "relu"][config["use_ReLU"]])
        orth = config["use_orthogonal"]
        if not config["use_feature_normalization"]:
            raise NotImplementedError("use_feature_normalization=False is not built (the kernels fuse the input LayerNorm)")
        self.feature_norm = nn.LayerNorm(in_dim)

        def block(d):
            return nn.Sequential(_ma_init(nn.Linear(d, H), gain, orth), nn.ELU(), nn.LayerNorm(H))
        mlp = nn.Module()
        mlp.fc1 = block(in_dim)
        mlp.fc2 = nn.ModuleList([block(H) for _ in range(layer_N)])
        self.mlp = mlp


class _MAFlatNet(nn.Module):
    """One MAPPO network over a flat fp32 parameter vector (layout: include/safepo_hip.h, section f3).  forward /
    backward run through spo_ma_forward / spo_ma_backward (in-tree fp32 MFMA GEMMs + fused LayerNorm/ELU kernels); the
    nn.Parameters are views into `theta` so state_dict() keeps the reference's keys and shapes."""

    def _finish(self, in_dim, out_dim, is_actor):
        self._net = _abi.MaNet(in_dim=in_dim, hidden=self.hidden_size, n_blocks=1 + self.config["layer_N"], out_dim=out_dim,
                               is_actor=int(is_actor))
        self.theta = None
        self._flatten()

    def _flatten(self):
        params = list(self.parameters())
        flat = torch.cat([p.detach().reshape(-1) for p in params]).to(torch.float32).contiguous()
        off = 0
        for p in params:
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self.theta = flat
        if flat.is_cuda:
            n = _abi.load().spo_ma_param_count(self._net)
            assert n == flat.numel(), (n, flat.numel())

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def offset(self, which: int, block: int = 0) -> int:
        return int(_abi.load().spo_ma_param_offset(self._net, which, block))

    def net_forward(self, x: torch.Tensor, keep: bool = False):
        """Head output [rows, out] for x [rows, in]; with keep=True also the activation workspace for net_backward."""
        x = _abi.require_gpu_tensor(x.reshape(-1, self._net.in_dim).contiguous(), "x", torch.float32)
        lib = _abi.load()
        rows = x.shape[0]
        ws = torch.empty(int(lib.spo_ma_workspace_floats(self._net, rows)), dtype=torch.float32, device=x.device)
        out = torch.empty((rows, self._net.out_dim), dtype=torch.float32, device=x.device)
        _abi.check(lib.spo_ma_forward(_abi.ptr(self.theta), self._net, _abi.ptr(x), rows, _abi.ptr(ws), _abi.ptr(out),
                                      _abi.stream_ptr()), "spo_ma_forward")
        return (out, (x, ws)) if keep else out

    def net_jvp(self, saved, tangent: torch.Tensor) -> torch.Tensor:
        """d(out)/d(theta) . tangent [rows, out] at the point of `saved` (forward-mode pass, spo_ma_jvp)."""
        x, ws = saved
        lib = _abi.load()
        rows = x.shape[0]
        tangent = _abi.require_gpu_tensor(tangent.contiguous(), "tangent", torch.float32)
        assert tangent.numel() == self.theta.numel()
        scratch = torch.empty(int(lib.spo_ma_jvp_scratch_floats(self._net, rows)), dtype=torch.float32, device=x.device)
        dout = torch.empty((rows, self._net.out_dim), dtype=torch.float32, device=x.device)
        _abi.check(lib.spo_ma_jvp(_abi.ptr(self.theta), self._net, _abi.ptr(tangent), rows, _abi.ptr(ws), _abi.ptr(dout),
                                  _abi.ptr(scratch), _abi.stream_ptr()), "spo_ma_jvp")
        return dout

    def net_backward(self, saved, dout: torch.Tensor, grad: torch.Tensor) -> None:
        x, ws = saved
        lib = _abi.load()
        rows = x.shape[0]
        scratch = torch.empty(int(lib.spo_ma_backward_scratch_floats(self._net, rows)), dtype=torch.float32, device=x.device)
        _abi.check(lib.spo_ma_backward(_abi.ptr(self.theta), self._net, _abi.ptr(x), rows, _abi.ptr(ws), _abi.ptr(dout.contiguous()),
                                       _abi.ptr(grad), _abi.ptr(scratch), _abi.stream_ptr()), "spo_ma_backward")


class MultiAgentActor(_MAFlatNet):
    """safepo/common/model.py:172-301: MLPBase + ACTLayer(DiagGaussian).  forward(obs, rnn_states, masks) ->
    (actions, per-dimension action_log_probs, rnn_states); evaluate_actions(...) -> (action_log_probs, dist_entropy)."""

    def __init__(self, config, obs_space, action_space, device=torch.device("cuda")):
        super().__init__()
        self.config, self.hidden_size = config, config["hidden_size"]
        self.tpdv = dict(dtype=torch.float32, device=device)
        obs_dim, act_dim = obs_space.shape[0], action_space.shape[0]
        self.base = _MATrunk(config, obs_dim)
        act = nn.Module()
        out = nn.Module()
        out.log_std = nn.Parameter(torch.ones(act_dim) * config["std_x_coef"])          # distributions.py:36-37
        out.fc_mean = _ma_init(nn.Linear(self.hidden_size, act_dim), config["actor_gain"], config["use_orthogonal"])
        act.action_out = out
        self.act = act
        self.act_dim = act_dim
        self.std_x_coef, self.std_y_coef = float(config["std_x_coef"]), float(config["std_y_coef"])
        self._finish(obs_dim, act_dim, True)
        self.to(device)

    @property
    def log_std(self) -> torch.Tensor:
        o = self.offset(6)
        return self.theta[o:o + self.act_dim]

    def forward(self, obs, rnn_states, masks, available_actions=None, deterministic=False, eps=None):
        mean = self.net_forward(torch.as_tensor(obs, **self.tpdv))
        rows = mean.shape[0]
        if not deterministic and eps is None:
            eps = torch.randn((rows, self.act_dim), **self.tpdv)
        act, logp = torch.empty_like(mean), torch.empty_like(mean)
        _abi.check(_abi.load().spo_ma_sample(_abi.ptr(mean), _abi.ptr(self.log_std), _abi.ptr(eps) if eps is not None else None,
                                             self.std_x_coef, self.std_y_coef, int(bool(deterministic)), _abi.ptr(act),
                                             _abi.ptr(logp), rows, self.act_dim, _abi.stream_ptr()), "spo_ma_sample")
        return act, logp, rnn_states

    def evaluate_actions(self, obs, rnn_states, action, masks, available_actions=None, active_masks=None):
        mean = self.net_forward(torch.as_tensor(obs, **self.tpdv))
        action = _abi.require_gpu_tensor(torch.as_tensor(action, **self.tpdv).reshape(mean.shape).contiguous(), "action", torch.float32)
        logp = torch.empty_like(mean)
        _abi.check(_abi.load().spo_ma_log_probs(_abi.ptr(mean), _abi.ptr(self.log_std), _abi.ptr(action), self.std_x_coef,
                                                self.std_y_coef, _abi.ptr(logp), mean.shape[0], self.act_dim,
                                                _abi.stream_ptr()), "spo_ma_log_probs")
        std = torch.sigmoid(self.log_std / self.std_x_coef) * self.std_y_coef
        ent = 0.5 + 0.5 * np.log(2 * np.pi) + torch.log(std)                        # Normal.entropy, same in every row
        if active_masks is not None and self.config["use_policy_active_masks"]:
            dist_entropy = ent.sum()                                                 # (ent * mask).sum() / mask.sum()
        else:
            dist_entropy = ent.mean()
        if self.config.get("algorithm_name") == "macpo":       # model.py:282-290 (evaluate_actions_trpo): + mean and stddev
            return logp, dist_entropy, mean, std.expand_as(mean)
        return logp, dist_entropy


class MultiAgentCritic(_MAFlatNet):
    """safepo/common/model.py:304-363: MLPBase + v_out (initialised with gain 0).  forward(cent_obs, rnn_states, masks)
    -> (values [rows, 1], rnn_states)."""

    def __init__(self, config, cent_obs_space, device=torch.device("cuda")):
        super().__init__()
        self.config, self.hidden_size = config, config["hidden_size"]
        self.tpdv = dict(dtype=torch.float32, device=device)
        in_dim = cent_obs_space.shape[0]
        self.base = _MATrunk(config, in_dim)
        self.v_out = _ma_init(nn.Linear(self.hidden_size, 1), 0, config["use_orthogonal"])
        self._finish(in_dim, 1, False)
        self.to(device)

    def forward(self, cent_obs, rnn_states, masks):
        return self.net_forward(torch.as_tensor(cent_obs, **self.tpdv)), rnn_states
