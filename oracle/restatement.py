"""TEST INFRASTRUCTURE -- CPU restatement (oracle) of SafePO's single-agent
collect -> reward/cost GAE -> PPO-Lagrangian / CPO update path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / the timed CPU baseline.  The product
path (safe-policy-optimization_amd/safepo) never imports it.

PARITY PINNING: the reference's own tests hold no golden vectors for this path
(SURVEY.md section 4 / 8c).  This restatement is pinned instead against outputs
of the unmodified reference code executed in the build container
(oracle/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).

Arithmetic that lives in third-party torch in the reference (nn.Linear, tanh,
Normal.log_prob, kl_divergence, mse_loss, clip_grad_norm_, Adam, std) is used
through the same torch CPU calls here; everything the reference writes itself
(GAE recurrence, boundary logic, losses, CG, CPO case analysis) is restated.
Each function cites the reference file:line (relative to /root/reference) it follows.
"""
from __future__ import annotations

import math
from collections import deque

import numpy as np
import torch


# --------------------------------------------------------------------------
# a-5  GAE
# --------------------------------------------------------------------------
def gae_path(values: np.ndarray, rewards: np.ndarray, gamma: float, lam: float):
    """One path.  `values` has length L+1 (bootstrap appended), `rewards` length L+1
    (last entry unused).  safepo/common/buffer.py:191-201 + 167-188.

    delta in fp32 with three separately rounded ops and gamma rounded to fp32
    (buffer.py:198); reverse discounted cumsum in fp64 with the python-double
    product gamma*lam (buffer.py:182-188,199); target = adv(f64) + v(f32->f64)
    (buffer.py:200); both rounded to fp32 when stored (buffer.py:135-138).
    """
    values = np.asarray(values, dtype=np.float32)
    rewards = np.asarray(rewards, dtype=np.float32)
    g32 = np.float32(gamma)
    deltas = (rewards[:-1] + g32 * values[1:]) - values[:-1]          # fp32, separately rounded
    x = deltas.astype(np.float64)
    disc = float(gamma) * float(lam)                                     # python double product
    c = x[-1] if len(x) else 0.0
    for i in range(len(x) - 2, -1, -1):
        c = x[i] + disc * c
        x[i] = c
    adv = x
    tgt = adv + values[:-1].astype(np.float64)
    return adv.astype(np.float32), tgt.astype(np.float32)


def gae_dense(reward, cost, value_r, value_c, seg_end, boot_r, boot_c,
              gamma: float, lam: float, lam_c: float):
    """Dense [N,T] restatement of finish_path() applied at every segment end
    (safepo/common/buffer.py:97-140).  seg_end[n,t] != 0 marks the last step of a
    path; boot_*[n,t] is the bootstrap value appended for that path.
    Sequential in t (exactly the reference order), vectorised over envs.
    Returns adv_r, adv_c, tgt_r, tgt_c (fp32 [N,T]).
    """
    reward = np.asarray(reward, np.float32)
    N, T = reward.shape
    seg = np.asarray(seg_end).astype(bool)
    assert seg[:, T - 1].all(), "every row must end a path at t=T-1 (epoch end)"
    out = []
    for rew, val, boot, lm in ((reward, value_r, boot_r, lam), (cost, value_c, boot_c, lam_c)):
        rew = np.asarray(rew, np.float32)
        val = np.asarray(val, np.float32)
        boot = np.asarray(boot, np.float32)
        g32 = np.float32(gamma)
        nxt = np.empty_like(val)
        nxt[:, :-1] = val[:, 1:]
        nxt[:, -1] = 0
        nxt = np.where(seg, boot, nxt)
        delta = (rew + g32 * nxt) - val                                   # fp32 x3 roundings
        x = delta.astype(np.float64)
        disc = float(gamma) * float(lm)
        adv = np.empty((N, T), np.float64)
        c = np.zeros(N, np.float64)
        for t in range(T - 1, -1, -1):
            c = np.where(seg[:, t], x[:, t], x[:, t] + disc * c)          # select, not multiply
            adv[:, t] = c
        tgt = adv + val.astype(np.float64)
        out.append((adv.astype(np.float32), tgt.astype(np.float32)))
    (adv_r, tgt_r), (adv_c, tgt_c) = out
    return adv_r, adv_c, tgt_r, tgt_c


def adv_standardize(adv_r: torch.Tensor, adv_c: torch.Tensor,
                    standardized_adv_r: bool = True, standardized_adv_c: bool = True):
    """safepo/common/buffer.py:154-160: unbiased std, +1e-8 on std; adv_c only centred."""
    adv_mean = adv_r.mean()
    adv_std = adv_r.std()
    cadv_mean = adv_c.mean()
    if standardized_adv_r:
        adv_r = (adv_r - adv_mean) / (adv_std + 1e-8)
    if standardized_adv_c:
        adv_c = adv_c - cadv_mean
    return adv_r, adv_c


def adv_mix(adv_r: torch.Tensor, adv_c: torch.Tensor, lam: float):
    """safepo/single_agent/ppo_lag.py:280-281 (two separate fp32 ops)."""
    advantage = adv_r - lam * adv_c
    advantage = advantage / (lam + 1)
    return advantage


# --------------------------------------------------------------------------
# a-4  path-boundary logic
# --------------------------------------------------------------------------
def boundary_step(terminated, truncated, epoch_end: bool, v_next_r, v_next_c, v_final_r, v_final_c):
    """safepo/single_agent/ppo_lag.py:198-234 for one vector step.
    Returns (seg_end[N] bool, boot_r[N], boot_c[N]).
    done -> 0; else value(next obs) at epoch end, overridden by value(final obs) if time_out.
    """
    terminated = np.asarray(terminated).astype(bool)
    truncated = np.asarray(truncated).astype(bool)
    n = terminated.shape[0]
    seg = np.zeros(n, bool)
    br = np.zeros(n, np.float32)
    bc = np.zeros(n, np.float32)
    for idx in range(n):
        done, time_out = terminated[idx], truncated[idx]
        if epoch_end or done or time_out:
            seg[idx] = True
            if not done:
                if epoch_end:
                    br[idx], bc[idx] = v_next_r[idx], v_next_c[idx]
                if time_out:
                    br[idx], bc[idx] = v_final_r[idx], v_final_c[idx]
    return seg, br, bc


# --------------------------------------------------------------------------
# a-1  model
# --------------------------------------------------------------------------
def make_mlp(sizes):
    """safepo/common/model.py:30-48 (tanh hidden, identity out, kaiming_uniform a=sqrt(5))."""
    layers = []
    for j in range(len(sizes) - 1):
        lin = torch.nn.Linear(sizes[j], sizes[j + 1])
        torch.nn.init.kaiming_uniform_(lin.weight, a=np.sqrt(5))
        layers.append(lin)
        layers.append(torch.nn.Tanh() if j < len(sizes) - 2 else torch.nn.Identity())
    return torch.nn.Sequential(*layers)


class OraclePolicy(torch.nn.Module):
    """Same parameter names / registration order as ActorVCritic
    (safepo/common/model.py:131-135): reward_critic.critic.*, cost_critic.critic.*,
    actor.log_std, actor.mean.*  (SURVEY.md appendix A item 15)."""

    class _Actor(torch.nn.Module):
        def __init__(self, obs_dim, act_dim, hidden):
            super().__init__()
            self.mean = make_mlp([obs_dim] + list(hidden) + [act_dim])
            self.log_std = torch.nn.Parameter(torch.zeros(act_dim))

        def forward(self, obs):
            return torch.distributions.Normal(self.mean(obs), torch.exp(self.log_std))

    class _Critic(torch.nn.Module):
        def __init__(self, obs_dim, hidden):
            super().__init__()
            self.critic = make_mlp([obs_dim] + list(hidden) + [1])

        def forward(self, obs):
            return torch.squeeze(self.critic(obs), -1)

    def __init__(self, obs_dim, act_dim, hidden_sizes=(64, 64)):
        super().__init__()
        self.reward_critic = self._Critic(obs_dim, hidden_sizes)
        self.cost_critic = self._Critic(obs_dim, hidden_sizes)
        self.actor = self._Actor(obs_dim, act_dim, hidden_sizes)

    def step_with_eps(self, obs, eps):
        """model.py:149-170 with the rsample noise supplied (a = mean + std*eps)."""
        dist = self.actor(obs)
        act = dist.mean + dist.stddev * eps
        logp = dist.log_prob(act).sum(axis=-1)
        return act, logp, self.reward_critic(obs), self.cost_critic(obs)


# --------------------------------------------------------------------------
# a-10 / a-11  PPO-Lagrangian minibatch step and KL
# --------------------------------------------------------------------------
def ppo_lag_losses(policy: OraclePolicy, obs_b, act_b, logp_b, tgt_r_b, tgt_c_b, adv_b,
                   use_critic_norm: bool = True, use_value_coefficient: bool = False,
                   clip: float = 0.2):
    """safepo/single_agent/ppo_lag.py:306-323."""
    loss_r = torch.nn.functional.mse_loss(policy.reward_critic(obs_b), tgt_r_b)
    loss_c = torch.nn.functional.mse_loss(policy.cost_critic(obs_b), tgt_c_b)
    if use_critic_norm:
        for p in policy.reward_critic.parameters():
            loss_r = loss_r + p.pow(2).sum() * 0.001
        for p in policy.cost_critic.parameters():
            loss_c = loss_c + p.pow(2).sum() * 0.001
    dist = policy.actor(obs_b)
    logp = dist.log_prob(act_b).sum(dim=-1)
    ratio = torch.exp(logp - logp_b)
    ratio_clipped = torch.clamp(ratio, 1.0 - clip, 1.0 + clip)
    loss_pi = -torch.min(ratio * adv_b, ratio_clipped * adv_b).mean()
    total = loss_pi + 2 * loss_r + loss_c if use_value_coefficient else loss_pi + loss_r + loss_c
    return total, loss_pi, loss_r, loss_c


def flat_grads(policy: torch.nn.Module) -> torch.Tensor:
    return torch.cat([p.grad.reshape(-1) for p in policy.parameters()])


def flat_params(policy: torch.nn.Module) -> torch.Tensor:
    return torch.cat([p.detach().reshape(-1) for p in policy.parameters()])


class PPOLagUpdater:
    """Optimisers exactly as safepo/single_agent/ppo_lag.py:104-117 (3x Adam lr 3e-4,
    actor LinearLR 1->0 over `epochs`)."""

    def __init__(self, policy: OraclePolicy, epochs: int = 1, lr: float = 3e-4,
                 max_grad_norm: float = 40.0, **loss_kw):
        self.policy = policy
        self.opt_a = torch.optim.Adam(policy.actor.parameters(), lr=lr)
        self.sched = torch.optim.lr_scheduler.LinearLR(
            self.opt_a, start_factor=1.0, end_factor=0.0, total_iters=epochs)
        self.opt_r = torch.optim.Adam(policy.reward_critic.parameters(), lr=lr)
        self.opt_c = torch.optim.Adam(policy.cost_critic.parameters(), lr=lr)
        self.max_grad_norm = max_grad_norm
        self.loss_kw = loss_kw

    def minibatch_step(self, obs_b, act_b, logp_b, tgt_r_b, tgt_c_b, adv_b, record=None):
        """ppo_lag.py:306-329.  `record` (dict) receives pre-clip flat grad and norm."""
        self.opt_r.zero_grad()
        self.opt_c.zero_grad()
        self.opt_a.zero_grad()
        total, loss_pi, loss_r, loss_c = ppo_lag_losses(
            self.policy, obs_b, act_b, logp_b, tgt_r_b, tgt_c_b, adv_b, **self.loss_kw)
        total.backward()
        if record is not None:
            record["grad_preclip"] = flat_grads(self.policy).clone()
        norm = torch.nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
        if record is not None:
            record["grad_norm"] = float(norm)
        self.opt_r.step()
        self.opt_c.step()
        self.opt_a.step()
        return loss_r.item(), loss_c.item(), loss_pi.item()


def actor_kl(policy: OraclePolicy, obs, old_mean, old_std) -> float:
    """ppo_lag.py:338-345: KL(old||new).sum(-1, keepdim).mean()."""
    old = torch.distributions.Normal(old_mean, old_std)
    with torch.no_grad():
        new = policy.actor(obs)
        return float(torch.distributions.kl.kl_divergence(old, new).sum(-1, keepdim=True).mean())


def ppo_lag_update(policy, updater: PPOLagUpdater, data: dict, lam: float, perms,
                   learning_iters: int = 40, batch_size: int = 64, target_kl: float = 0.02,
                   trace=None):
    """ppo_lag.py:275-350 with the shuffles supplied (`perms[i]` = permutation of [0,M) used
    by learning iter i; consecutive chunks of batch_size, last partial kept)."""
    with torch.no_grad():
        old = policy.actor(data["obs"])
        old_mean, old_std = old.mean.clone(), old.stddev.clone()
    advantage = adv_mix(data["adv_r"], data["adv_c"], lam)
    M = data["obs"].shape[0]
    losses = []
    stop_iter, final_kl = 0, 1.0
    for it in range(learning_iters):
        perm = torch.as_tensor(perms[it], dtype=torch.long)
        for s in range(0, M, batch_size):
            idx = perm[s:s + batch_size]
            rec = {} if trace is not None else None
            l = updater.minibatch_step(data["obs"][idx], data["act"][idx], data["log_prob"][idx],
                                       data["target_value_r"][idx], data["target_value_c"][idx],
                                       advantage[idx], record=rec)
            losses.append(l)
            if trace is not None:
                rec["losses"] = l
                trace.append(rec)
        final_kl = actor_kl(policy, data["obs"], old_mean, old_std)
        stop_iter += 1
        if final_kl > target_kl:
            break
    updater.sched.step()
    return {"losses": np.asarray(losses, np.float64), "stop_iter": stop_iter, "kl": final_kl}


# --------------------------------------------------------------------------
# a-7  Lagrange multiplier
# --------------------------------------------------------------------------

class OracleLagrange:
    """safepo/common/lagrange.py:24-105: scalar Parameter, Adam(lr), loss -lambda*(Jc-limit),
    clamp >= 0; read-out = relu(lambda).item()."""

    def __init__(self, cost_limit, lagrangian_multiplier_init, lagrangian_multiplier_lr, lagrangian_upper_bound=None):
        self.cost_limit = cost_limit
        self.upper = lagrangian_upper_bound
        self._lam = torch.nn.Parameter(torch.as_tensor(max(lagrangian_multiplier_init, 0.0)))
        self._opt = torch.optim.Adam([self._lam], lr=lagrangian_multiplier_lr)

    @property
    def lagrangian_multiplier(self) -> float:
        return torch.relu(self._lam).detach().item()

    def update_lagrange_multiplier(self, Jc: float) -> None:
        self._opt.zero_grad()
        loss = -self._lam * (Jc - self.cost_limit)
        loss.backward()
        self._opt.step()
        self._lam.data.clamp_(0.0, self.upper)              # lagrange.py:103-105


# ------------------------------------------------------------------ FOCOPS / CUP
FOCOPS_LAM = 1.50          # focops.py:44
CUP_LAMBDA = 0.95          # cup.py:44


def focops_actor_loss(policy: OraclePolicy, obs_b, act_b, logp_b, adv_b, old_mean_b, old_std_b, target_kl: float):
    """safepo/single_agent/focops.py:326-337, shapes as there: ratio and adv_b are [B], temp_kl is [B, 1], so the
    product broadcasts to a [B, B] matrix before .mean()."""
    old = torch.distributions.Normal(old_mean_b, old_std_b)
    dist = policy.actor(obs_b)
    logp = dist.log_prob(act_b).sum(dim=-1)
    ratio = torch.exp(logp - logp_b)
    temp_kl = torch.distributions.kl_divergence(dist, old).sum(-1, keepdim=True)
    loss_pi = (temp_kl - (1 / FOCOPS_LAM) * ratio * adv_b) * (temp_kl.detach() <= target_kl).type(torch.float32)
    return loss_pi.mean()


def cup_second_stage_loss(policy: OraclePolicy, obs_b, act_b, logp_b, adv_c_b, old_mean_b, old_std_b,
                          lagrangian_multiplier: float, gamma: float):
    """safepo/single_agent/cup.py:372-383 (same [B] vs [B, 1] broadcast)."""
    old = torch.distributions.Normal(old_mean_b, old_std_b)
    dist = policy.actor(obs_b)
    logp = dist.log_prob(act_b).sum(dim=-1)
    ratio = torch.exp(logp - logp_b)
    temp_kl = torch.distributions.kl_divergence(dist, old).sum(-1, keepdim=True)
    coef = (1 - gamma * CUP_LAMBDA) / (1 - gamma)
    return (lagrangian_multiplier * coef * ratio * adv_c_b + temp_kl).mean()


class KLPenaltyUpdater(PPOLagUpdater):
    """The optimisers of PPOLagUpdater driving the FOCOPS minibatch step (focops.py:312-347) and CUP's actor-only
    second stage (cup.py:370-386)."""

    def focops_step(self, obs_b, act_b, logp_b, tgt_r_b, tgt_c_b, adv_b, old_mean_b, old_std_b, target_kl, record=None):
        self.opt_r.zero_grad()
        self.opt_c.zero_grad()
        self.opt_a.zero_grad()
        pol = self.policy
        loss_r = torch.nn.functional.mse_loss(pol.reward_critic(obs_b), tgt_r_b)
        loss_c = torch.nn.functional.mse_loss(pol.cost_critic(obs_b), tgt_c_b)
        if self.loss_kw.get("use_critic_norm", True):
            for prm in pol.reward_critic.parameters():
                loss_r = loss_r + prm.pow(2).sum() * 0.001
            for prm in pol.cost_critic.parameters():
                loss_c = loss_c + prm.pow(2).sum() * 0.001
        loss_pi = focops_actor_loss(pol, obs_b, act_b, logp_b, adv_b, old_mean_b, old_std_b, target_kl)
        total = loss_pi + 2 * loss_r + loss_c if self.loss_kw.get("use_value_coefficient", False) \
            else loss_pi + loss_r + loss_c
        total.backward()
        if record is not None:
            record["grad_preclip"] = flat_grads(pol).clone()
        torch.nn.utils.clip_grad_norm_(pol.parameters(), self.max_grad_norm)
        self.opt_r.step()
        self.opt_c.step()
        self.opt_a.step()
        return loss_r.item(), loss_c.item(), loss_pi.item()

    def cup_second_stage_step(self, obs_b, act_b, logp_b, adv_c_b, old_mean_b, old_std_b, lagrangian_multiplier,
                              gamma, record=None):
        self.opt_a.zero_grad()
        loss = cup_second_stage_loss(self.policy, obs_b, act_b, logp_b, adv_c_b, old_mean_b, old_std_b,
                                     lagrangian_multiplier, gamma)
        loss.backward()
        if record is not None:
            record["grad_preclip"] = actor_flat_grads(self.policy.actor).clone()
        torch.nn.utils.clip_grad_norm_(self.policy.actor.parameters(), self.max_grad_norm)
        self.opt_a.step()
        return loss.item()

def _kl_stopped_passes(policy, data, perms, perm0, learning_iters, batch_size, target_kl, old_mean, old_std, step_fn):
    M = data["obs"].shape[0]
    losses, stop_iter, final_kl = [], 0, 1.0
    for it in range(learning_iters):
        perm = torch.as_tensor(perms[perm0 + it], dtype=torch.long)
        for s in range(0, M, batch_size):
            losses.append(step_fn(perm[s:s + batch_size]))
        final_kl = actor_kl(policy, data["obs"], old_mean, old_std)
        stop_iter += 1
        if final_kl > target_kl:
            break
    return losses, stop_iter, final_kl


def focops_update(policy, updater: KLPenaltyUpdater, data: dict, lam: float, perms, learning_iters: int = 40,
                  batch_size: int = 64, target_kl: float = 0.02, trace=None):
    """focops.py:279-367 with the shuffles supplied."""
    with torch.no_grad():
        old = policy.actor(data["obs"])
        old_mean, old_std = old.mean.clone(), old.stddev.clone()
    advantage = adv_mix(data["adv_r"], data["adv_c"], lam)            # focops.py:286-287

    def step(idx):
        rec = {} if trace is not None else None
        l = updater.focops_step(data["obs"][idx], data["act"][idx], data["log_prob"][idx], data["target_value_r"][idx],
                                data["target_value_c"][idx], advantage[idx], old_mean[idx], old_std[idx], target_kl,
                                record=rec)
        if trace is not None:
            trace.append(rec)
        return l
    losses, stop_iter, kl = _kl_stopped_passes(policy, data, perms, 0, learning_iters, batch_size, target_kl,
                                               old_mean, old_std, step)
    updater.sched.step()
    return {"losses": np.asarray(losses, np.float64), "stop_iter": stop_iter, "kl": kl}


def cup_update(policy, updater: KLPenaltyUpdater, data: dict, lam: float, perms, gamma: float,
               learning_iters: int = 40, batch_size: int = 64, target_kl: float = 0.02, trace=None):
    """cup.py:279-405 with the shuffles supplied (first-stage passes first, then the second stage's)."""
    with torch.no_grad():
        old = policy.actor(data["obs"])
        old_mean, old_std = old.mean.clone(), old.stddev.clone()

    def step1(idx):
        return updater.minibatch_step(data["obs"][idx], data["act"][idx], data["log_prob"][idx],
                                      data["target_value_r"][idx], data["target_value_c"][idx], data["adv_r"][idx])
    losses, stop_iter, kl1 = _kl_stopped_passes(policy, data, perms, 0, learning_iters, batch_size, target_kl,
                                                old_mean, old_std, step1)
    with torch.no_grad():
        old = policy.actor(data["obs"])
        old_mean, old_std = old.mean.clone(), old.stddev.clone()

    def step2(idx):
        rec = {} if trace is not None else None
        l = updater.cup_second_stage_step(data["obs"][idx], data["act"][idx], data["log_prob"][idx],
                                          data["adv_c"][idx], old_mean[idx], old_std[idx], lam, gamma, record=rec)
        if trace is not None:
            trace.append(rec)
        return l
    losses2, stop_iter2, kl2 = _kl_stopped_passes(policy, data, perms, stop_iter, learning_iters, batch_size, target_kl,
                                                  old_mean, old_std, step2)
    updater.sched.step()
    return {"losses": np.asarray(losses, np.float64), "stop_iter": stop_iter, "second_stage_losses": losses2,
            "second_stage_stop_iter": stop_iter2, "kl": kl2, "kl_first_stage": kl1}


# --------------------------------------------------------------------------
# a-2  observation normalisation (gymnasium NormalizeObservation; source not vendored
#      in /root/reference -> restated from its documented algorithm; PARITY UNPINNED)
# --------------------------------------------------------------------------
class RunningMeanStd:
    """gymnasium.wrappers.normalize.RunningMeanStd (gymnasium 0.28/0.29 line, named by
    safepo/common/wrappers.py:27): mean 0, var 1, count 1e-4; batch parallel-variance merge."""

    def __init__(self, shape, epsilon=1e-4):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, x):
        x = np.asarray(x, np.float64)
        b_mean, b_var, b_n = x.mean(axis=0), x.var(axis=0), x.shape[0]
        delta = b_mean - self.mean
        tot = self.count + b_n
        new_mean = self.mean + delta * b_n / tot
        m2 = self.var * self.count + b_var * b_n + np.square(delta) * self.count * b_n / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot

    def normalize(self, x, eps=1e-8):
        self.update(x)
        return (x - self.mean) / np.sqrt(self.var + eps)


# --------------------------------------------------------------------------
# a-13 .. a-17  CPO
# --------------------------------------------------------------------------
def actor_flat_params(actor) -> torch.Tensor:
    """safepo/single_agent/cpo.py:70-78 (named_parameters order: log_std first)."""
    return torch.cat([p.data.view(-1) for _, p in actor.named_parameters() if p.requires_grad])


def actor_set_flat_params(actor, vals: torch.Tensor) -> None:
    """cpo.py:109-121."""
    i = 0
    for _, p in actor.named_parameters():
        n = p.numel()
        p.data = vals[i:i + n].view(p.size())
        i += n
    assert i == len(vals)


def actor_flat_grads(actor) -> torch.Tensor:
    """cpo.py:123-130."""
    return torch.cat([p.grad.view(-1) for _, p in actor.named_parameters() if p.grad is not None])


def cpo_fvp(v: torch.Tensor, policy: OraclePolicy, obs) -> torch.Tensor:
    """cpo.py:132-157: Hessian of mean(KL(old||cur)) (mean over rows AND action dims)
    at cur==old, times v, + 0.1*v damping."""
    policy.actor.zero_grad()
    cur = policy.actor(obs)
    with torch.no_grad():
        old = policy.actor(obs)
    kl = torch.distributions.kl.kl_divergence(old, cur).mean()
    grads = torch.autograd.grad(kl, tuple(policy.actor.parameters()), create_graph=True)
    flat = torch.cat([g.view(-1) for g in grads])
    kl_p = (flat * v).sum()
    grads2 = torch.autograd.grad(kl_p, tuple(policy.actor.parameters()), retain_graph=False)
    return torch.cat([g.contiguous().view(-1) for g in grads2]) + v * 0.1


def cpo_cg(fvp_fn, b: torch.Tensor, num_steps: int = 15, residual_tol: float = 1e-10,
           eps: float = 1e-6) -> torch.Tensor:
    """cpo.py:81-106."""
    x = torch.zeros_like(b)
    r = b - fvp_fn(x)
    p = r.clone()
    rdotr = torch.dot(r, r)
    for _ in range(num_steps):
        z = fvp_fn(p)
        alpha = rdotr / (torch.dot(p, z) + eps)
        x += alpha * p
        r -= alpha * z
        new_rdotr = torch.dot(r, r)
        if torch.sqrt(new_rdotr) < residual_tol:
            break
        mu = new_rdotr / (rdotr + eps)
        p = r + mu * p
        rdotr = new_rdotr
    return x


def cpo_surrogate(policy: OraclePolicy, data: dict, which: str):
    """cpo.py:356-361 / 372-378: loss_pi_r = -mean(ratio*adv_r); loss_pi_c = mean(ratio*adv_c)."""
    dist = policy.actor(data["obs"])
    logp = dist.log_prob(data["act"]).sum(dim=-1)
    ratio = torch.exp(logp - data["log_prob"])
    if which == "r":
        return -(ratio * data["adv_r"]).mean()
    return (ratio * data["adv_c"]).mean()


def cpo_step_direction(x, p, g, b, xHx, ep_costs: float, target_kl: float):
    """cpo.py:384-463: q,r,s; A,B; case analysis; lambda*, nu*; step direction.
    All scalars are 0-dim fp32 torch tensors as in the reference."""
    q = xHx
    r = g.dot(p)
    s = b.dot(p)
    if b.dot(b) <= 1e-6 and ep_costs < 0:
        A = torch.zeros(1)
        B = torch.zeros(1)
        case = 4
    else:
        A = q - r ** 2 / (s + 1e-8)
        B = 2 * target_kl - ep_costs ** 2 / (s + 1e-8)
        if ep_costs < 0 and B < 0:
            case = 3
        elif ep_costs < 0 <= B:
            case = 2
        elif ep_costs >= 0 and B >= 0:
            case = 1
        else:
            case = 0
    if case in (3, 4):
        alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
        nu_star = torch.zeros(1)
        lambda_star = 1 / (alpha + 1e-8)
        step = alpha * x
    elif case in (1, 2):
        lambda_a = torch.sqrt(A / B)
        lambda_b = torch.sqrt(q / (2 * target_kl))
        r_num = r.item()
        eps_cost = ep_costs + 1e-8
        if ep_costs < 0:
            la = torch.clamp(lambda_a, torch.as_tensor(0.0), r_num / eps_cost)
            lb = torch.clamp(lambda_b, r_num / eps_cost, torch.as_tensor(torch.inf))
        else:
            la = torch.clamp(lambda_a, r_num / eps_cost, torch.as_tensor(torch.inf))
            lb = torch.clamp(lambda_b, torch.as_tensor(0.0), r_num / eps_cost)
        f_a = -0.5 * (A / (la + 1e-8) + B * la) - r * ep_costs / (s + 1e-8)
        f_b = -0.5 * (q / (lb + 1e-8) + 2 * target_kl * lb)
        lambda_star = la if f_a >= f_b else lb
        nu_star = torch.clamp(lambda_star * ep_costs - r, min=0) / (s + 1e-8)
        step = 1.0 / (lambda_star + 1e-8) * (x - nu_star * p)
    else:
        lambda_star = torch.zeros(1)
        nu_star = torch.sqrt(2 * target_kl / (s + 1e-8))
        step = -nu_star * p
    return {"case": case, "step": step, "q": q, "r": r, "s": s, "A": A, "B": B,
            "lambda_star": lambda_star, "nu_star": nu_star}


def cpo_line_search(policy, data, old_mean, old_std, theta_old, step_direction, g,
                    loss_r_before: float, loss_c_before: float, ep_costs: float, case: int,
                    target_kl: float, max_steps: int = 15, decay: float = 0.8):
    """cpo.py:465-519."""
    step_frac = 1.0
    old = torch.distributions.Normal(old_mean, old_std)
    kl = torch.zeros(1)
    accept = 0
    for step in range(max_steps):
        actor_set_flat_params(policy.actor, theta_old + step_frac * step_direction)
        accept = step + 1
        with torch.no_grad():
            loss_r = cpo_surrogate(policy, data, "r")
            loss_c = cpo_surrogate(policy, data, "c")
            kl = torch.distributions.kl.kl_divergence(old, policy.actor(data["obs"])).mean()
        improve = loss_r_before - loss_r.item()
        cost_diff = loss_c.item() - loss_c_before
        if not torch.isfinite(kl):
            continue                      # NB: reference does not decay step_frac here (cpo.py:498-500)
        if (improve < 0) if case > 1 else False:
            pass
        elif cost_diff > max(-ep_costs, 0):
            pass
        elif kl > target_kl:
            pass
        else:
            break
        step_frac *= decay
    else:
        step_direction = torch.zeros_like(step_direction)
        accept = 0
    theta_new = theta_old + step_frac * step_direction
    actor_set_flat_params(policy.actor, theta_new)
    return {"accept": accept, "step_frac": step_frac, "kl": float(kl), "theta_new": theta_new,
            "step_direction": step_direction}


def cpo_policy_update(policy: OraclePolicy, data: dict, ep_costs: float, target_kl: float = 0.01,
                      cg_iters: int = 15):
    """cpo.py:350-532 (actor part).  `ep_costs` = Jc - cost_limit."""
    obs = data["obs"]
    theta_old = actor_flat_params(policy.actor).clone()
    policy.actor.zero_grad()
    loss_pi_r = cpo_surrogate(policy, data, "r")
    loss_r_before = loss_pi_r.item()
    with torch.no_grad():
        od = policy.actor(obs)
        old_mean, old_std = od.mean.clone(), od.stddev.clone()
    loss_pi_r.backward()
    g = -actor_flat_grads(policy.actor)
    fv = lambda v: cpo_fvp(v, policy, obs)
    x = cpo_cg(fv, g, cg_iters)
    xHx = torch.dot(x, fv(x))
    alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
    policy.actor.zero_grad()
    loss_pi_c = cpo_surrogate(policy, data, "c")
    loss_c_before = loss_pi_c.item()
    loss_pi_c.backward()
    b = actor_flat_grads(policy.actor).clone()
    p = cpo_cg(fv, b, cg_iters)
    sd = cpo_step_direction(x, p, g, b, xHx, ep_costs, target_kl)
    ls = cpo_line_search(policy, data, old_mean, old_std, theta_old, sd["step"], g,
                         loss_r_before, loss_c_before, ep_costs, sd["case"], target_kl)
    out = {"g": g, "b": b, "x": x, "p": p, "xHx": xHx, "alpha": alpha,
           "loss_r_before": loss_r_before, "loss_c_before": loss_c_before}
    out.update(sd)
    out.update(ls)
    return out


def trust_region_policy_update(policy: OraclePolicy, data: dict, advantage: torch.Tensor, target_kl: float = 0.01,
                               line_search: bool = False, cg_iters: int = 15, search_steps: int = 15, decay: float = 0.8):
    """The unconstrained trust-region step of the f4 siblings (actor part):
    natural_pg.py:350-381 (line_search=False: theta_old + alpha * x, KL of the new policy logged) -- rcpo.py:320-326 runs the
    same on the Lagrangian mix of the advantages -- and trpo.py:366-428 (line_search=True: backtracking over TRPO_SEARCHING_STEPS
    = 15 candidates, accepted when the surrogate does not get worse and KL <= target_kl) -- trpo_lag.py:320-327 on the mix.
    `advantage`: data["adv_r"] or the mixed advantage, in the dtype of the policy."""
    obs = data["obs"]
    theta_old = actor_flat_params(policy.actor).clone()
    policy.actor.zero_grad()
    dist = policy.actor(obs)
    logp = dist.log_prob(data["act"]).sum(dim=-1)
    ratio = torch.exp(logp - data["log_prob"])
    loss_pi = -(ratio * advantage).mean()
    loss_before = loss_pi.item()
    with torch.no_grad():
        od = policy.actor(obs)
        old = torch.distributions.Normal(od.mean.clone(), od.stddev.clone())
    loss_pi.backward()
    g = -actor_flat_grads(policy.actor)
    fv = lambda v: cpo_fvp(v, policy, obs)
    x = cpo_cg(fv, g, cg_iters)
    xHx = torch.dot(x, fv(x))
    alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
    step_direction = x * alpha
    accept, step_frac, loss_actor, final_kl = None, 1.0, loss_before, 0.0
    if not line_search:
        actor_set_flat_params(policy.actor, theta_old + step_direction)
        with torch.no_grad():
            final_kl = torch.distributions.kl.kl_divergence(old, policy.actor(obs)).mean().item()
    else:
        accept = 0
        for step in range(search_steps):
            actor_set_flat_params(policy.actor, theta_old + step_frac * step_direction)
            with torch.no_grad():
                lp = policy.actor(obs).log_prob(data["act"]).sum(dim=-1)
                loss_new = -(torch.exp(lp - data["log_prob"]) * advantage).mean()
                kl = torch.distributions.kl.kl_divergence(old, policy.actor(obs)).mean().item()
            loss_actor = loss_new.item()
            improve = loss_before - loss_new.item()
            if not torch.isfinite(loss_new):
                pass
            elif improve < 0:
                pass
            elif kl > target_kl:
                pass
            else:
                accept = step + 1
                final_kl = kl
                break
            step_frac *= decay
        else:
            step_direction = torch.zeros_like(step_direction)
            accept = 0
        actor_set_flat_params(policy.actor, theta_old + step_frac * step_direction)
    return {"g": g, "x": x, "xHx": xHx, "alpha": alpha, "step_direction": step_direction, "step_frac": step_frac,
            "accept": accept, "kl": final_kl, "loss_before": loss_before, "loss_actor": loss_actor}


def pcpo_policy_update(policy: OraclePolicy, data: dict, ep_costs: float, target_kl: float = 0.01, cg_iters: int = 15,
                       search_steps: int = 200, decay: float = 0.8):
    """pcpo.py:352-470 (actor part): the reward step sqrt(2 delta / (q + 1e-8)) * (H x) -- the reference names fvp(x)
    "H_inv_g" (pcpo.py:370) -- projected onto the cost constraint, - max(0, (sqrt(2 delta / q) r + c) / s) p, then CPO's
    acceptance rules with optim_case = 0 over PCPO_SEARCHING_STEPS = 200 candidates (pcpo.py:44,407-458).  `ep_costs` = Jc -
    cost_limit."""
    obs = data["obs"]
    theta_old = actor_flat_params(policy.actor).clone()
    policy.actor.zero_grad()
    loss_pi_r = cpo_surrogate(policy, data, "r")
    loss_r_before = loss_pi_r.item()
    with torch.no_grad():
        od = policy.actor(obs)
        old_mean, old_std = od.mean.clone(), od.stddev.clone()
    loss_pi_r.backward()
    g = -actor_flat_grads(policy.actor)
    fv = lambda v: cpo_fvp(v, policy, obs)
    x = cpo_cg(fv, g, cg_iters)
    Hx = fv(x)
    xHx = torch.dot(x, Hx)
    alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
    policy.actor.zero_grad()
    loss_pi_c = cpo_surrogate(policy, data, "c")
    loss_c_before = loss_pi_c.item()
    loss_pi_c.backward()
    b = actor_flat_grads(policy.actor).clone()
    p = cpo_cg(fv, b, cg_iters)
    q, r, s_ = xHx, g.dot(p), b.dot(p)
    step = (torch.sqrt(2 * target_kl / (q + 1e-8)) * Hx
            - torch.clamp_min((torch.sqrt(2 * target_kl / q) * r + ep_costs) / s_, torch.zeros((), dtype=q.dtype)) * p)
    ls = cpo_line_search(policy, data, old_mean, old_std, theta_old, step, g, loss_r_before, loss_c_before, ep_costs, 0,
                         target_kl, max_steps=search_steps, decay=decay)
    out = {"g": g, "b": b, "x": x, "p": p, "Hx": Hx, "xHx": xHx, "alpha": alpha, "case": 0, "step": step,
           "loss_r_before": loss_r_before, "loss_c_before": loss_c_before}
    out.update(ls)
    return out


class CriticFitter:
    """cpo.py:534-571: two critics, Adam lr 1e-3, MSE + 0.001*L2, clip_grad_norm_ over ALL
    policy parameters (the actor's .grad still holds the stale cost gradient b and is
    rescaled in place by every clip whose coefficient is < 1)."""

    def __init__(self, policy: OraclePolicy, lr: float = 1e-3, max_grad_norm: float = 40.0):
        self.policy = policy
        self.opt_r = torch.optim.Adam(policy.reward_critic.parameters(), lr=lr)
        self.opt_c = torch.optim.Adam(policy.cost_critic.parameters(), lr=lr)
        self.max_grad_norm = max_grad_norm

    def minibatch_step(self, obs_b, tgt_r_b, tgt_c_b):
        self.opt_r.zero_grad()
        self.opt_c.zero_grad()
        loss_r = torch.nn.functional.mse_loss(self.policy.reward_critic(obs_b), tgt_r_b)
        loss_c = torch.nn.functional.mse_loss(self.policy.cost_critic(obs_b), tgt_c_b)
        for prm in self.policy.reward_critic.parameters():
            loss_r = loss_r + prm.pow(2).sum() * 0.001
        for prm in self.policy.cost_critic.parameters():
            loss_c = loss_c + prm.pow(2).sum() * 0.001
        (loss_r + loss_c).backward()
        torch.nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
        self.opt_r.step()
        self.opt_c.step()
        return loss_r.item(), loss_c.item()


# --------------------------------------------------------------------------
# Whole-epoch port of the reference main loop: used (a) to pin the restatement against
# traces of the real reference main(), (b) as bench.py's cpu_baseline ("port").
# It keeps the reference's data structures and per-env / per-sample Python loops, because
# those loops ARE the reference CPU cost being measured.
# --------------------------------------------------------------------------
class PerEnvBuffer:
    """safepo/common/buffer.py:40-164 (per-env tensors, python pointer lists)."""

    KEYS = ("obs", "act", "reward", "cost", "done", "value_r", "value_c", "adv_r", "adv_c",
            "target_value_r", "target_value_c", "log_prob")

    def __init__(self, obs_dim, act_dim, size, num_envs, gamma=0.99, lam=0.95, lam_c=0.95):
        def fresh():
            d = {k: torch.zeros(size, dtype=torch.float32) for k in self.KEYS}
            d["obs"] = torch.zeros((size, obs_dim), dtype=torch.float32)
            d["act"] = torch.zeros((size, act_dim), dtype=torch.float32)
            return d
        self.envs = [fresh() for _ in range(num_envs)]
        self.gamma, self.lam, self.lam_c = gamma, lam, lam_c
        self.ptr = [0] * num_envs
        self.start = [0] * num_envs
        self.size = size

    def store(self, **data):
        for i, e in enumerate(self.envs):
            assert self.ptr[i] < self.size, "Buffer overflow"
            for k, v in data.items():
                e[k][self.ptr[i]] = v[i]
            self.ptr[i] += 1

    def finish_path(self, last_r, last_c, idx):
        e = self.envs[idx]
        sl = slice(self.start[idx], self.ptr[idx])
        for rk, vk, ak, tk, last, lm in (("reward", "value_r", "adv_r", "target_value_r", last_r, self.lam),
                                         ("cost", "value_c", "adv_c", "target_value_c", last_c, self.lam_c)):
            rew = torch.cat([e[rk][sl], last])
            val = torch.cat([e[vk][sl], last])
            deltas = rew[:-1] + self.gamma * val[1:] - val[:-1]
            x = deltas.type(torch.float64)
            disc = self.gamma * lm
            c = x[-1]
            for i in reversed(range(x.shape[0] - 1)):
                c = x[i] + disc * c
                x[i] = c
            e[ak][sl] = x
            e[tk][sl] = x + val[:-1]
        self.start[idx] = self.ptr[idx]

    def get(self):
        data = {k: torch.cat([e[k] for e in self.envs], dim=0) for k in self.KEYS}
        data["adv_r"], data["adv_c"] = adv_standardize(data["adv_r"], data["adv_c"])
        n = len(self.envs)
        self.ptr, self.start = [0] * n, [0] * n
        return data


class StatsLog:
    """The slice of EpochLogger the path reads back: store / get_stats / dump
    (safepo/common/logger.py:344-373)."""

    def __init__(self):
        self.cur = {}
        self.headers = set()

    def store(self, **kw):
        for k, v in kw.items():
            self.cur.setdefault(k, []).append(v)

    def get_stats(self, key):
        if key not in self.headers:
            return 0.0
        return np.mean(self.cur[key])

    def dump(self):
        row = {k: (float(np.mean(v)) if len(v) else float("nan")) for k, v in self.cur.items()}
        self.headers.update(self.cur.keys())
        self.cur = {k: [] for k in self.cur}
        return row


def collect_epoch_port(env, policy: OraclePolicy, obs, num_envs: int, local_steps: int, stats: StatsLog, deques, ep_acc,
                       gamma: float):
    """The rollout half of one epoch, identical in safepo/single_agent/ppo_lag.py:159-234 and cpo.py:240-313: batched
    policy step, per-env Python store loop, per-env boundary scan with single-row bootstrap forwards and per-path GAE.
    Returns (next obs, filled PerEnvBuffer)."""
    obs_dim, act_dim = obs.shape[-1], policy.actor.log_std.shape[0]
    buf = PerEnvBuffer(obs_dim, act_dim, local_steps, num_envs, gamma=gamma)
    rew_dq, cost_dq, len_dq = deques
    ep_ret, ep_cost, ep_len = ep_acc
    for step in range(local_steps):
        with torch.no_grad():
            dist = policy.actor(obs)
            act = dist.rsample()
            logp = dist.log_prob(act).sum(axis=-1)
            v_r, v_c = policy.reward_critic(obs), policy.cost_critic(obs)
        nobs, rew, cost, term, trunc, info = env.step(act.detach().squeeze().cpu().numpy())
        ep_ret += rew
        ep_cost += cost
        ep_len += 1
        nobs, rew, cost, term, trunc = (torch.as_tensor(x, dtype=torch.float32)
                                        for x in (nobs, rew, cost, term, trunc))
        if "final_observation" in info:
            fo = np.array([a if a is not None else np.zeros(obs_dim) for a in info["final_observation"]])
            info["final_observation"] = torch.as_tensor(fo, dtype=torch.float32)
        buf.store(obs=obs, act=act, reward=rew, cost=cost, value_r=v_r, value_c=v_c, log_prob=logp)
        obs = nobs
        epoch_end = step >= local_steps - 1
        for idx, (done, time_out) in enumerate(zip(term, trunc)):
            if epoch_end or done or time_out:
                last_r = torch.zeros(1)
                last_c = torch.zeros(1)
                if not done:
                    if epoch_end:
                        with torch.no_grad():
                            _, _, last_r, last_c = policy.step_with_eps(obs[idx], torch.randn(act_dim))
                    if time_out:
                        with torch.no_grad():
                            _, _, last_r, last_c = policy.step_with_eps(
                                info["final_observation"][idx], torch.randn(act_dim))
                    last_r, last_c = last_r.unsqueeze(0), last_c.unsqueeze(0)
                if done or time_out:
                    rew_dq.append(ep_ret[idx])
                    cost_dq.append(ep_cost[idx])
                    len_dq.append(ep_len[idx])
                    stats.store(**{"Metrics/EpRet": np.mean(rew_dq), "Metrics/EpCost": np.mean(cost_dq),
                                   "Metrics/EpLen": np.mean(len_dq)})
                    ep_ret[idx] = ep_cost[idx] = ep_len[idx] = 0.0
                buf.finish_path(last_r, last_c, idx)
    return obs, buf


def ppo_lag_epoch_port(env, policy: OraclePolicy, updater: PPOLagUpdater, lagrange: OracleLagrange,
                       obs, num_envs: int, local_steps: int, stats: StatsLog, deques, ep_acc,
                       cfg: dict, timers: dict | None = None):
    """One epoch of safepo/single_agent/ppo_lag.py:159-350 (eval off), torch CPU.
    Returns (next obs, per-epoch info)."""
    import time
    from torch.utils.data import DataLoader, TensorDataset
    t0 = time.time()
    obs, buf = collect_epoch_port(env, policy, obs, num_envs, local_steps, stats, deques, ep_acc, cfg["gamma"])
    t1 = time.time()
    lagrange.update_lagrange_multiplier(stats.get_stats("Metrics/EpCost"))
    data = buf.get()
    old = policy.actor(data["obs"])
    lam = lagrange.lagrangian_multiplier
    advantage = data["adv_r"] - lam * data["adv_c"]
    advantage /= (lam + 1)
    loader = DataLoader(TensorDataset(data["obs"], data["act"], data["log_prob"], data["target_value_r"],
                                      data["target_value_c"], advantage),
                        batch_size=cfg["batch_size"], shuffle=True)
    stop_iter, kl = 0, 1.0
    for _ in range(cfg["learning_iters"]):
        for ob, ab, lb, trb, tcb, advb in loader:
            lr_, lc_, lp_ = updater.minibatch_step(ob, ab, lb, trb, tcb, advb)
            stats.store(**{"Loss/Loss_reward_critic": lr_, "Loss/Loss_cost_critic": lc_,
                           "Loss/Loss_actor": lp_})
        new = policy.actor(data["obs"])
        kl = torch.distributions.kl.kl_divergence(old, new).sum(-1, keepdim=True).mean().item()
        stop_iter += 1
        if kl > cfg["target_kl"]:
            break
    t2 = time.time()
    updater.sched.step()
    if timers is not None:
        timers["rollout"] = t1 - t0
        timers["update"] = t2 - t1
    return obs, {"stop_iter": stop_iter, "kl": kl, "lambda": lam, "data": data}


def cpo_epoch_port(env, policy: OraclePolicy, fitter: CriticFitter, obs, num_envs: int, local_steps: int,
                   stats: StatsLog, deques, ep_acc, cfg: dict, cost_limit: float = 25.0, timers: dict | None = None):
    """One epoch of safepo/single_agent/cpo.py:240-571 (eval off), torch CPU: the shared rollout, then the trust-region
    actor update (two surrogate gradients, two 15-step CG solves = 33 Fisher-vector products by double backward, case
    analysis, line search) and the critic fit through a shuffling DataLoader (batch 128 x 10 iterations)."""
    import time
    from torch.utils.data import DataLoader, TensorDataset
    t0 = time.time()
    obs, buf = collect_epoch_port(env, policy, obs, num_envs, local_steps, stats, deques, ep_acc, cfg["gamma"])
    t1 = time.time()
    data = buf.get()
    ep_costs = stats.get_stats("Metrics/EpCost") - cost_limit
    out = cpo_policy_update(policy, data, ep_costs, target_kl=cfg.get("target_kl", 0.01), cg_iters=cfg.get("cg_iters", 15))
    loader = DataLoader(TensorDataset(data["obs"], data["target_value_r"], data["target_value_c"]),
                        batch_size=cfg["batch_size"], shuffle=True)
    for _ in range(cfg["learning_iters"]):
        for ob, trb, tcb in loader:
            lr_, lc_ = fitter.minibatch_step(ob, trb, tcb)
            stats.store(**{"Loss/Loss_reward_critic": lr_, "Loss/Loss_cost_critic": lc_})
    t2 = time.time()
    if timers is not None:
        timers["rollout"] = t1 - t0
        timers["update"] = t2 - t1
    return obs, {"case": out["case"], "accept_step": out.get("accept"), "data": data}
