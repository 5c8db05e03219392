"""MAPPO-Lagrangian (multi-agent): reference safepo/multi_agent/mappolag.py, SURVEY.md 8 f3.

Same surface -- MAPPO_L_Policy, MAPPO_L_Trainer, Runner, train(args, cfg_train) -- on the MI355X kernels of
csrc/ma_net.hip and csrc/multi_agent.hip: per-agent actor / critic / cost-critic networks are flat device vectors whose
forward and backward run on the in-tree fp32 MFMA GEMM kernels with fused LayerNorm/ELU kernels; the clipped HAPPO surrogate, entropy bonus,
in-loop multiplier step, PopArt statistics, clipped Huber value losses and clip_grad_norm_ + Adam are single kernels;
GAE + PopArt de-normalisation for rewards and costs is one kernel per agent (spo_ma_gae).  Everything between the
environment and the logger stays in HBM; there is no CPU fallback.

Regime note: num_mini_batch = 1 and learning_iters = 5, so one epoch is 5 full-batch steps per network per agent over
episode_length x n_rollout_threads rows -- large GEMMs, the opposite of the single-agent path's 327 680 tiny steps.
"""
from __future__ import annotations

import copy
import os
import sys
import time

import numpy as np
import torch

from safepo import _abi
from safepo.common.buffer import SeparatedReplayBuffer
from safepo.common.logger import EpochLogger
from safepo.common.model import MultiAgentActor as Actor, MultiAgentCritic as Critic
from safepo.common.popart import PopArt
from safepo.parallel import Comm, init_from_env

# Defaults of the reference's marl_cfg/mappolag/config.yaml, and its `mamujoco` overrides (applied for the MuJoCo
# velocity / multi-goal tasks, safepo/utils/config.py:236-241).
default_cfg = dict(
    env_name="mappolag", algorithm_name="mappolag", experiment_name="check", seed=0, run_dir="./runs/",
    num_env_steps=100000000, episode_length=8, n_rollout_threads=1, n_eval_rollout_threads=1, hidden_size=512,
    use_render=False, recurrent_N=1, use_single_network=False, save_interval=1, use_eval=False, eval_interval=25,
    log_interval=25, eval_episodes=10000, cost_limit=25, lagrangian_coef_rate=1.0e-5, lamda_lagr=0.78, gamma=0.96,
    gae_lambda=0.95, use_gae=True, use_popart=True, use_valuenorm=True, use_proper_time_limits=False, target_kl=0.016,
    searching_steps=10, accept_ratio=0.5, clip_param=0.2, learning_iters=5, num_mini_batch=1, data_chunk_length=None,
    value_loss_coef=1, entropy_coef=0.0, max_grad_norm=10, huber_delta=10.0, use_recurrent_policy=False,
    use_naive_recurrent_policy=False, use_max_grad_norm=True, use_clipped_value_loss=True, use_huber_loss=True,
    use_value_active_masks=False, use_policy_active_masks=False, actor_lr=9.0e-5, critic_lr=5.0e-3, opti_eps=1.0e-5,
    weight_decay=0.0, gain=0.01, actor_gain=0.01, use_orthogonal=True, use_feature_normalization=True, use_ReLU=True,
    stacked_frames=1, layer_N=2, std_x_coef=1, std_y_coef=0.5)
mamujoco_cfg = dict(
    num_env_steps=10000000, episode_length=1000, n_rollout_threads=10, n_eval_rollout_threads=10, hidden_size=128,
    gamma=0.99, entropy_coef=0.01, actor_lr=5.0e-4, critic_lr=5.0e-4, max_grad_norm=10.0, use_value_active_masks=True,
    use_policy_active_masks=True, data_chunk_length=10)


def check(x):
    return torch.from_numpy(x) if type(x) == np.ndarray else x


class _Adam:
    """Flat optimiser state of one network; `step` = clip_grad_norm_ + torch.optim.Adam through spo_ma_clip_adam."""

    def __init__(self, net, lr, eps, weight_decay):
        self.net, self.lr, self.eps, self.wd = net, float(lr), float(eps), float(weight_decay)
        self.m, self.v = torch.zeros_like(net.theta), torch.zeros_like(net.theta)
        self.grad = torch.zeros_like(net.theta)
        self.t = 0
        self.norm = torch.zeros(1, dtype=torch.float32, device=net.theta.device)
        self.partial = torch.zeros(1024, dtype=torch.float64, device=net.theta.device)

    def zero_grad(self):
        self.grad.zero_()

    def step(self, max_grad_norm, use_max_grad_norm=True, comm=None):
        th = self.net.theta
        if comm is not None and comm.world_size > 1:
            comm.all_reduce_sum_(self.grad)        # every rank holds its share of the global-batch gradient
        _abi.check(_abi.load().spo_ma_clip_adam(_abi.ptr(th), _abi.ptr(self.grad), _abi.ptr(self.m), _abi.ptr(self.v),
                                                th.numel(), self.t, self.lr, self.eps, self.wd, float(max_grad_norm),
                                                int(bool(use_max_grad_norm)), _abi.ptr(self.norm), _abi.ptr(self.partial),
                                                _abi.stream_ptr()), "spo_ma_clip_adam")
        self.t += 1
        return self.norm.clone().reshape(())


class MAPPO_L_Policy:
    """mappolag.py:45-113: actor on the agent's observation, critic and cost critic on the shared observation.
    `use_cost = False` (HAPPO_Policy / MAPPO_Policy, happo.py:46-93, mappo.py:46-93) drops the cost critic."""
    use_cost = True

    def __init__(self, config, obs_space, cent_obs_space, act_space):
        self.config, self.obs_space, self.act_space, self.share_obs_space = config, obs_space, act_space, cent_obs_space
        dev = torch.device(config["device"])
        if dev.type != "cuda":
            raise _abi.SpoError("MAPPO-L (MI355X) runs on a ROCm GPU only (--device cuda); there is no CPU fallback")
        self.actor = Actor(config, obs_space, act_space, dev)
        self.critic = Critic(config, cent_obs_space, dev)
        self.cost_critic = Critic(config, cent_obs_space, dev) if self.use_cost else None
        self.actor_optimizer = _Adam(self.actor, config["actor_lr"], config["opti_eps"], config["weight_decay"])
        self.critic_optimizer = _Adam(self.critic, config["critic_lr"], config["opti_eps"], config["weight_decay"])
        self.cost_optimizer = (_Adam(self.cost_critic, config["critic_lr"], config["opti_eps"], config["weight_decay"])
                               if self.use_cost else None)

    def networks(self):
        return [n for n in (self.actor, self.critic, self.cost_critic) if n is not None]

    def get_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, masks, available_actions=None,
                    deterministic=False, rnn_states_cost=None):
        actions, action_log_probs, rnn_states_actor = self.actor(obs, rnn_states_actor, masks, available_actions, deterministic)
        values, rnn_states_critic = self.critic(cent_obs, rnn_states_critic, masks)
        if rnn_states_cost is None:
            return values, actions, action_log_probs, rnn_states_actor, rnn_states_critic
        cost_preds, rnn_states_cost = self.cost_critic(cent_obs, rnn_states_cost, masks)
        return values, actions, action_log_probs, rnn_states_actor, rnn_states_critic, cost_preds, rnn_states_cost

    def get_values(self, cent_obs, rnn_states_critic, masks):
        return self.critic(cent_obs, rnn_states_critic, masks)[0]

    def get_cost_values(self, cent_obs, rnn_states_cost, masks):
        return self.cost_critic(cent_obs, rnn_states_cost, masks)[0]

    def evaluate_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, action, masks, available_actions=None,
                         active_masks=None, rnn_states_cost=None):
        action_log_probs, dist_entropy = self.actor.evaluate_actions(obs, rnn_states_actor, action, masks, available_actions,
                                                                     active_masks)
        values, _ = self.critic(cent_obs, rnn_states_critic, masks)
        if rnn_states_cost is None:
            return values, action_log_probs, dist_entropy
        cost_values, _ = self.cost_critic(cent_obs, rnn_states_cost, masks)
        return values, action_log_probs, dist_entropy, cost_values

    def act(self, obs, rnn_states_actor, masks, available_actions=None, deterministic=False):
        actions, _, rnn_states_actor = self.actor(obs, rnn_states_actor, masks, available_actions, deterministic)
        return actions, rnn_states_actor


class MAPPO_L_Trainer:
    """mappolag.py:115-249.  `lamda_lagr` is a device scalar updated inside every minibatch step (mappolag.py:178-182).
    `algo` selects the sibling trainers built on the same kernels: "happo" (happo.py:96-205: no cost side, value loss with
    use_value_active_masks) and "mappo" (mappo.py:96-189: additionally per-dimension ratios and no sequential factor)."""
    algo = "mappolag"

    def __init__(self, config, policy, comm: Comm | None = None):
        self.config, self.policy = config, policy
        self.comm = comm or Comm()
        self.dev = torch.device(config["device"])
        self.tpdv = dict(dtype=torch.float32, device=self.dev)
        self.value_normalizer = PopArt(1, device=self.dev)
        self._popart_state = torch.zeros(3, **self.tpdv)         # {running_mean, running_mean_sq, debiasing_term}
        self.use_cost = self.algo == "mappolag"
        self._lamda = torch.tensor([float(config["lamda_lagr"]) if self.use_cost else 0.0], **self.tpdv)
        self._partial = torch.zeros(1024 * (4 + 16), dtype=torch.float64, device=self.dev)
        self._scalars = torch.zeros(5, **self.tpdv)
        self._sums2 = torch.zeros(2, dtype=torch.float64, device=self.dev)
        self._loss_cfg = _abi.MaLossCfg(clip_param=float(config["clip_param"]), entropy_coef=float(config["entropy_coef"]),
                                        std_x_coef=float(config["std_x_coef"]), std_y_coef=float(config["std_y_coef"]),
                                        use_policy_active_masks=int(bool(config["use_policy_active_masks"])),
                                        per_dim_ratio=int(self.algo == "mappo"))
        self._zeros = self._ones = None

    @property
    def lamda_lagr(self):
        return self._lamda[0]

    # PopArt statistics live in one device vector for the kernels; the nn.Module view is refreshed from it
    def _sync_normalizer(self):
        vn, s = self.value_normalizer, self._popart_state
        vn.running_mean.copy_(s[0:1]); vn.running_mean_sq.copy_(s[1:2]); vn.debiasing_term.copy_(s[2])

    def _normalize_returns(self, returns, out):
        vn, lib, st = self.value_normalizer, _abi.load(), _abi.stream_ptr()
        rows = returns.numel()
        _abi.check(lib.spo_ma_popart_stats(_abi.ptr(returns), rows, _abi.ptr(self._sums2), _abi.ptr(self._partial), st),
                   "spo_ma_popart_stats")
        self.comm.all_reduce_sum_(self._sums2)                  # batch mean / mean of squares of the GLOBAL batch
        _abi.check(lib.spo_ma_popart_forward(_abi.ptr(returns), rows, _abi.ptr(self._popart_state), float(vn.beta), float(vn.epsilon),
                                             1, _abi.ptr(self._sums2), rows * self.comm.world_size, _abi.ptr(out), st),
                   "spo_ma_popart_forward")

    def _normed_returns(self, returns):
        """value_normalizer(return_batch) twice, as cal_value_loss does (two statistics updates, in this order)."""
        returns = returns.reshape(-1).contiguous()
        n1, n2 = torch.empty_like(returns), torch.empty_like(returns)
        self._normalize_returns(returns, n1)                    # value_normalizer(return_batch) for error_clipped ...
        self._normalize_returns(returns, n2)                    # ... and again for error_original: two statistics updates
        return n1, n2

    def _value_step(self, net, opt, inputs, value_preds, returns, active=None, active_sum=None, normed=None, partial=None):
        """cal_value_loss (mappolag.py:126-138; happo.py:106-122 with `active`) + backward + clip + Adam for one critic.
        normed: the two PopArt-normalised copies of `returns` when the caller formed them already (side-stream form)."""
        c, lib = self.config, _abi.load()
        values, saved = net.net_forward(inputs, keep=True)
        rows = values.shape[0]
        n1, n2 = normed if normed is not None else self._normed_returns(returns)
        partial = self._partial if partial is None else partial
        dvalues, loss = torch.empty_like(values), torch.empty(1, **self.tpdv)
        denom = float(active_sum) if active is not None else float(rows * self.comm.world_size)
        _abi.check(lib.spo_ma_value_loss(_abi.ptr(values), _abi.ptr(value_preds.reshape(-1).contiguous()), _abi.ptr(n1), _abi.ptr(n2),
                                         _abi.ptr(active) if active is not None else None, denom,
                                         float(c["clip_param"]), float(c["huber_delta"]), float(c["value_loss_coef"]), rows,
                                         rows * self.comm.world_size, _abi.ptr(dvalues), _abi.ptr(loss), _abi.ptr(partial),
                                         _abi.stream_ptr()), "spo_ma_value_loss")
        self.comm.all_reduce_sum_(loss)
        net.net_backward(saved, dvalues, opt.grad)
        norm = opt.step(c["max_grad_norm"], c["use_max_grad_norm"], self.comm)
        return loss.reshape(()), norm

    def ppo_update(self, sample):
        if self.use_cost:
            (share_obs_batch, obs_batch, _rnn, _rnn_c, actions_batch, value_preds_batch, return_batch, _masks, active_masks_batch,
             old_action_log_probs_batch, adv_targ, _avail, factor_batch, cost_preds_batch, cost_returns_batch, _rnn_k,
             cost_adv_targ, aver_episode_costs) = sample
        else:       # happo.py:124-127 / mappo.py:119-121: the 13-tuple
            (share_obs_batch, obs_batch, _rnn, _rnn_c, actions_batch, value_preds_batch, return_batch, _masks, active_masks_batch,
             old_action_log_probs_batch, adv_targ, _avail, factor_batch) = sample[:13]
        c, lib, pol = self.config, _abi.load(), self.policy
        f = lambda t: _abi.require_gpu_tensor(check(t).to(**self.tpdv).contiguous(), "sample", torch.float32)
        obs_batch, share_obs_batch, actions_batch = f(obs_batch), f(share_obs_batch), f(actions_batch)
        side = None
        if self.use_cost and c.get("train_streams", True) and self.comm.world_size == 1:
            # The three networks of an update are independent (mappolag.py:140-199 runs them one after the other): the two
            # critics go to two side streams while the actor runs here, so the small reduction kernels of one network hide
            # behind the block kernels of another.  The four PopArt statistics updates keep their order: formed first, here.
            main = torch.cuda.current_stream()
            if getattr(self, "_side", None) is None:
                self._side = (torch.cuda.Stream(), torch.cuda.Stream())
                self._side_partial = (torch.zeros_like(self._partial), torch.zeros_like(self._partial))
            vp, rb, cpb, crb = f(value_preds_batch), f(return_batch), f(cost_preds_batch), f(cost_returns_batch)
            normed = (self._normed_returns(rb), self._normed_returns(crb))
            side = []
            for k, (net, opt, preds) in enumerate(((pol.critic, pol.critic_optimizer, vp), (pol.cost_critic, pol.cost_optimizer, cpb))):
                st = self._side[k]
                st.wait_stream(main)
                for t in (share_obs_batch, preds) + normed[k]:
                    t.record_stream(st)
                with torch.cuda.stream(st):
                    side.append(self._value_step(net, opt, share_obs_batch, preds, None, normed=normed[k],
                                                 partial=self._side_partial[k]))
        old_lp, adv, active = f(old_action_log_probs_batch), f(adv_targ).reshape(-1), f(active_masks_batch).reshape(-1)
        if self._zeros is None or self._zeros.numel() != adv.numel():
            self._zeros, self._ones = torch.zeros_like(adv), torch.ones_like(adv)
        cadv = f(cost_adv_targ).reshape(-1) if self.use_cost else self._zeros       # lamda is 0 as well: A - 0 * 0
        factor = self._ones if self.algo == "mappo" else f(factor_batch).reshape(-1)
        rows, A = obs_batch.shape[0], pol.actor.act_dim
        # ---- actor: clipped HAPPO surrogate on the hybrid advantage, entropy bonus (mappolag.py:150-176)
        mean, saved = pol.actor.net_forward(obs_batch, keep=True)
        dmean = torch.empty_like(mean)
        opt = pol.actor_optimizer
        ls_off = pol.actor.offset(6)
        rows_g = rows * self.comm.world_size                 # data parallel over rollout threads: equal shards
        hoisted = getattr(self, "_train_scope", None)         # host scalars train() formed once for all its updates
        if c["use_policy_active_masks"]:
            if hoisted is not None and "active_sum" in hoisted and hoisted["rows"] == rows:
                denom = hoisted["active_sum"]
            else:
                asum = active.sum().reshape(1).double()
                self.comm.all_reduce_sum_(asum)
                denom = float(asum.item())
        else:
            denom = float(rows_g)
        _abi.check(lib.spo_ma_actor_loss(_abi.ptr(mean), _abi.ptr(pol.actor.log_std), _abi.ptr(actions_batch), _abi.ptr(old_lp),
                                         _abi.ptr(adv), _abi.ptr(cadv), _abi.ptr(factor), _abi.ptr(active), _abi.ptr(self._lamda),
                                         self._loss_cfg, rows, A, denom, rows_g, _abi.ptr(dmean), _abi.ptr(opt.grad[ls_off:ls_off + A]),
                                         _abi.ptr(self._scalars), _abi.ptr(self._partial), _abi.stream_ptr()), "spo_ma_actor_loss")
        if self.comm.world_size > 1:
            ent = self._scalars[1].clone()                   # the entropy of a state-independent sigma is not a sum over rows
            self.comm.all_reduce_sum_(self._scalars)
            self._scalars[1] = ent
        pol.actor.net_backward(saved, dmean, opt.grad)
        actor_grad_norm = opt.step(c["max_grad_norm"], c["use_max_grad_norm"], self.comm)
        scal = self._scalars.clone()
        policy_loss, dist_entropy, imp_mean = scal[0], scal[1], scal[2]
        if not self.use_cost:
            masked = self.algo == "happo" and c["use_value_active_masks"]
            if masked and not c["use_policy_active_masks"]:
                asum = active.sum().reshape(1).double()
                self.comm.all_reduce_sum_(asum)
                denom = float(asum.item())
            value_loss, critic_grad_norm = self._value_step(pol.critic, pol.critic_optimizer, share_obs_batch, f(value_preds_batch),
                                                            f(return_batch), active if masked else None, denom if masked else None)
            self._sync_normalizer()
            return value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_mean
        # ---- multiplier (mappolag.py:178-182); aver_episode_costs.mean() is a host scalar of the buffer
        aver = (hoisted["aver_cost"] if hoisted is not None and "aver_cost" in hoisted
                else float(check(aver_episode_costs).float().mean().item()))
        _abi.check(lib.spo_ma_lamda_update(_abi.ptr(self._lamda), _abi.ptr(self._scalars), aver, float(c["cost_limit"]), float(c["gamma"]),
                                           float(c["lagrangian_coef_rate"]), _abi.stream_ptr()), "spo_ma_lamda_update")
        # ---- critics (mappolag.py:183-197); both share the one PopArt normaliser
        if side is not None:
            for st in self._side:
                torch.cuda.current_stream().wait_stream(st)
            (value_loss, critic_grad_norm), (cost_loss, cost_grad_norm) = side
        else:
            value_loss, critic_grad_norm = self._value_step(pol.critic, pol.critic_optimizer, share_obs_batch, f(value_preds_batch), f(return_batch))
            cost_loss, cost_grad_norm = self._value_step(pol.cost_critic, pol.cost_optimizer, share_obs_batch, f(cost_preds_batch), f(cost_returns_batch))
        self._sync_normalizer()
        return value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_mean, cost_loss, cost_grad_norm

    def train(self, buffer, logger, perm_fn=None):
        """mappolag.py:201-236 (advantage standardisation with torch.mean / torch.std over the NaN-masked copy, as written)."""
        c = self.config
        self._sync_normalizer()
        eps = 1e-8 if self.use_cost else 1e-5            # happo.py:171-175 / mappo.py:163-167: no NaN masking, + 1e-5

        def standardised(returns, preds):
            adv = returns[:-1] - self.value_normalizer.denormalize(preds[:-1])
            cp = adv
            if self.use_cost:          # (torch.where, not boolean-index assignment: that one runs nonzero() = a host sync)
                cp = torch.where(buffer.active_masks[:-1] == 0.0, torch.full_like(adv, float("nan")), adv)
            if self.comm.world_size == 1:
                return (adv - torch.mean(cp)) / (torch.std(cp) + eps)
            # torch.mean / torch.std (unbiased) of the rows of ALL ranks, NaN-propagating like the single-rank form
            d = cp.double()
            sums = torch.stack([d.sum(), (d * d).sum(), torch.tensor(float(d.numel()), dtype=torch.float64, device=d.device)])
            self.comm.all_reduce_sum_(sums)
            mean = sums[0] / sums[2]
            var = (sums[1] - sums[2] * mean * mean) / (sums[2] - 1.0)
            return (adv - mean.float()) / (torch.sqrt(var.clamp(min=0.0)).float() + eps)
        advantages = standardised(buffer.returns, buffer.value_preds)
        cost_adv = standardised(buffer.cost_returns, buffer.cost_preds) if self.use_cost else None
        # Host scalars every update of this call needs, formed ONCE (each .item() drains the queue: the GPU then idles until
        # the host has launched the next kernels).  With one minibatch per iteration the minibatch is the whole buffer in
        # some order, so the active-mask sum is the buffer's; the episode-cost average does not change during training.
        scope = {}
        if self.use_cost:
            scope["aver_cost"] = float(check(buffer.aver_episode_costs).float().mean().item())
        if c["num_mini_batch"] == 1 and c["use_policy_active_masks"]:
            asum = buffer.active_masks[:-1].sum().reshape(1).double()
            self.comm.all_reduce_sum_(asum)
            scope["active_sum"] = float(asum.item())
            scope["rows"] = int(buffer.active_masks[:-1].numel())
        self._train_scope = scope
        out, rows_logged = None, []
        try:
            for it in range(c["learning_iters"]):
                perm = perm_fn(it) if perm_fn is not None else None
                for sample in buffer.feed_forward_generator(advantages, c["num_mini_batch"], cost_adv=cost_adv, perm=perm):
                    out = self.ppo_update(sample)
                if logger is None:
                    continue
                # the iteration's log row stays on the device; all rows come to the host in one copy after the last update
                if not self.use_cost:
                    value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_weights = out
                    rows_logged.append(torch.stack([t.detach().reshape(()).float() for t in
                                                    (value_loss, policy_loss, critic_grad_norm, dist_entropy, imp_weights.detach().mean())]))
                else:
                    value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_weights, cost_loss, cost_grad_norm = out
                    rows_logged.append(torch.stack([t.detach().reshape(()).float() for t in
                                                    (value_loss, cost_loss, policy_loss, critic_grad_norm, cost_grad_norm, dist_entropy,
                                                     imp_weights.detach().mean())]))
        finally:
            self._train_scope = None
        if rows_logged:
            keys = (("Loss/Loss_reward_critic", "Loss/Loss_actor", "Misc/Reward_critic_norm", "Misc/Entropy", "Misc/Ratio")
                    if not self.use_cost else
                    ("Loss/Loss_reward_critic", "Loss/Loss_cost_critic", "Loss/Loss_actor", "Misc/Reward_critic_norm",
                     "Misc/Cost_critic_norm", "Misc/Entropy", "Misc/Ratio"))
            for row in torch.stack(rows_logged).tolist():
                logger.store(**dict(zip(keys, row)))
        return out

    def prep_training(self):
        pass        # no dropout / batch-norm in these networks: train() / eval() modes are identical

    def prep_rollout(self):
        pass


class Runner:
    """mappolag.py:252-604: collect -> insert -> compute -> train with sequential (HAPPO) agent updates.  The happo / mappo
    runners (happo.py:208-540, mappo.py:192-532) are this class with `policy_cls` / `trainer_cls` swapped: no cost
    predictions in collect / insert / compute, the shorter log table."""
    policy_cls = MAPPO_L_Policy
    trainer_cls = MAPPO_L_Trainer
    log_keys = ("Loss/Loss_reward_critic", "Loss/Loss_cost_critic", "Loss/Loss_actor", "Misc/Reward_critic_norm",
                "Misc/Cost_critic_norm", "Misc/Entropy", "Misc/Ratio")

    def __init__(self, vec_env, vec_eval_env, config, model_dir="", comm: Comm | None = None):
        """Data parallel: one process per GPU, each with its own shard of rollout threads (config["n_rollout_threads"] is
        the PER-RANK count here); replicas start identical (broadcast) and stay identical (all-reduced gradients,
        statistics and agent order); rank 0 logs and saves."""
        self.envs, self.eval_envs, self.config, self.model_dir = vec_env, vec_eval_env, config, model_dir
        self.comm = comm or Comm()
        self.is_root = self.comm.rank == 0
        self.num_agents = self.envs.num_agents
        self.dev = torch.device(config["device"])
        log_dir = config["log_dir"] if self.is_root else os.path.join(config["log_dir"], f"rank{self.comm.rank}")
        self.logger = EpochLogger(log_dir=log_dir, seed=str(config["seed"]), verbose=self.is_root)
        self.save_dir = str(config["log_dir"] + "/models_seed{}".format(config["seed"]))
        os.makedirs(self.save_dir, exist_ok=True)
        self.logger.save_config(config)
        self.use_cost = self.policy_cls.use_cost
        self.policy = [self.policy_cls(config, self.envs.observation_space[a], self.envs.share_observation_space[a],
                                       self.envs.action_space[a]) for a in range(self.num_agents)]
        if self.model_dir != "":
            self.restore()
        for pol in self.policy:
            for net in pol.networks():
                self.comm.broadcast_(net.theta, 0)
        self.trainer = [self.trainer_cls(config, self.policy[a], self.comm) for a in range(self.num_agents)]
        self.buffer = [SeparatedReplayBuffer(config, self.envs.observation_space[a], self.envs.share_observation_space[a],
                                             self.envs.action_space[a]) for a in range(self.num_agents)]
        self._stack_buffers()

    _STACKED = ("share_obs", "obs", "actions", "action_log_probs", "value_preds", "cost_preds", "rewards", "costs", "masks",
                "active_masks")

    def _stack_buffers(self):
        """Homogeneous agents: the per-agent buffers become views of ONE [agents, T(+1), N, ...] tensor per field, so a step
        is inserted with one strided copy per field instead of one per field per agent (the per-step work of collect /
        insert is pure launch overhead: ~85 tiny copies)."""
        b0 = self.buffer[0]
        same = all(getattr(b, f).shape == getattr(b0, f).shape for b in self.buffer for f in self._STACKED)
        self._stack = None
        if not same:
            return
        self._stack = {}
        for f in self._STACKED:
            st = torch.stack([getattr(b, f) for b in self.buffer])
            self._stack[f] = st
            for a, b in enumerate(self.buffer):
                setattr(b, f, st[a])

    def run(self):
        c = self.config
        self.warmup()
        start = time.time()
        episodes = int(c["num_env_steps"]) // c["episode_length"] // (c["n_rollout_threads"] * self.comm.world_size)
        train_episode_rewards = torch.zeros(1, c["n_rollout_threads"], device=self.dev)
        train_episode_costs = torch.zeros(1, c["n_rollout_threads"], device=self.dev)
        eval_rewards, eval_costs = 0.0, 0.0
        for episode in range(episodes):
            done_rewards, done_costs = [], []
            for step in range(c["episode_length"]):
                out = self.collect(step)
                values, actions, action_log_probs, rnn_states, rnn_states_critic = out[:5]
                cost_preds, rnn_states_cost = (out[5], out[6]) if self.use_cost else (None, None)
                obs, share_obs, rewards, costs, dones, infos, _ = self.envs.step(actions)
                dones_env = torch.all(dones, dim=1)
                train_episode_rewards += torch.mean(rewards, dim=1).flatten()
                train_episode_costs += torch.mean(costs, dim=1).flatten()
                if bool(dones_env.any()):                     # one host sync per step instead of one per rollout thread
                    idx = torch.nonzero(dones_env).flatten().tolist()
                    for t in idx:
                        done_rewards.append(train_episode_rewards[:, t].clone())
                        done_costs.append(train_episode_costs[:, t].clone())
                    train_episode_rewards[:, idx] = 0
                    train_episode_costs[:, idx] = 0
                done_episodes_costs_aver = train_episode_costs.mean()
                self.insert((obs, share_obs, rewards, costs, dones, infos, values, actions, action_log_probs, rnn_states,
                             rnn_states_critic, cost_preds, rnn_states_cost, done_episodes_costs_aver))
            self.compute()
            self.train()
            total_num_steps = (episode + 1) * c["episode_length"] * c["n_rollout_threads"] * self.comm.world_size
            if self.is_root and (episode % c["save_interval"] == 0 or episode == episodes - 1):
                self.save()
            end = time.time()
            if episode % c["eval_interval"] == 0 and c["use_eval"]:
                eval_rewards, eval_costs = self.eval()
            n_done = torch.tensor([float(len(done_rewards)), float(torch.stack(done_rewards).sum()) if done_rewards else 0.0,
                                   float(torch.stack(done_costs).sum()) if done_costs else 0.0], dtype=torch.float64, device=self.dev)
            self.comm.all_reduce_sum_(n_done)               # finished episodes of every shard
            if n_done[0].item() != 0:
                aver_episode_rewards = (n_done[1] / n_done[0]).float()
                aver_episode_costs = (n_done[2] / n_done[0]).float()
                self.return_aver_cost(aver_episode_costs)
                self.logger.store(**{"Metrics/EpRet": aver_episode_rewards.item(), "Metrics/EpCost": aver_episode_costs.item(),
                                     "Eval/EpRet": eval_rewards, "Eval/EpCost": eval_costs})
                self.logger.log_tabular("Metrics/EpRet", min_and_max=True, std=True)
                self.logger.log_tabular("Metrics/EpCost", min_and_max=True, std=True)
                self.logger.log_tabular("Eval/EpRet")
                self.logger.log_tabular("Eval/EpCost")
                self.logger.log_tabular("Train/Epoch", episode)
                self.logger.log_tabular("Train/TotalSteps", total_num_steps)
                keys = self.log_keys
                for k in keys:
                    if self.use_cost or "ost_critic" not in k:
                        self.logger.log_tabular(k)
                self.logger.log_tabular("Time/Total", end - start)
                self.logger.log_tabular("Time/FPS", int(total_num_steps / (end - start)))
                self.logger.dump_tabular()

    def return_aver_cost(self, aver_episode_costs):
        for b in self.buffer:
            b.return_aver_insert(aver_episode_costs)

    def warmup(self):
        obs, share_obs, _ = self.envs.reset()
        for a in range(self.num_agents):
            self.buffer[a].share_obs[0].copy_(share_obs[:, a])
            self.buffer[a].obs[0].copy_(obs[:, a])

    @torch.no_grad()
    def _collect_fused(self, share_obs, obs, rnn, rnn_c, rnn_k, masks, slots=None):
        """policy.get_actions of every agent as ONE launch (spo_ma_collect_forward: all networks of all agents, each row tile
        through its whole network on chip, Gaussian sampling included).  Same results bit for bit as the per-network calls
        below, same torch.randn draws in the same order.  Returns None when the networks are outside the fused kernel's
        geometry (hidden != 128, wide observations, ...): the caller then launches network by network.
        slots: {"value_preds", "cost_preds": [agents, N, 1], "actions", "action_log_probs": [agents, N, A]} -- the buffer rows
        of this step; the kernel then writes its results where insert() would copy them."""
        lib = _abi.load()
        jobs = []                                            # (net, input, agent, kind)
        for a in range(self.num_agents):
            pol = self.trainer[a].policy
            jobs.append((pol.actor, obs[a], a, "actor"))
            jobs.append((pol.critic, share_obs[a], a, "critic"))
            if self.use_cost:
                jobs.append((pol.cost_critic, share_obs[a], a, "cost"))
        if len(jobs) > _abi.MA_COLLECT_MAX_NETS:
            return None
        arr = (_abi.MaCollectNet * len(jobs))()
        keep, acts, lps = [], [None] * self.num_agents, [None] * self.num_agents
        N = obs[0].reshape(-1, self.trainer[0].policy.actor._net.in_dim).shape[0]
        if slots is not None:
            vals, cps = slots["value_preds"], slots.get("cost_preds")
            ok = all(t is None or (t.dtype == torch.float32 and t[a].is_contiguous()) for t in (vals, cps, slots["actions"],
                     slots["action_log_probs"]) for a in range(self.num_agents))
            if not ok or vals.shape[1:] != (N, 1):
                return None
        else:
            vals = torch.empty((self.num_agents, N, 1), dtype=torch.float32, device=self.dev)
            cps = torch.empty((self.num_agents, N, 1), dtype=torch.float32, device=self.dev) if self.use_cost else None
        for i, (net, x, a, kind) in enumerate(jobs):
            x = _abi.require_gpu_tensor(torch.as_tensor(x, **net.tpdv).reshape(-1, net._net.in_dim).contiguous(), "x", torch.float32)
            if x.shape[0] != N:
                return None
            keep.append(x)
            c = arr[i]
            c.theta, c.net, c.x, c.deterministic = _abi.ptr(net.theta), net._net, _abi.ptr(x), 0
            if kind == "actor":
                eps = torch.randn((N, net.act_dim), **net.tpdv)
                if slots is not None:
                    acts[a], lps[a] = slots["actions"][a], slots["action_log_probs"][a]
                    if acts[a].shape != eps.shape or lps[a].shape != eps.shape:
                        return None
                else:
                    acts[a], lps[a] = torch.empty_like(eps), torch.empty_like(eps)
                keep.append(eps)
                c.eps, c.act, c.logp, c.out = _abi.ptr(eps), _abi.ptr(acts[a]), _abi.ptr(lps[a]), None
                c.std_x_coef, c.std_y_coef = net.std_x_coef, net.std_y_coef
            else:
                c.out = _abi.ptr((vals if kind == "critic" else cps)[a])
        scratch = getattr(self, "_mc_scratch", None)
        need = int(lib.spo_ma_collect_scratch_floats(len(jobs)))
        if scratch is None or scratch.numel() < need:
            scratch = self._mc_scratch = torch.empty(need, dtype=torch.float32, device=self.dev)
        rc = lib.spo_ma_collect_forward(len(jobs), arr, N, _abi.ptr(scratch), _abi.stream_ptr())
        if rc == _abi.MA_COLLECT_UNSUPPORTED:
            return None
        _abi.check(rc, "spo_ma_collect_forward")
        tr = lambda xs: torch.transpose(torch.stack(xs), 1, 0)
        if slots is not None:
            # recurrent policies are not built: the states handed back are the all-zero inputs, stacked once and reused
            zr = getattr(self, "_rnn_zero_out", None)
            if zr is None or zr[0].shape[0] != N:
                zr = self._rnn_zero_out = tuple(tr([torch.zeros_like(t) for t in grp]) for grp in (rnn, rnn_c, rnn_k))
            r_out, rc_out, rk_out = zr
        else:
            r_out, rc_out, rk_out = tr(list(rnn)), tr(list(rnn_c)), (tr(list(rnn_k)) if self.use_cost else None)
        if not self.use_cost:
            return vals.transpose(1, 0), acts, lps, r_out, rc_out
        return vals.transpose(1, 0), acts, lps, r_out, rc_out, cps.transpose(1, 0), rk_out

    @torch.no_grad()
    def _collect_eager(self, share_obs, obs, rnn, rnn_c, rnn_k, masks):
        if self.config.get("collect_fused", True) and not getattr(self, "_fused_unsupported", False):
            out = self._collect_fused(share_obs, obs, rnn, rnn_c, rnn_k, masks)
            if out is not None:
                return out
            self._fused_unsupported = True                   # geometry: decided once, the networks do not change shape
        vals, acts, lps, o_rnn, o_rnn_c, cps, o_rnn_k = [], [], [], [], [], [], []
        tr = lambda xs: torch.transpose(torch.stack(xs), 1, 0)
        if not self.use_cost:
            for a in range(self.num_agents):
                v, act, lp, r, rc = self.trainer[a].policy.get_actions(share_obs[a], obs[a], rnn[a], rnn_c[a], masks[a])
                vals.append(v); acts.append(act); lps.append(lp); o_rnn.append(r); o_rnn_c.append(rc)
            return tr(vals), acts, lps, tr(o_rnn), tr(o_rnn_c)
        for a in range(self.num_agents):
            v, act, lp, r, rc, cp, rk = self.trainer[a].policy.get_actions(share_obs[a], obs[a], rnn[a], rnn_c[a], masks[a],
                                                                         rnn_states_cost=rnn_k[a])
            vals.append(v); acts.append(act); lps.append(lp); o_rnn.append(r); o_rnn_c.append(rc); cps.append(cp); o_rnn_k.append(rk)
        return tr(vals), acts, lps, tr(o_rnn), tr(o_rnn_c), tr(cps), tr(o_rnn_k)

    @torch.no_grad()
    def collect(self, step):
        """mappolag.py:411-447.  The 12 network forwards + sampling of one step are ~110 small launches (8 k per epoch,
        launch-bound); with config["collect_graph"] (default on) they are captured once into a HIP graph over static
        input buffers and replayed every step.  Any capture failure falls back to eager launches for good."""
        bufs = self.buffer
        ins = ([b.share_obs[step] for b in bufs], [b.obs[step] for b in bufs], [b.rnn_states[step] for b in bufs],
               [b.rnn_states_critic[step] for b in bufs], [b.rnn_states_cost[step] for b in bufs], [b.masks[step] for b in bufs])
        stacked = getattr(self, "_stack", None) is not None
        if not self.config.get("collect_graph", True) or getattr(self, "_graph_failed", False):
            return self._collect_eager(*ins)
        if (stacked and self.config.get("collect_inplace", True) and self.config.get("collect_fused", True)
                and getattr(self, "_inplace_ok", True)):
            out = self._collect_step_graph(step, ins)
            if out is not None:
                return out
        if getattr(self, "_graph", None) is None:
            try:
                if stacked:      # one static [agents, N, ...] tensor per input; the per-agent graph inputs are its views
                    self._static_stk = {f: self._stack[f][:, step].clone() for f in ("share_obs", "obs", "masks")}
                    self._static_in = ([self._static_stk["share_obs"][a] for a in range(self.num_agents)],
                                       [self._static_stk["obs"][a] for a in range(self.num_agents)],
                                       [t.clone() for t in ins[2]], [t.clone() for t in ins[3]], [t.clone() for t in ins[4]],
                                       [self._static_stk["masks"][a] for a in range(self.num_agents)])
                else:
                    self._static_in = tuple([t.clone() for t in group] for group in ins)
                self._collect_eager(*self._static_in)            # warm-up: workspaces, allocator pools
                torch.cuda.synchronize(self.dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._static_out = self._collect_eager(*self._static_in)
                self._graph = g
            except Exception as e:                                   # noqa: BLE001 -- eager launches are always valid
                self._graph_failed = True
                self._graph = None
                torch.cuda.synchronize(self.dev)
                if self.is_root:
                    print(f"[safepo] collect graph capture failed ({type(e).__name__}: {e}); using eager launches", file=sys.stderr)
                return self._collect_eager(*ins)
        if stacked:
            for f in ("share_obs", "obs", "masks"):
                self._static_stk[f].copy_(self._stack[f][:, step])
        else:
            for gi, (dst_group, src_group) in enumerate(zip(self._static_in, ins)):
                if gi in (2, 3, 4):
                    continue                                      # rnn states: zeros on both sides, nothing to copy
                for dst, src in zip(dst_group, src_group):
                    dst.copy_(src)
        self._graph.replay()
        if not self.use_cost:
            v, acts, lps, r, rc = self._static_out
            return v.clone(), [x.clone() for x in acts], [x.clone() for x in lps], r, rc
        v, acts, lps, r, rc, cp, rk = self._static_out
        return v.clone(), [x.clone() for x in acts], [x.clone() for x in lps], r, rc, cp.clone(), rk

    def _collect_step_graph(self, step, ins):
        """One captured graph PER STEP INDEX over the one-launch collect, reading the step's rows of the stacked buffers and
        writing values / actions / log-probabilities / cost predictions straight into the rows insert() fills: a collect step
        is one graph replay -- no staging copies in, no clones out -- and insert() skips the copies whose source already is
        the destination.  The tensors handed back are views of the buffer (valid like the reference's until the step is
        overwritten one episode later).  None = not available (geometry outside the fused kernel, capture failure)."""
        graphs = self.__dict__.setdefault("_step_graphs", {})
        # the graphs hold raw pointers: parameter vectors or buffers that were re-created since the capture invalidate them
        key = tuple(n.theta.data_ptr() for t in self.trainer for n in t.policy.networks()) + (self._stack["obs"].data_ptr(),)
        if getattr(self, "_step_graph_key", key) != key:
            graphs.clear()
        self._step_graph_key = key
        ent = graphs.get(step)
        if ent is None:
            st = self._stack
            slots = {"value_preds": st["value_preds"][:, step], "actions": st["actions"][:, step],
                     "action_log_probs": st["action_log_probs"][:, step]}
            if self.use_cost:
                slots["cost_preds"] = st["cost_preds"][:, step]
            try:
                if not hasattr(self, "_inplace_ok"):
                    if self._collect_fused(*ins, slots=slots) is None:   # warm-up + geometry check, once
                        self._inplace_ok = False
                        return None
                    torch.cuda.synchronize(self.dev)
                    self._inplace_ok = True
                    self._step_pool = torch.cuda.graph_pool_handle()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self._step_pool, capture_error_mode="thread_local"):
                    out = self._collect_fused(*ins, slots=slots)
                ent = graphs[step] = (g, out)
            except Exception as e:                                   # noqa: BLE001 -- the single-graph / eager paths stay valid
                self._inplace_ok = False
                torch.cuda.synchronize(self.dev)
                if self.is_root:
                    print(f"[safepo] per-step collect graph capture failed ({type(e).__name__}: {e}); using the staged graph",
                          file=sys.stderr)
                return None
        ent[0].replay()
        return ent[1]

    def insert(self, data, aver_episode_costs=0):
        (obs, share_obs, rewards, costs, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic,
         cost_preds, rnn_states_cost, done_episodes_costs_aver) = data
        def torch_masks():
            # mappolag.py:458-472, as mask arithmetic on the device instead of boolean-index assignment
            # (boolean-index assignment would run nonzero(): a host synchronisation in every step)
            dones_env = torch.all(dones, axis=1)
            keep = (~dones_env).float()
            return (keep.view(-1, 1, 1).expand(-1, self.num_agents, 1),
                    (dones_env.view(-1, 1) | ~dones).float().unsqueeze(-1))             # 0 for a done agent of a live env
        # (rnn states are zeros throughout: recurrent policies are not built, so there is nothing to reset at episode ends)
        if self._stack is not None:
            st, s0 = self._stack, self.buffer[0].step
            tr = lambda t: t.transpose(0, 1)                                  # [N, agents, ...] -> [agents, N, ...]
            dense = lambda t, dt: torch.is_tensor(t) and t.is_cuda and t.dtype == dt and t.is_contiguous()
            fused = (self.config.get("insert_fused", True) and dense(obs, torch.float32) and dense(share_obs, torch.float32)
                     and dense(rewards, torch.float32) and (not self.use_cost or dense(costs, torch.float32))
                     and dense(dones, torch.bool) and tuple(dones.shape) == (obs.shape[0], self.num_agents)
                     and rewards.numel() == dones.numel())
            if fused:
                # observations, shared observations, rewards, costs and both masks of the step in ONE launch
                dst = lambda f, slot: (_abi.ptr(st[f][0, slot]), st[f].stride(0))
                args = (dst("obs", s0 + 1) + dst("share_obs", s0 + 1) + dst("rewards", s0)
                        + (dst("costs", s0) if self.use_cost else (None, 0)) + dst("masks", s0 + 1) + dst("active_masks", s0 + 1))
                _abi.check(_abi.load().spo_ma_insert_step(
                    _abi.ptr(obs), _abi.ptr(share_obs), _abi.ptr(rewards), _abi.ptr(costs) if self.use_cost else None, _abi.ptr(dones),
                    *args, obs.shape[0], self.num_agents, obs.shape[-1], share_obs.shape[-1], _abi.stream_ptr()), "spo_ma_insert_step")
            else:
                st["share_obs"][:, s0 + 1].copy_(tr(share_obs)); st["obs"][:, s0 + 1].copy_(tr(obs))
            def put(dst, src):                                                # src() only when the row is not already in place
                s = src()
                if not (s.data_ptr() == dst.data_ptr() and s.shape == dst.shape and s.stride() == dst.stride()):
                    dst.copy_(s)
            in_place = lambda xs, f: all(x.data_ptr() == st[f][a, s0].data_ptr() and x.shape == st[f][a, s0].shape
                                         and x.is_contiguous() for a, x in enumerate(xs))
            if not in_place(actions, "actions"):
                st["actions"][:, s0].copy_(torch.stack(actions))
            if not in_place(action_log_probs, "action_log_probs"):
                st["action_log_probs"][:, s0].copy_(torch.stack(action_log_probs))
            put(st["value_preds"][:, s0], lambda: tr(values))
            if self.use_cost:       # happo.py:403-414 / mappo.py:395-406 keep costs out of the buffer
                put(st["cost_preds"][:, s0], lambda: tr(cost_preds))
            if not fused:
                st["rewards"][:, s0].copy_(tr(rewards))
                if self.use_cost:
                    st["costs"][:, s0].copy_(tr(costs))
                masks, active_masks = torch_masks()
                st["masks"][:, s0 + 1].copy_(tr(masks)); st["active_masks"][:, s0 + 1].copy_(tr(active_masks))
            for b in self.buffer:
                b.step = (s0 + 1) % b.episode_length
            return
        masks, active_masks = torch_masks()
        for a in range(self.num_agents):
            if not self.use_cost:
                self.buffer[a].insert(share_obs[:, a], obs[:, a], rnn_states[:, a], rnn_states_critic[:, a], actions[a],
                                      action_log_probs[a], values[:, a], rewards[:, a], masks[:, a], None, active_masks[:, a], None)
                continue
            self.buffer[a].insert(share_obs[:, a], obs[:, a], rnn_states[:, a], rnn_states_critic[:, a], actions[a],
                                  action_log_probs[a], values[:, a], rewards[:, a], masks[:, a], None, active_masks[:, a], None,
                                  costs=costs[:, a], cost_preds=cost_preds[:, a], rnn_states_cost=rnn_states_cost[:, a],
                                  done_episodes_costs_aver=done_episodes_costs_aver, aver_episode_costs=aver_episode_costs)

    def train(self, order=None, perm_fn=None):
        c = self.config
        factor = torch.ones(c["episode_length"], c["n_rollout_threads"], 1, device=self.dev)
        if order is None:
            order = torch.randperm(self.num_agents).to(self.dev)
            self.comm.broadcast_(order, 0)                   # the same HAPPO update order on every rank
            order = order.tolist()
        for agent_id in order:
            a = int(agent_id)
            b = self.buffer[a]
            action_dim = b.actions.shape[-1]
            b.update_factor(factor)
            flat = lambda t: t.reshape(-1, *t.shape[2:])
            args = (flat(b.obs[:-1]), flat(b.rnn_states[0:1]), flat(b.actions), flat(b.masks[:-1]), None, flat(b.active_masks[:-1]))
            old_logp = self.trainer[a].policy.actor.evaluate_actions(*args)[0]
            self.trainer[a].train(b, logger=self.logger, perm_fn=(lambda it, a=a: perm_fn(a, it)) if perm_fn else None)
            new_logp = self.trainer[a].policy.actor.evaluate_actions(*args)[0]
            action_prod = torch.prod(torch.exp(new_logp - old_logp).reshape(c["episode_length"], c["n_rollout_threads"], action_dim),
                                     dim=-1, keepdim=True)
            factor = factor * action_prod
            b.after_update()

    def save(self):
        for a in range(self.num_agents):
            torch.save(self.trainer[a].policy.actor.state_dict(), str(self.save_dir) + "/actor_agent" + str(a) + ".pt")
            torch.save(self.trainer[a].policy.critic.state_dict(), str(self.save_dir) + "/critic_agent" + str(a) + ".pt")

    def restore(self):
        for a in range(self.num_agents):
            self.policy[a].actor.load_state_dict(torch.load(str(self.model_dir) + "/actor_agent" + str(a) + ".pt"))
            self.policy[a].critic.load_state_dict(torch.load(str(self.model_dir) + "/critic_agent" + str(a) + ".pt"))

    @torch.no_grad()
    def eval(self, eval_episodes=1):
        c = self.config
        n = c["n_eval_rollout_threads"]
        done, rets, csts = 0, [], []
        one_r, one_c = torch.zeros(1, n, device=self.dev), torch.zeros(1, n, device=self.dev)
        eval_obs, _, _ = self.eval_envs.reset()
        rnn = torch.zeros(n, self.num_agents, c["recurrent_N"], 1, device=self.dev)
        masks = torch.ones(n, self.num_agents, 1, device=self.dev)
        while True:
            acts = []
            for a in range(self.num_agents):
                act, r = self.trainer[a].policy.act(eval_obs[:, a], rnn[:, a], masks[:, a], deterministic=True)
                rnn[:, a] = r
                acts.append(act)
            eval_obs, _, rewards, costs, dones, _, _ = self.eval_envs.step(acts)
            one_r += torch.mean(rewards, dim=1).flatten()
            one_c += torch.mean(costs, dim=1).flatten()
            dones_env = torch.all(dones, dim=1)
            masks = (~dones_env).float().view(-1, 1, 1).expand(-1, self.num_agents, 1).contiguous()
            rnn = rnn * masks.unsqueeze(-1)
            for i in torch.nonzero(dones_env).flatten().tolist():
                done += 1
                rets.append(one_r[:, i].mean().item()); one_r[:, i] = 0
                csts.append(one_c[:, i].mean().item()); one_c[:, i] = 0
            if done >= eval_episodes:
                return np.mean(rets), np.mean(csts)

    @torch.no_grad()
    def compute(self):
        """mappolag.py:587-603: both recurrences of an agent in ONE kernel (spo_ma_gae) with PopArt de-normalisation."""
        for a in range(self.num_agents):
            b, tr = self.buffer[a], self.trainer[a]
            tr._sync_normalizer()
            next_value = tr.policy.get_values(b.share_obs[-1], b.rnn_states_critic[-1], b.masks[-1])
            if not self.use_cost:
                b.compute_returns(next_value, tr.value_normalizer)
                continue
            next_cost = tr.policy.get_cost_values(b.share_obs[-1], b.rnn_states_cost[-1], b.masks[-1])
            b.compute_returns_and_cost_returns(next_value, next_cost, tr.value_normalizer, tr.value_normalizer)


def train(args, cfg_train, runner_cls=None):
    from safepo.common.env import make_ma_synth_env
    runner_cls = runner_cls or Runner
    if not str(args.task).startswith("Synth"):
        raise NotImplementedError("this build has no simulator (safety_gymnasium / Isaac Gym are not installed here); "
                                  "use a Synth* multi-agent task, or pass your own vector env to Runner(...)")
    comm = init_from_env()
    if comm.world_size > 1:
        # one process per GPU (torchrun): shard the rollout threads, one device per rank
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        cfg_train = dict(cfg_train)
        cfg_train["device"] = f"cuda:{local_rank}"
        torch.cuda.set_device(local_rank)
        assert cfg_train["n_rollout_threads"] % comm.world_size == 0, "n_rollout_threads must divide over the ranks"
        cfg_train["n_rollout_threads"] //= comm.world_size
    env = make_ma_synth_env(cfg_train, seed=args.seed + 1000 * comm.rank)
    cfg_eval = copy.deepcopy(cfg_train)
    cfg_eval["seed"] = args.seed + 10000
    cfg_eval["n_rollout_threads"] = cfg_eval["n_eval_rollout_threads"]
    eval_env = make_ma_synth_env(cfg_eval, seed=args.seed + 10000)
    runner = runner_cls(env, eval_env, cfg_train, args.model_dir, comm=comm)
    if args.model_dir != "":
        runner.eval(100000)
    else:
        runner.run()
    return runner


def cli(algo, train_fn):
    """The `if __name__ == "__main__"` block shared by the multi-agent scripts (mappolag.py:606-637)."""
    from safepo.utils.config import multi_agent_args
    args, cfg_env, cfg_train = multi_agent_args(algo=algo)
    torch.manual_seed(cfg_train.get("seed", 0))
    np.random.seed(cfg_train.get("seed", 0))
    if args.write_terminal:
        train_fn(args=args, cfg_train=cfg_train)
    else:
        os.makedirs(cfg_train["log_dir"], exist_ok=True)
        with open(os.path.join(cfg_train["log_dir"], f"seed{args.seed}_terminal.log"), "w", encoding="utf-8") as f_out, \
                open(os.path.join(cfg_train["log_dir"], f"seed{args.seed}_error.log"), "w", encoding="utf-8") as f_err:
            sys.stdout, sys.stderr = f_out, f_err
            train_fn(args=args, cfg_train=cfg_train)


if __name__ == "__main__":
    cli("mappolag", train)
