"""MACPO (multi-agent constrained policy optimisation): reference safepo/multi_agent/macpo.py.

Same networks, buffers, PopArt critics and Runner as MAPPO-L (mappolag.py); the actor takes a trust-region step per
agent (macpo.py:201-371) instead of clipped-surrogate Adam steps:

  * gradients of the reward / cost surrogates mean(factor * prod_a ratio_a * adv) through the loss kernel
    (spo_ma_actor_loss with the clip switched off) and spo_ma_backward;
  * two conjugate-gradient solves against the Fisher matrix.  The reference differentiates its KL expression twice
    (macpo.py:187-199).  At theta = theta_old the mean enters that expression only through (mu_old - mu)^2, so its
    Hessian is exactly J^T M J with M = 2 / (1e-8 + 2 sigma^2) per action dimension, plus a diagonal block for
    log_std (sigma is state-independent) -- no second-order autograd: one forward-mode pass (spo_ma_jvp: in-tree MFMA
    GEMMs + a LayerNorm/ELU tangent kernel) and one ordinary backward pass per product;
  * the case analysis for (lam, nu), the step and the backtracking line search on the host, as written.

Surface: MACPO_Policy, MACPO_Trainer, Runner, train(args, cfg_train).  Single GPU (the trust-region step is not
sharded: MACPO_Trainer refuses world_size > 1).
"""
from __future__ import annotations

import numpy as np
import torch

from safepo import _abi
from safepo.multi_agent import mappolag as _base
from safepo.multi_agent.mappolag import check

# marl_cfg/macpo/config.yaml and its `mamujoco` block
default_cfg = dict(_base.default_cfg, env_name="macpo", algorithm_name="macpo", EPS=1.0e-8, safety_gamma=0.09, step_fraction=0.5,
                   g_step_dir_coef=0.1, b_step_dir_coef=0.1, fraction_coef=0.1, conjugate_gradient_iters=10)
for _k in ("lagrangian_coef_rate", "lamda_lagr", "use_single_network"):
    default_cfg.pop(_k)
mamujoco_cfg = dict(layer_N=1, num_env_steps=10000000, episode_length=1000, n_rollout_threads=10, n_eval_rollout_threads=10,
                    hidden_size=128, gamma=0.99, safety_gamma=0.2, target_kl=0.01, learning_iters=15, entropy_coef=0.01)


class MACPO_Policy(_base.MAPPO_L_Policy):
    """macpo.py:45-96: actor, critic, cost critic; evaluate_actions also returns the action mean and stddev."""

    def evaluate_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, action, masks, available_actions=None,
                         active_masks=None, rnn_states_cost=None):
        action_log_probs, dist_entropy, action_mu, action_std = self.actor.evaluate_actions(
            obs, rnn_states_actor, action, masks, available_actions, active_masks)
        values, _ = self.critic(cent_obs, rnn_states_critic, masks)
        cost_values, _ = self.cost_critic(cent_obs, rnn_states_cost, masks)
        return values, action_log_probs, dist_entropy, cost_values, action_mu, action_std


class MACPO_Trainer(_base.MAPPO_L_Trainer):
    """macpo.py:98-423."""
    algo = "macpo"

    def __init__(self, config, policy, comm=None):
        super().__init__(config, policy, comm)
        if self.comm.world_size > 1:
            raise NotImplementedError("MACPO's trust-region step is not sharded over ranks; run it on one GPU")
        self.use_cost = True
        self._lamda.zero_()                       # no multiplier in the surrogate
        self._surr_cfg = _abi.MaLossCfg(clip_param=3.0e38, entropy_coef=0.0, std_x_coef=float(config["std_x_coef"]),
                                        std_y_coef=float(config["std_y_coef"]), use_policy_active_masks=0, per_dim_ratio=0)

    # ---- pieces of trpo_update
    def _std(self):
        a = self.policy.actor
        return torch.sigmoid(a.log_std / a.std_x_coef) * a.std_y_coef

    def _surrogate_grad(self, saved, mean, actions, old_lp, adv, factor, active):
        """-mean(factor * prod ratio * adv) and its flat gradient (macpo.py:238-249): loss kernel without clip, entropy or
        multiplier, then the network backward."""
        pol, lib = self.policy, _abi.load()
        A, rows = pol.actor.act_dim, mean.shape[0]
        grad = torch.zeros_like(pol.actor.theta)
        dmean = torch.empty_like(mean)
        ls = pol.actor.offset(6)
        _abi.check(lib.spo_ma_actor_loss(_abi.ptr(mean), _abi.ptr(pol.actor.log_std), _abi.ptr(actions), _abi.ptr(old_lp), _abi.ptr(adv),
                                         _abi.ptr(self._zeros), _abi.ptr(factor), _abi.ptr(active), _abi.ptr(self._lamda),
                                         self._surr_cfg, rows, A, float(rows), rows, _abi.ptr(dmean), _abi.ptr(grad[ls:ls + A]),
                                         _abi.ptr(self._scalars), _abi.ptr(self._partial), _abi.stream_ptr()), "spo_ma_actor_loss")
        loss = self._scalars[0].clone()
        pol.actor.net_backward(saved, dmean, grad)
        return loss, grad

    def _kl_hessian_logstd(self):
        """d^2/d log_std^2 of  -log s + s_old^2 / (1e-8 + 2 s^2)  at s = s_old, with s = y * sigmoid(log_std / x): the
        log_std block of the Hessian of macpo.py:160-166 (identical in every row, so the row mean changes nothing)."""
        a = self.policy.actor
        ell = a.log_std.double()
        sg = torch.sigmoid(ell / a.std_x_coef)
        s = sg * a.std_y_coef
        s1 = a.std_y_coef / a.std_x_coef * sg * (1 - sg)
        s2 = a.std_y_coef / a.std_x_coef ** 2 * sg * (1 - sg) * (1 - 2 * sg)
        c, den = s * s, 1e-8 + 2 * s * s
        f1 = -1 / s - 4 * c * s / den ** 2
        f2 = 1 / (s * s) - 4 * c / den ** 2 + 32 * c * s * s / den ** 3
        return (f2 * s1 * s1 + f1 * s2).float()

    def fisher_vector_product(self, saved, p, m_diag, h_ls):
        """macpo.py:187-199: Hessian of the mean KL times p, + 0.1 p."""
        a = self.policy.actor
        rows = saved[0].shape[0]
        dmu = a.net_jvp(saved, p)
        out = torch.zeros_like(p)
        a.net_backward(saved, dmu * (m_diag / rows), out)
        ls = a.offset(6)
        out[ls:ls + a.act_dim] = h_ls * p[ls:ls + a.act_dim]
        return out + 0.1 * p

    def conjugate_gradient(self, saved, b, nsteps, m_diag, h_ls, residual_tol=1e-10):
        """macpo.py:168-185."""
        x = torch.zeros_like(b)
        r, p = b.clone(), b.clone()
        rdotr = torch.dot(r, r)
        for _ in range(nsteps):
            avp = self.fisher_vector_product(saved, p, m_diag, h_ls)
            alpha = rdotr / (torch.dot(p, avp) + 1e-8)
            x += alpha * p
            r -= alpha * avp
            new_rdotr = torch.dot(r, r)
            p = r + (new_rdotr / rdotr) * p
            rdotr = new_rdotr
            if rdotr < residual_tol:
                break
        return x

    def kl_divergence(self, mu, std, mu_old, std_old):
        """macpo.py:153-166 (as written: log(std_old) - log(std) + ...)."""
        kl = torch.log(std_old) - torch.log(std) + (std_old.pow(2) + (mu_old - mu).pow(2)) / (1e-8 + 2.0 * std.pow(2)) - 0.5
        return kl.sum(1, keepdim=True)

    def trpo_update(self, sample):
        (share_obs_batch, obs_batch, _rnn, _rnn_c, actions_batch, value_preds_batch, return_batch, _masks, active_masks_batch,
         old_action_log_probs_batch, adv_targ, _avail, factor_batch, cost_preds_batch, cost_returns_batch, _rnn_k, cost_adv_targ,
         aver_episode_costs) = sample
        c, pol = self.config, self.policy
        f = lambda t: _abi.require_gpu_tensor(check(t).to(**self.tpdv).contiguous(), "sample", torch.float32)
        obs_batch, share_obs_batch, actions_batch = f(obs_batch), f(share_obs_batch), f(actions_batch)
        old_lp, adv, cadv = f(old_action_log_probs_batch), f(adv_targ).reshape(-1), f(cost_adv_targ).reshape(-1)
        factor, active = f(factor_batch).reshape(-1), f(active_masks_batch).reshape(-1)
        if self._zeros is None or self._zeros.numel() != adv.numel():
            self._zeros, self._ones = torch.zeros_like(adv), torch.ones_like(adv)
        actor = pol.actor
        # ---- critics first (macpo.py:218-230); both share the one PopArt normaliser
        value_loss, critic_grad_norm = self._value_step(pol.critic, pol.critic_optimizer, share_obs_batch, f(value_preds_batch),
                                                        f(return_batch))
        _cost_value_loss, cost_grad_norm = self._value_step(pol.cost_critic, pol.cost_optimizer, share_obs_batch, f(cost_preds_batch),
                                                       f(cost_returns_batch))
        self._sync_normalizer()
        rescale_constraint_val = (float(check(aver_episode_costs).float().mean().item()) - c["cost_limit"]) * (1 - c["gamma"])
        if rescale_constraint_val == 0:
            rescale_constraint_val = 1e-8
        # ---- surrogate gradients at the old parameters
        mean, saved = actor.net_forward(obs_batch, keep=True)
        std_vec = self._std()

        def entropy_of(std):                          # Normal.entropy of the state-independent sigma, as MultiAgentActor.evaluate_actions
            ent = 0.5 + 0.5 * np.log(2 * np.pi) + torch.log(std)
            return ent.sum() if c["use_policy_active_masks"] else ent.mean()
        dist_entropy = entropy_of(std_vec)
        reward_loss, reward_loss_grad = self._surrogate_grad(saved, mean, actions_batch, old_lp, adv, factor, active)
        neg_cost_loss, neg_cost_grad = self._surrogate_grad(saved, mean, actions_batch, old_lp, cadv, factor, active)
        cost_loss, cost_loss_grad = -neg_cost_loss, -neg_cost_grad
        B_cost_loss_grad = cost_loss_grad
        m_diag = (2.0 / (1e-8 + 2.0 * std_vec * std_vec)).reshape(1, -1)
        h_ls = self._kl_hessian_logstd()
        iters = int(c["conjugate_gradient_iters"])
        g_step_dir = self.conjugate_gradient(saved, reward_loss_grad, iters, m_diag, h_ls)
        b_step_dir = self.conjugate_gradient(saved, B_cost_loss_grad, iters, m_diag, h_ls)
        q_coef = float(torch.dot(reward_loss_grad, g_step_dir))
        fraction = c["step_fraction"]
        B_cost_loss_grad_dot = float(torch.dot(B_cost_loss_grad, B_cost_loss_grad))
        tkl = float(c["target_kl"])
        # ---- case analysis (macpo.py:271-327), host scalars
        if B_cost_loss_grad_dot <= 1e-8 and rescale_constraint_val < 0:
            b_step_dir = torch.zeros_like(g_step_dir)
            r_coef = s_coef = positive_Cauchy_value = whether_recover_policy_value = 0.0
            optim_case = 4
        else:
            r_coef = float(torch.dot(reward_loss_grad, b_step_dir))
            s_coef = float(torch.dot(cost_loss_grad, b_step_dir))
            if r_coef == 0:
                r_coef = 1e-8
            if s_coef == 0:
                s_coef = 1e-8
            positive_Cauchy_value = q_coef - (r_coef ** 2) / (1e-8 + s_coef)
            whether_recover_policy_value = 2 * tkl - (rescale_constraint_val ** 2) / (1e-8 + s_coef)
            if rescale_constraint_val < 0 and whether_recover_policy_value < 0:
                optim_case = 3
            elif rescale_constraint_val < 0 and whether_recover_policy_value >= 0:
                optim_case = 2
            elif rescale_constraint_val >= 0 and whether_recover_policy_value >= 0:
                optim_case = 1
            else:
                optim_case = 0
        if whether_recover_policy_value == 0:
            whether_recover_policy_value = 1e-8
        sqrt = lambda v: float(torch.sqrt(torch.tensor(float(v))))           # torch.sqrt: NaN for a negative argument
        if optim_case in [3, 4]:
            lam, nu = sqrt(q_coef / (2 * tkl)), 0.0
        elif optim_case in [1, 2]:
            LA, LB = [0, r_coef / rescale_constraint_val], [r_coef / rescale_constraint_val, np.inf]
            LA, LB = (LA, LB) if rescale_constraint_val < 0 else (LB, LA)
            proj = lambda x, L: max(L[0], min(L[1], x))
            lam_a = proj(sqrt(positive_Cauchy_value / whether_recover_policy_value), LA)
            lam_b = proj(sqrt(q_coef / (2 * tkl)), LB)
            f_a = lambda lam: -0.5 * (positive_Cauchy_value / (1e-8 + lam) + whether_recover_policy_value * lam) \
                - r_coef * rescale_constraint_val / (1e-8 + s_coef)
            f_b = lambda lam: -0.5 * (q_coef / (1e-8 + lam) + 2 * tkl * lam)
            lam = lam_a if f_a(lam_a) >= f_b(lam_b) else lam_b
            nu = max(0, lam * rescale_constraint_val - r_coef) / (1e-8 + s_coef)
        else:
            lam, nu = 0.0, sqrt(2 * tkl / (1e-8 + s_coef))
        x_a = (1.0 / (lam + 1e-8)) * (g_step_dir + nu * b_step_dir)
        x_b = nu * b_step_dir
        x = x_a if optim_case > 0 else x_b
        # ---- backtracking line search (macpo.py:329-366)
        params = actor.theta.clone()
        mu_old, std_old = mean, std_vec.clone().reshape(1, -1)
        expected_improve = -torch.dot(x, reward_loss_grad)
        flag = False
        fraction_coef = c["fraction_coef"]
        kl = torch.zeros((), **self.tpdv)
        loss_improve = torch.zeros((), **self.tpdv)
        ratio = None
        lib = _abi.load()
        for i in range(int(c["searching_steps"])):
            x_norm = torch.norm(x)
            if float(x_norm) > 0.5:
                x = x * 0.5 / x_norm
            actor.theta.copy_(params - fraction_coef * (fraction ** i) * x)
            mu = actor.net_forward(obs_batch)
            logp = torch.empty_like(mu)
            _abi.check(lib.spo_ma_log_probs(_abi.ptr(mu), _abi.ptr(actor.log_std), _abi.ptr(actions_batch), actor.std_x_coef,
                                            actor.std_y_coef, _abi.ptr(logp), mu.shape[0], actor.act_dim, _abi.stream_ptr()),
                       "spo_ma_log_probs")
            ratio = torch.prod(torch.exp(logp - old_lp), dim=-1, keepdim=True)
            w = ratio.reshape(-1) * factor
            new_reward_loss = -(w * adv).mean()
            new_cost_loss = (w * cadv).mean()
            loss_improve = new_reward_loss - reward_loss
            std_new = self._std()
            dist_entropy = entropy_of(std_new)         # the reference returns the entropy of the last parameters it tried
            kl = self.kl_divergence(mu, std_new.reshape(1, -1), mu_old, std_old).mean()
            if (float(kl) < tkl and (float(loss_improve) < 0 if optim_case > 1 else True)
                    and float(new_cost_loss - cost_loss) <= max(-rescale_constraint_val, 0)):
                flag = True
                break
            expected_improve = expected_improve * fraction
        if not flag:
            actor.theta.copy_(params)
        # the reference re-binds `cost_loss` to the cost SURROGATE before returning (macpo.py:245), so that is what its
        # "Loss/Loss_cost_critic" column holds; kept as is
        return (value_loss, critic_grad_norm, kl, loss_improve, expected_improve, dist_entropy, ratio, cost_loss, cost_grad_norm,
                whether_recover_policy_value, cost_preds_batch, cost_returns_batch, B_cost_loss_grad, lam, nu, g_step_dir,
                b_step_dir, x, mu_old, std_old.expand_as(mu_old), B_cost_loss_grad_dot)

    def train(self, buffer, logger, perm_fn=None):
        """macpo.py:373-412: plain mean / std standardisation (+ 1e-5) of both advantages, ONE pass over the minibatches."""
        c = self.config
        self._sync_normalizer()

        def standardised(returns, preds):
            adv = returns[:-1] - self.value_normalizer.denormalize(preds[:-1])
            return (adv - torch.mean(adv)) / (torch.std(adv) + 1e-5)
        advantages = standardised(buffer.returns, buffer.value_preds)
        cost_adv = standardised(buffer.cost_returns, buffer.cost_preds)
        out = None
        perm = perm_fn(0) if perm_fn is not None else None
        for sample in buffer.feed_forward_generator(advantages, c["num_mini_batch"], cost_adv=cost_adv, perm=perm):
            out = self.trpo_update(sample)
            if logger is not None:
                logger.store(**{"Loss/Loss_reward_critic": out[0].item(), "Loss/Loss_cost_critic": out[7].item(),
                                "Loss/Loss_actor_improve": float(out[3]), "Loss/Loss_actor_expected_improve": float(out[4]),
                                "Misc/Reward_critic_norm": out[1].item(), "Misc/Cost_critic_norm": out[8].item(),
                                "Misc/Entropy": float(out[5]), "Misc/Ratio": out[6].detach().mean().item(), "Misc/KL": float(out[2])})
        return out

    ppo_update = trpo_update


class Runner(_base.Runner):
    """macpo.py:426-800."""
    policy_cls = MACPO_Policy
    trainer_cls = MACPO_Trainer
    log_keys = ("Loss/Loss_reward_critic", "Loss/Loss_cost_critic", "Loss/Loss_actor_improve", "Loss/Loss_actor_expected_improve",
                "Misc/Reward_critic_norm", "Misc/Cost_critic_norm", "Misc/Entropy", "Misc/Ratio", "Misc/KL")


def train(args, cfg_train):
    return _base.train(args, cfg_train, runner_cls=Runner)


if __name__ == "__main__":
    _base.cli("macpo", train)
