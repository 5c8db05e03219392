"""TEST INFRASTRUCTURE -- CPU restatement of the multi-agent MAPPO-L networks and trainer step.

Only tests/ may import this file (checker, never the product).  Pinned by tests/test_oracle_golden.py against
tests/golden/ma_mappolag.npz, which oracle/make_golden.py::golden_ma_mappolag produced by running the reference's own
MAPPO_L_Policy / MAPPO_L_Trainer here.  Each function cites the reference lines it restates.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class MANet(nn.Module):
    """MLPBase + head (safepo/utils/mlp.py:1-71; safepo/common/model.py:218-224, 322-331):
    LayerNorm(in) -> n_blocks x [Linear, ELU, LayerNorm(hidden)] -> Linear(hidden, out).  Actor nets own `log_std`
    (safepo/utils/distributions.py:36-37); std = sigmoid(log_std / x_coef) * y_coef (distributions.py:41).
    Parameter registration order equals the reference's state_dict order."""

    def __init__(self, in_dim, hidden, n_blocks, out_dim, is_actor, std_x_coef=1.0, std_y_coef=0.5):
        super().__init__()
        self.feature_norm = nn.LayerNorm(in_dim)
        self.blocks = nn.ModuleList()                      # per block: [Linear, LayerNorm] -> W_k, b_k, ln_k.weight, ln_k.bias
        d = in_dim
        for _ in range(n_blocks):
            self.blocks.append(nn.ModuleList([nn.Linear(d, hidden), nn.LayerNorm(hidden)]))
            d = hidden
        self.is_actor = bool(is_actor)
        if self.is_actor:
            self.log_std = nn.Parameter(torch.ones(out_dim) * std_x_coef)
        self.head = nn.Linear(hidden, out_dim)
        self.xc, self.yc = std_x_coef, std_y_coef

    def ordered_parameters(self):
        ps = [self.feature_norm.weight, self.feature_norm.bias]
        for lin, ln in self.blocks:
            ps += [lin.weight, lin.bias, ln.weight, ln.bias]
        if self.is_actor:
            ps.append(self.log_std)
        ps += [self.head.weight, self.head.bias]
        return ps

    def load_reference_state_dict(self, sd: dict):
        """`sd`: state_dict of the reference MultiAgentActor / MultiAgentCritic (same order as ordered_parameters)."""
        vals = list(sd.values())
        ps = self.ordered_parameters()
        assert len(vals) == len(ps), (len(vals), len(ps))
        with torch.no_grad():
            for p, v in zip(ps, vals):
                p.copy_(torch.as_tensor(v).reshape(p.shape))

    def flat(self) -> torch.Tensor:
        return torch.cat([p.detach().reshape(-1) for p in self.ordered_parameters()])

    def flat_grad(self) -> torch.Tensor:
        return torch.cat([p.grad.reshape(-1) for p in self.ordered_parameters()])

    def forward(self, x):
        h = self.feature_norm(x)
        for lin, ln in self.blocks:
            h = ln(nn.functional.elu(lin(h)))
        return self.head(h)

    def std(self):
        return torch.sigmoid(self.log_std / self.xc) * self.yc


def log_probs(mean, std, act):
    """FixedNormal.log_probs: per-dimension Normal.log_prob (distributions.py:9-11)."""
    return torch.distributions.Normal(mean, std).log_prob(act)


def huber_loss(e, d):
    """safepo/utils/util.py huber_loss (the branch for e < -d contributes nothing)."""
    a = (e.abs() <= d).float()
    b = (e > d).float()
    return a * e ** 2 / 2 + b * d * (e.abs() - d / 2)


class OraclePopArt:
    """safepo/common/popart.py:45-133 with input_shape 1."""

    def __init__(self, beta=0.99999, epsilon=1e-5):
        self.beta, self.epsilon = beta, epsilon
        self.running_mean = torch.zeros(1)
        self.running_mean_sq = torch.zeros(1)
        self.debiasing_term = torch.tensor(0.0)

    def mean_var(self):
        dm = self.running_mean / self.debiasing_term.clamp(min=self.epsilon)
        dsq = self.running_mean_sq / self.debiasing_term.clamp(min=self.epsilon)
        return dm, (dsq - dm ** 2).clamp(min=1e-2)

    def __call__(self, x, train=True):
        if train:
            d = x.detach()
            self.running_mean.mul_(self.beta).add_(d.mean(dim=0) * (1.0 - self.beta))
            self.running_mean_sq.mul_(self.beta).add_((d ** 2).mean(dim=0) * (1.0 - self.beta))
            self.debiasing_term.mul_(self.beta).add_(1.0 * (1.0 - self.beta))
        m, v = self.mean_var()
        return (x - m[None]) / torch.sqrt(v)[None]

    def denormalize(self, x):
        m, v = self.mean_var()
        return (x * torch.sqrt(v)[None] + m[None]).detach()


class OracleMATrainer:
    """MAPPO_L_Policy optimisers (mappolag.py:57-66) + MAPPO_L_Trainer.ppo_update (mappolag.py:140-199); with
    algo = "happo" / "mappo" the HAPPO_Trainer (happo.py:96-169) and MAPPO_Trainer (mappo.py:96-161) steps."""

    def __init__(self, cfg: dict, actor: MANet, critic: MANet, cost_critic: MANet | None = None, algo: str = "mappolag"):
        self.cfg, self.actor, self.critic, self.cost_critic, self.algo = cfg, actor, critic, cost_critic, algo
        mk = lambda net, lr: torch.optim.Adam(net.ordered_parameters(), lr=lr, eps=cfg["opti_eps"], weight_decay=cfg["weight_decay"])
        self.opt_a, self.opt_r = mk(actor, cfg["actor_lr"]), mk(critic, cfg["critic_lr"])
        self.opt_c = mk(cost_critic, cfg["critic_lr"]) if cost_critic is not None else None
        self.popart = OraclePopArt()
        self.lamda = torch.tensor(float(cfg.get("lamda_lagr", 0.0)))

    def value_loss(self, values, value_preds, returns, active=None):
        c = self.cfg
        vpc = value_preds + (values - value_preds).clamp(-c["clip_param"], c["clip_param"])
        e_c = self.popart(returns) - vpc              # each call updates the statistics (mappolag.py:129-130)
        e_o = self.popart(returns) - values
        vl = torch.max(huber_loss(e_o, c["huber_delta"]), huber_loss(e_c, c["huber_delta"]))
        if active is not None:                        # happo.py:117-120
            return (vl * active).sum() / active.sum()
        return vl.mean()

    def _ppo_update_unconstrained(self, s: dict):
        """happo.py:124-169 (joint ratio x factor) and mappo.py:119-161 (per-dimension ratios, no factor)."""
        c = self.cfg
        mean = self.actor(s["obs"])
        std = self.actor.std()
        logp = log_probs(mean, std, s["actions"])
        ent = torch.distributions.Normal(mean, std.expand_as(mean)).entropy()
        if c["use_policy_active_masks"]:
            dist_entropy = (ent * s["active_masks"]).sum() / s["active_masks"].sum()
        else:
            dist_entropy = ent.mean()
        values = self.critic(s["share_obs"])
        imp = torch.exp(logp - s["old_logp"])
        if self.algo == "happo":
            imp = torch.prod(imp, dim=-1, keepdim=True)
        surr1 = imp * s["adv"]
        surr2 = torch.clamp(imp, 1.0 - c["clip_param"], 1.0 + c["clip_param"]) * s["adv"]
        inner = torch.min(surr1, surr2)
        if self.algo == "happo":
            inner = s["factor"] * inner
        m = torch.sum(inner, dim=-1, keepdim=True)
        if c["use_policy_active_masks"]:
            policy_loss = (-m * s["active_masks"]).sum() / s["active_masks"].sum()
        else:
            policy_loss = -m.mean()
        self.opt_a.zero_grad()
        (policy_loss - dist_entropy * c["entropy_coef"]).backward()
        rec = {"actor_grad": self.actor.flat_grad().clone()}
        a_norm = nn.utils.clip_grad_norm_(self.actor.ordered_parameters(), c["max_grad_norm"])
        self.opt_a.step()
        masked = self.algo == "happo" and c.get("use_value_active_masks", False)
        vl = self.value_loss(values, s["value_preds"], s["returns"], s["active_masks"] if masked else None)
        self.opt_r.zero_grad()
        (vl * c["value_loss_coef"]).backward()
        rec["critic_grad"] = self.critic.flat_grad().clone()
        r_norm = nn.utils.clip_grad_norm_(self.critic.ordered_parameters(), c["max_grad_norm"])
        self.opt_r.step()
        f = lambda t: float(t.detach()) if torch.is_tensor(t) else float(t)
        rec["row"] = [f(vl), f(r_norm), f(policy_loss), f(dist_entropy), f(a_norm), f(imp.mean()),
                      float(self.popart.running_mean), float(self.popart.running_mean_sq), float(self.popart.debiasing_term)]
        return rec

    # ------------------------------------------------------------------ MACPO (macpo.py:153-371)
    def _actor_out(self, actor, s):
        mean = actor(s["obs"])
        std = actor.std().expand_as(mean)
        return mean, std, log_probs(mean, actor.std(), s["actions"])

    def _kl(self, s, new_actor, old_actor):
        mu, std, _ = self._actor_out(new_actor, s)
        mu_old, std_old, _ = self._actor_out(old_actor, s)
        mu_old, std_old = mu_old.detach(), std_old.detach()
        kl = torch.log(std_old) - torch.log(std) + (std_old.pow(2) + (mu_old - mu).pow(2)) / (1e-8 + 2.0 * std.pow(2)) - 0.5
        return kl.sum(1, keepdim=True)

    def _fvp(self, s, p):
        ps = self.actor.ordered_parameters()
        kl = self._kl(s, self.actor, self.actor).mean()
        g = torch.autograd.grad(kl, ps, create_graph=True, allow_unused=True)
        flat = torch.cat([x.reshape(-1) for x in g if x is not None])
        h = torch.autograd.grad((flat * p).sum(), ps, allow_unused=True)
        return torch.cat([x.contiguous().reshape(-1) for x in h if x is not None]).data + 0.1 * p

    def _cg(self, s, b, nsteps, residual_tol=1e-10):
        x = torch.zeros_like(b)
        r, p = b.clone(), b.clone()
        rdotr = torch.dot(r, r)
        for _ in range(nsteps):
            avp = self._fvp(s, p)
            alpha = rdotr / (torch.dot(p, avp) + 1e-8)
            x += alpha * p
            r -= alpha * avp
            new_rdotr = torch.dot(r, r)
            p = r + (new_rdotr / rdotr) * p
            rdotr = new_rdotr
            if rdotr < residual_tol:
                break
        return x

    def trpo_update(self, s: dict):
        import copy
        import numpy as np
        c = self.cfg
        values, cost_values = self.critic(s["share_obs"]), self.cost_critic(s["share_obs"])
        vl = self.value_loss(values, s["value_preds"], s["returns"])
        self.opt_r.zero_grad()
        (vl * c["value_loss_coef"]).backward()
        r_norm = nn.utils.clip_grad_norm_(self.critic.ordered_parameters(), c["max_grad_norm"])
        self.opt_r.step()
        cl = self.value_loss(cost_values, s["cost_preds"], s["cost_returns"])
        self.opt_c.zero_grad()
        (cl * c["value_loss_coef"]).backward()
        c_norm = nn.utils.clip_grad_norm_(self.cost_critic.ordered_parameters(), c["max_grad_norm"])
        self.opt_c.step()
        rescale = float((s["aver_episode_costs"].mean() - c["cost_limit"]) * (1 - c["gamma"]))
        if rescale == 0:
            rescale = 1e-8
        ps = self.actor.ordered_parameters()
        fl = lambda gs: torch.cat([g.reshape(-1) for g in gs if g is not None])
        _, _, logp = self._actor_out(self.actor, s)
        ratio = torch.prod(torch.exp(logp - s["old_logp"]), dim=-1, keepdim=True)
        reward_loss = -torch.sum(ratio * s["factor"] * s["adv"], dim=-1, keepdim=True).mean()
        g = fl(torch.autograd.grad(reward_loss, ps, retain_graph=True, allow_unused=True))
        cost_loss = torch.sum(ratio * s["factor"] * s["cost_adv"], dim=-1, keepdim=True).mean()
        b = fl(torch.autograd.grad(cost_loss, ps, retain_graph=True, allow_unused=True))
        iters = int(c["conjugate_gradient_iters"])
        g_dir, b_dir = self._cg(s, g.data.clone(), iters), self._cg(s, b.data.clone(), iters)
        q = float(torch.dot(g, g_dir))
        tkl = float(c["target_kl"])
        bb = float(torch.dot(b, b))
        if bb <= 1e-8 and rescale < 0:
            b_dir = torch.zeros_like(g_dir)
            r_c = s_c = pcv = wrp = 0.0
            case = 4
        else:
            r_c, s_c = float(torch.dot(g, b_dir)), float(torch.dot(b, b_dir))
            r_c = 1e-8 if r_c == 0 else r_c
            s_c = 1e-8 if s_c == 0 else s_c
            pcv = q - r_c ** 2 / (1e-8 + s_c)
            wrp = 2 * tkl - rescale ** 2 / (1e-8 + s_c)
            case = 3 if (rescale < 0 and wrp < 0) else 2 if (rescale < 0) else 1 if wrp >= 0 else 0
        if wrp == 0:
            wrp = 1e-8
        # the reference takes these roots of fp32 tensors; a float64 evaluation (the tests' yardstick) takes them in double
        f64 = self.actor.head.weight.dtype == torch.float64
        sqrt = (lambda v: math.sqrt(float(v))) if f64 else (lambda v: float(torch.sqrt(torch.tensor(float(v)))))
        if case in (3, 4):
            lam, nu = sqrt(q / (2 * tkl)), 0.0
        elif case in (1, 2):
            LA, LB = [0, r_c / rescale], [r_c / rescale, np.inf]
            LA, LB = (LA, LB) if rescale < 0 else (LB, LA)
            proj = lambda x, L: max(L[0], min(L[1], x))
            lam_a, lam_b = proj(sqrt(pcv / wrp), LA), proj(sqrt(q / (2 * tkl)), LB)
            f_a = lambda l: -0.5 * (pcv / (1e-8 + l) + wrp * l) - r_c * rescale / (1e-8 + s_c)
            f_b = lambda l: -0.5 * (q / (1e-8 + l) + 2 * tkl * l)
            lam = lam_a if f_a(lam_a) >= f_b(lam_b) else lam_b
            nu = max(0, lam * rescale - r_c) / (1e-8 + s_c)
        else:
            lam, nu = 0.0, sqrt(2 * tkl / (1e-8 + s_c))
        x = (1.0 / (lam + 1e-8)) * (g_dir + nu * b_dir) if case > 0 else nu * b_dir
        reward_loss, cost_loss = reward_loss.detach(), cost_loss.detach()
        params = self.actor.flat().clone()
        old_actor = copy.deepcopy(self.actor)
        expected = -torch.dot(x, g).detach()
        flag, kl, improve = False, torch.tensor(0.0), torch.tensor(0.0)

        def set_params(vec):
            off = 0
            with torch.no_grad():
                for prm in self.actor.ordered_parameters():
                    n = prm.numel()
                    prm.copy_(vec[off:off + n].view_as(prm))
                    off += n
        for i in range(int(c["searching_steps"])):
            xn = torch.norm(x)
            if xn > 0.5:
                x = x * 0.5 / xn
            set_params(params - c["fraction_coef"] * (c["step_fraction"] ** i) * x)
            with torch.no_grad():
                _, _, lp = self._actor_out(self.actor, s)
                ratio = torch.prod(torch.exp(lp - s["old_logp"]), dim=-1, keepdim=True)
                new_r = -torch.sum(ratio * s["factor"] * s["adv"], dim=-1, keepdim=True).mean()
                new_c = torch.sum(ratio * s["factor"] * s["cost_adv"], dim=-1, keepdim=True).mean()
                improve = new_r - reward_loss
                kl = self._kl(s, self.actor, old_actor).mean()
            if (kl < tkl) and (improve < 0 if case > 1 else True) and (new_c - cost_loss <= max(-rescale, 0)):
                flag = True
                break
            expected = expected * c["step_fraction"]
        if not flag:
            set_params(params)
        f = lambda t: float(t.detach()) if torch.is_tensor(t) else float(t)
        return {"row": [f(vl), f(r_norm), f(kl), f(improve), f(expected), f(cost_loss), f(c_norm), f(wrp), f(lam), f(nu), bb,
                        float(self.popart.running_mean), float(self.popart.running_mean_sq), float(self.popart.debiasing_term)],
                "case": case, "accepted": flag,
                "g": g.detach().clone(), "b": b.detach().clone(), "g_dir": g_dir.clone(), "b_dir": b_dir.clone(), "x": x.clone()}

    def ppo_update(self, s: dict):
        if self.algo == "macpo":
            return self.trpo_update(s)
        if self.algo != "mappolag":
            return self._ppo_update_unconstrained(s)
        c = self.cfg
        mean = self.actor(s["obs"])
        std = self.actor.std()
        logp = log_probs(mean, std, s["actions"])
        ent = torch.distributions.Normal(mean, std.expand_as(mean)).entropy()
        if c["use_policy_active_masks"]:
            dist_entropy = (ent * s["active_masks"]).sum() / s["active_masks"].sum()
        else:
            dist_entropy = ent.mean()
        values, cost_values = self.critic(s["share_obs"]), self.cost_critic(s["share_obs"])
        adv_h = s["adv"] - self.lamda * s["cost_adv"]
        imp = torch.prod(torch.exp(logp - s["old_logp"]), dim=-1, keepdim=True)
        surr1 = imp * adv_h
        surr2 = torch.clamp(imp, 1.0 - c["clip_param"], 1.0 + c["clip_param"]) * adv_h
        m = torch.sum(s["factor"] * torch.min(surr1, surr2), dim=-1, keepdim=True)
        if c["use_policy_active_masks"]:
            policy_loss = (-m * s["active_masks"]).sum() / s["active_masks"].sum()
        else:
            policy_loss = -m.mean()
        self.opt_a.zero_grad()
        (policy_loss - dist_entropy * c["entropy_coef"]).backward()
        rec = {"actor_grad": self.actor.flat_grad().clone()}
        a_norm = nn.utils.clip_grad_norm_(self.actor.ordered_parameters(), c["max_grad_norm"])
        self.opt_a.step()
        delta = -((s["aver_episode_costs"].mean() - c["cost_limit"]) * (1 - c["gamma"]) + (imp * s["cost_adv"])).mean().detach()
        self.lamda = torch.relu(self.lamda - delta * c["lagrangian_coef_rate"])
        vl = self.value_loss(values, s["value_preds"], s["returns"])
        self.opt_r.zero_grad()
        (vl * c["value_loss_coef"]).backward()
        rec["critic_grad"] = self.critic.flat_grad().clone()
        r_norm = nn.utils.clip_grad_norm_(self.critic.ordered_parameters(), c["max_grad_norm"])
        self.opt_r.step()
        cl = self.value_loss(cost_values, s["cost_preds"], s["cost_returns"])
        self.opt_c.zero_grad()
        (cl * c["value_loss_coef"]).backward()
        rec["cost_grad"] = self.cost_critic.flat_grad().clone()
        c_norm = nn.utils.clip_grad_norm_(self.cost_critic.ordered_parameters(), c["max_grad_norm"])
        self.opt_c.step()
        f = lambda t: float(t.detach()) if torch.is_tensor(t) else float(t)
        rec["row"] = [f(vl), f(r_norm), f(policy_loss), f(dist_entropy), f(a_norm), f(imp.mean()),
                      f(cl), f(c_norm), f(self.lamda), float(self.popart.running_mean),
                      float(self.popart.running_mean_sq), float(self.popart.debiasing_term)]
        return rec


def nets_from_golden(z, tag: str, which: str = "init"):
    """Build the three oracle nets of golden case `tag` from tests/golden/ma_mappolag.npz."""
    H, nb = int(z[f"{tag}_cfg_hidden_size"]), 1 + int(z[f"{tag}_cfg_layer_N"])
    xc, yc = float(z[f"{tag}_cfg_std_x_coef"]), float(z[f"{tag}_cfg_std_y_coef"])
    D, S, A = z[f"{tag}_obs"].shape[1], z[f"{tag}_share_obs"].shape[1], z[f"{tag}_actions"].shape[1]
    nets = {"actor": MANet(D, H, nb, A, True, xc, yc), "critic": MANet(S, H, nb, 1, False)}
    if any(k.startswith(f"{tag}_{which}_cost_critic_") for k in z.files):
        nets["cost_critic"] = MANet(S, H, nb, 1, False)
    for nm, net in nets.items():
        pre = f"{tag}_{which}_{nm}_"
        net.load_reference_state_dict({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
    return nets


def cfg_from_golden(z, tag: str) -> dict:
    pre = f"{tag}_cfg_"
    cfg = {k[len(pre):]: float(z[k]) for k in z.files if k.startswith(pre)}
    cfg["use_policy_active_masks"] = bool(cfg["use_policy_active_masks"])
    if "use_value_active_masks" in cfg:
        cfg["use_value_active_masks"] = bool(cfg["use_value_active_masks"])
    return cfg


def sample_from_golden(z, tag: str) -> dict:
    keys = ["share_obs", "obs", "actions", "value_preds", "returns", "active_masks", "old_logp", "adv", "factor", "cost_preds",
            "cost_returns", "cost_adv", "aver_episode_costs"]
    return {k: torch.from_numpy(z[f"{tag}_{k}"].copy()) for k in keys if f"{tag}_{k}" in z.files}


# ---------------------------------------------------------------------- Runner level (mappolag.py:475-504, 587-603)
def masked_gae(rewards, value_preds, masks, popart: OraclePopArt, gamma: float, lam: float):
    """SeparatedReplayBuffer.compute_returns with use_gae and PopArt (safepo/common/buffer.py:356-377): value_preds[-1]
    is the bootstrap value, fp32 tensor arithmetic in the reference's order."""
    T = rewards.shape[0]
    returns = torch.zeros_like(value_preds)
    gae = 0
    for step in reversed(range(T)):
        v_next, v_cur = popart.denormalize(value_preds[step + 1]), popart.denormalize(value_preds[step])
        delta = rewards[step] + gamma * v_next * masks[step + 1] - v_cur
        gae = delta + gamma * lam * masks[step + 1] * gae
        returns[step] = gae + v_cur
    return returns


def feed_forward_samples(buf: dict, advantages, cost_adv, perm, num_mini_batch: int):
    """feed_forward_generator (buffer.py:386-465) with the permutation supplied."""
    T, N = buf["rewards"].shape[0:2]
    mb = (T * N) // num_mini_batch
    flat = lambda t: t.reshape(-1, *t.shape[2:])
    cols = {"share_obs": flat(buf["share_obs"][:-1]), "obs": flat(buf["obs"][:-1]), "actions": flat(buf["actions"]),
            "value_preds": buf["value_preds"][:-1].reshape(-1, 1), "returns": buf["returns"][:-1].reshape(-1, 1),
            "active_masks": buf["active_masks"][:-1].reshape(-1, 1), "old_logp": flat(buf["action_log_probs"]),
            "adv": advantages.reshape(-1, 1), "factor": buf["factor"].reshape(-1, 1)}
    if cost_adv is not None:
        cols.update({"cost_preds": buf["cost_preds"][:-1].reshape(-1, 1), "cost_returns": buf["cost_returns"][:-1].reshape(-1, 1),
                     "cost_adv": cost_adv.reshape(-1, 1)})
    perm = torch.as_tensor(perm, dtype=torch.long)
    for i in range(num_mini_batch):
        idx = perm[i * mb:(i + 1) * mb]
        s = {k: v[idx] for k, v in cols.items()}
        if cost_adv is not None:
            s["aver_episode_costs"] = buf["aver_episode_costs"]
        yield s


def train_agent(tr: OracleMATrainer, buf: dict, perms, cfg: dict):
    """MAPPO_L_Trainer.train (mappolag.py:201-236): NaN-masked torch.mean / torch.std standardisation of both advantages,
    then learning_iters passes of num_mini_batch ppo_update steps.  Returns the rows the reference stores per pass."""
    if tr.algo == "macpo":
        # macpo.py:373-412: plain statistics (+ 1e-5) for both advantages, ONE pass over the minibatches
        def std_(returns, preds):
            adv = returns[:-1] - tr.popart.denormalize(preds[:-1])
            return (adv - torch.mean(adv)) / (torch.std(adv) + 1e-5)
        advantages, cost_adv = std_(buf["returns"], buf["value_preds"]), std_(buf["cost_returns"], buf["cost_preds"])
        return [tr.ppo_update(s_)["row"] for s_ in feed_forward_samples(buf, advantages, cost_adv, perms[0], int(cfg["num_mini_batch"]))]
    constrained = tr.algo == "mappolag"

    def standardised(returns, preds):
        adv = returns[:-1] - tr.popart.denormalize(preds[:-1])
        if not constrained:                            # happo.py:171-175 / mappo.py:163-167: plain statistics, + 1e-5
            return (adv - torch.mean(adv)) / (torch.std(adv) + 1e-5)
        cp = adv.clone()
        cp[buf["active_masks"][:-1] == 0.0] = float("nan")
        return (adv - torch.mean(cp)) / (torch.std(cp) + 1e-8)
    advantages = standardised(buf["returns"], buf["value_preds"])
    cost_adv = standardised(buf["cost_returns"], buf["cost_preds"]) if constrained else None
    rows = []
    for it in range(int(cfg["learning_iters"])):
        rec = None
        for s in feed_forward_samples(buf, advantages, cost_adv, perms[it], int(cfg["num_mini_batch"])):
            rec = tr.ppo_update(s)
        rows.append(rec["row"])
    return rows


def runner_train(trainers, bufs, order, perms_of, cfg: dict):
    """Runner.train (mappolag.py:475-504): agents in `order`, each with the running product of the previous agents'
    probability ratios as its factor."""
    T, N = bufs[0]["rewards"].shape[0:2]
    factor = torch.ones(T, N, 1)
    stored = []
    for a in order:
        b, tr = bufs[a], trainers[a]
        b["factor"] = factor.clone()
        obs, act = b["obs"][:-1].reshape(T * N, -1), b["actions"].reshape(T * N, -1)
        with torch.no_grad():
            old = log_probs(tr.actor(obs), tr.actor.std(), act)
        stored += train_agent(tr, b, perms_of[a], cfg)
        with torch.no_grad():
            new = log_probs(tr.actor(obs), tr.actor.std(), act)
        factor = factor * torch.prod(torch.exp(new - old).reshape(T, N, -1), dim=-1, keepdim=True)
    return stored
