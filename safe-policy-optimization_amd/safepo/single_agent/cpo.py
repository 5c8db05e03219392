"""CPO (Constrained Policy Optimization), MI355X-native hot path behind the reference entry point.

`main(args, cfg_env=None)` / `default_cfg` / CLI as in the reference script
(safepo/single_agent/cpo.py:47-54,160,604-651).  Collect and GAE are shared with PPO-Lag
(safepo.common.engine); the update (cpo.py:350-571) runs on HIP kernels:
  policy-gradient pair g, b          -> spo_cpo_surrogate_grad (full batch, MFMA fwd+bwd, fixed-order reduction)
  fvp (33 per epoch)                 -> spo_cpo_fvp (analytic Gauss-Newton product: forward tangent + backward)
  conjugate gradients (cpo.py:81-106)-> device vector ops around spo_cpo_fvp
  case analysis (cpo.py:384-463)     -> host scalars, exactly the reference's closed forms and comparisons
  line search (cpo.py:465-519)       -> spo_cpo_linesearch_eval per candidate
  critic fit (cpo.py:534-571)        -> spo_critic_fit_iter (persistent kernel, two critics, batch 128)
"""
from __future__ import annotations

import os
import random
import sys
import time

import numpy as np
import torch

from safepo import _abi
from safepo.common.wide import PermWindow
from safepo.common.engine import PPOLagEngine, _WideOps
from safepo.common.env import make_sa_mujoco_env
from safepo.common.logger import EpochLogger
from safepo.common.model import ActorVCritic
from safepo.parallel import dp_epoch_stat, init_from_env, require_equal_shards, shard_envs
from safepo.utils.config import isaac_gym_map, run_as_script

STEP_FRACTION = 0.8
CPO_SEARCHING_STEPS = 15
PCPO_SEARCHING_STEPS = 200          # reference pcpo.py:44 (its line search keeps halving far longer than CPO's, cpo.py:44)
CONJUGATE_GRADIENT_ITERS = 15

default_cfg = {
    'hidden_sizes': [64, 64],
    'gamma': 0.99,
    'target_kl': 0.01,
    'batch_size': 128,
    'learning_iters': 10,
    'max_grad_norm': 40.0,
}


class CPOEngine(PPOLagEngine):
    """Adds the CPO actor update and critic fit on top of the shared collect/GAE engine."""

    FAMILY = "cpo"

    def _require_policy(self, policy) -> None:
        policy._require_kernels("cpo")

    def __init__(self, policy: ActorVCritic, num_envs: int, steps: int, config: dict, device, comm=None):
        super().__init__(policy, num_envs, steps, config, device, comm=comm, lr=1e-3, critic_lr=1e-3)
        # data parallel over env shards: the actor update is full-batch, so it is EXACT -- the two
        # surrogate gradients, every Fisher-vector product and the line-search sums are all-reduced means; the critic fit
        # runs the persistent kernel with the in-kernel gradient exchange (needs peer-mapped regions).
        self._inv_world = 1.0 / self.comm.world_size
        self.ls_off = policy.log_std_offset
        self.Pa = policy.theta.numel() - self.ls_off
        self._alloc_full_batch_workspaces()
        self.loss_sum = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.ls_partials = torch.empty(3 * 1024, dtype=torch.float64, device=self.dev)
        self.ls_sums = torch.zeros(3, dtype=torch.float64, device=self.dev)
        self.stale_sq = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self._split = None          # two-launch form of the critic fit: None = not tried yet, False = unavailable

    def _alloc_full_batch_workspaces(self) -> None:
        nparts = self.lib.spo_cpo_num_partials(self.M)
        self.partial_ws = torch.empty(nparts * self.Pa, dtype=torch.float32, device=self.dev)
        self.loss_ws = torch.empty(nparts, dtype=torch.float64, device=self.dev)

    def _set_stale_actor_grad(self, vec: torch.Tensor) -> None:
        """actor.grad as the reference leaves it after the policy update: it still takes part in (and is rescaled by) the
        critic fit's clip_grad_norm_ over ALL policy parameters (cpo.py:557).  The persistent kernel only needs its norm."""
        self.stale_sq.copy_(vec.dot(vec).reshape(1))

    # ---------------------------------------------------------------- actor flat views (cpo.py:70-78,109-121)
    @property
    def theta_actor(self) -> torch.Tensor:
        return self.policy.theta[self.ls_off:]

    def _surrogate_grad_local(self, adv: torch.Tensor, sign: float) -> torch.Tensor:
        """This rank's d/dtheta_actor [sign * mean(ratio*adv)] (mean over the LOCAL rows); self.loss_sum = sum(ratio*adv)."""
        d = self.buffer.data
        g = torch.empty(self.Pa, dtype=torch.float32, device=self.dev)
        _abi.check(self.lib.spo_cpo_surrogate_grad(
            _abi.ptr(self.policy.theta), _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]),
            _abi.ptr(adv), float(sign), self.M, self.D, self.A, _abi.ptr(self.partial_ws), _abi.ptr(self.loss_ws),
            _abi.ptr(g), _abi.ptr(self.loss_sum), _abi.stream_ptr()), "spo_cpo_surrogate_grad")
        return g

    def surrogate_grad(self, adv: torch.Tensor, sign: float):
        """grad of sign*mean(ratio*adv) wrt the actor parameters, and mean(ratio*adv)."""
        g = self._surrogate_grad_local(adv, sign)
        if self.comm.world_size > 1:
            self.comm.all_reduce_sum_(g)
            g *= self._inv_world
            self.comm.all_reduce_sum_(self.loss_sum)
        return g, float(self.loss_sum.item()) / (self.M * self.comm.world_size)

    def _fvp_local(self, v: torch.Tensor) -> torch.Tensor:
        """J^T diag(1/sigma^2) J v / (M*A) over this rank's rows (0 on the log_std entries)."""
        out = torch.empty(self.Pa, dtype=torch.float32, device=self.dev)
        _abi.check(self.lib.spo_cpo_fvp(_abi.ptr(self.policy.theta), _abi.ptr(self.buffer.data["obs"]), _abi.ptr(v),
                                        self.M, self.D, self.A, _abi.ptr(self.partial_ws), _abi.ptr(self.loss_ws),
                                        _abi.ptr(out), _abi.stream_ptr()), "spo_cpo_fvp")
        return out

    def fvp(self, v: torch.Tensor) -> torch.Tensor:
        """cpo.py:132-157: H v + 0.1 v with H the Hessian of mean(KL(old||cur)) at cur == old."""
        v = v.contiguous()
        out = self._fvp_local(v)
        if self.comm.world_size > 1:
            self.comm.all_reduce_sum_(out)
            out *= self._inv_world
        out[:self.A] += (2.0 / self.A) * v[:self.A]          # log_std block of the Hessian
        return out + v * 0.1

    def conjugate_gradients(self, b: torch.Tensor, num_steps: int = CONJUGATE_GRADIENT_ITERS,
                            residual_tol: float = 1e-10, eps: float = 1e-6) -> torch.Tensor:
        """cpo.py:81-106 on device vectors."""
        x = torch.zeros_like(b)
        r = b - self.fvp(x)
        p = r.clone()
        rdotr = torch.dot(r, r)
        for _ in range(num_steps):
            z = self.fvp(p)
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

    def linesearch_eval(self, adv_a=None, adv_b=None):
        """(-mean(ratio*adv_a), mean(ratio*adv_b), kl) for the parameters currently in theta
        (cpo.py:473-491; adv_a/adv_b default to the buffer's adv_r/adv_c)."""
        d = self.buffer.data
        adv_a = d["adv_r"] if adv_a is None else adv_a
        adv_b = d["adv_c"] if adv_b is None else adv_b
        self._linesearch_sums_local(adv_a, adv_b)
        self.comm.all_reduce_sum_(self.ls_sums)
        s = self.ls_sums.cpu()
        Mg = self.M * self.comm.world_size
        return -float(s[0]) / Mg, float(s[1]) / Mg, float(s[2]) / (Mg * self.A)

    def _linesearch_sums_local(self, adv_a, adv_b) -> None:
        d = self.buffer.data
        _abi.check(self.lib.spo_cpo_linesearch_eval(
            _abi.ptr(self.policy.theta), _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]),
            _abi.ptr(adv_a), _abi.ptr(adv_b), _abi.ptr(self.mean_old), _abi.ptr(self.logstd_old),
            self.M, self.D, self.A, _abi.ptr(self.ls_partials), self.ls_partials.numel(), _abi.ptr(self.ls_sums),
            _abi.stream_ptr()), "spo_cpo_linesearch_eval")

    def policy_update(self, ep_costs: float, logger=None) -> dict:
        """cpo.py:350-532.  `ep_costs` = Jc - cost_limit.  Requires compute_gae() to have run."""
        target_kl = self.cfg["target_kl"]
        d = self.buffer.data
        theta_old = self.theta_actor.clone()
        self.snapshot_old_distribution()
        g_loss, mean_r = self.surrogate_grad(d["adv_r"], -1.0)          # grad of loss_pi_r = -mean(ratio*adv_r)
        loss_reward_before = -mean_r
        grads = -g_loss
        x = self.conjugate_gradients(grads)
        assert torch.isfinite(x).all(), "x is not finite"
        xHx = torch.dot(x, self.fvp(x))
        assert xHx.item() >= 0, "xHx is negative"
        alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
        b_grads, mean_c = self.surrogate_grad(d["adv_c"], 1.0)
        loss_cost_before = mean_c
        p = self.conjugate_gradients(b_grads)
        # host scalars for the case analysis (the reference mixes CPU tensors in here, cpo.py:390-391,413)
        q = xHx.detach().cpu()
        r = grads.dot(p).cpu()
        s = b_grads.dot(p).cpu()
        bb = b_grads.dot(b_grads).cpu()
        if bb <= 1e-6 and ep_costs < 0:
            A = torch.zeros(1)
            B = torch.zeros(1)
            optim_case = 4
        else:
            assert torch.isfinite(r).all(), "r is not finite"
            assert torch.isfinite(s).all(), "s is not finite"
            A = q - r ** 2 / (s + 1e-8)
            B = 2 * target_kl - ep_costs ** 2 / (s + 1e-8)
            if ep_costs < 0 and B < 0:
                optim_case = 3
            elif ep_costs < 0 <= B:
                optim_case = 2
            elif ep_costs >= 0 and B >= 0:
                optim_case = 1
                if logger:
                    logger.log("Alert! Attempting feasible recovery!", "yellow")
            else:
                optim_case = 0
                if logger:
                    logger.log("Alert! Attempting infeasible recovery!", "red")
        if optim_case in (3, 4):
            alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
            nu_star = torch.zeros(1)
            lambda_star = 1 / (alpha.cpu() + 1e-8)
            step_direction = alpha * x
        elif optim_case in (1, 2):
            lambda_a = torch.sqrt(A / B)
            lambda_b = torch.sqrt(q / (2 * target_kl))
            r_num = r.item()
            eps_cost = ep_costs + 1e-8
            if ep_costs < 0:
                lambda_a_star = torch.clamp(lambda_a, torch.as_tensor(0.0), r_num / eps_cost)
                lambda_b_star = torch.clamp(lambda_b, r_num / eps_cost, torch.as_tensor(torch.inf))
            else:
                lambda_a_star = torch.clamp(lambda_a, r_num / eps_cost, torch.as_tensor(torch.inf))
                lambda_b_star = torch.clamp(lambda_b, torch.as_tensor(0.0), r_num / eps_cost)
            f_a = -0.5 * (A / (lambda_a_star + 1e-8) + B * lambda_a_star) - r * ep_costs / (s + 1e-8)
            f_b = -0.5 * (q / (lambda_b_star + 1e-8) + 2 * target_kl * lambda_b_star)
            lambda_star = lambda_a_star if f_a >= f_b else lambda_b_star
            nu_star = torch.clamp(lambda_star * ep_costs - r, min=0) / (s + 1e-8)
            step_direction = (1.0 / (lambda_star + 1e-8)).to(self.dev) * (x - nu_star.to(self.dev) * p)
        else:
            lambda_star = torch.zeros(1)
            nu_star = torch.sqrt(2 * target_kl / (s + 1e-8))
            step_direction = -nu_star.to(self.dev) * p

        step_frac, step_direction, acceptance_step, kl = self._constrained_line_search(
            theta_old, step_direction, grads, optim_case, ep_costs, loss_reward_before, loss_cost_before, logger)
        self.theta_actor.copy_(theta_old + step_frac * step_direction)
        # the actor's .grad keeps the cost gradient b: it takes part in the critic fit's joint clip (cpo.py:557)
        self._set_stale_actor_grad(b_grads)
        return {"alpha": float(alpha), "final_step_norm": float(torch.norm(step_direction)), "xHx": float(xHx),
                "gradient_norm": float(torch.norm(grads)), "H_inv_g": float(x.norm()),
                "acceptance_step": acceptance_step, "loss_actor": loss_reward_before + loss_cost_before, "kl": kl,
                "case": optim_case, "g": grads, "b": b_grads, "x": x, "p": p, "step_direction": step_direction}

    def trust_region_update(self, advantage: torch.Tensor, line_search: bool, logger=None) -> dict:
        """Natural-gradient step x*alpha with alpha = sqrt(2*delta / xHx) (natural_pg.py:350-381), optionally
        followed by TRPO's backtracking line search on surrogate improvement and KL (trpo.py:384-428).
        `advantage`: flat device tensor (adv_r, or the Lagrangian mix for rcpo / trpo_lag)."""
        target_kl = self.cfg["target_kl"]
        theta_old = self.theta_actor.clone()
        self.snapshot_old_distribution()
        g_loss, mean_ra = self.surrogate_grad(advantage, -1.0)          # loss_pi = -mean(ratio*advantage)
        loss_before = -mean_ra
        grads = -g_loss
        x = self.conjugate_gradients(grads)
        assert torch.isfinite(x).all(), "x is not finite"
        xHx = torch.dot(x, self.fvp(x))
        assert xHx.item() >= 0, "xHx is negative"
        alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
        step_direction = x * alpha
        assert torch.isfinite(step_direction).all(), "step_direction is not finite"
        acceptance_step = None
        if not line_search:
            self.theta_actor.copy_(theta_old + step_direction)
            _, _, final_kl = self.linesearch_eval(advantage, advantage)
        else:
            step_frac, final_kl, acceptance_step = 1.0, 0.0, 0
            loss_pi = loss_before
            expected_improve = grads.dot(step_direction)
            for step in range(CPO_SEARCHING_STEPS):
                self.theta_actor.copy_(theta_old + step_frac * step_direction)
                loss_pi, _, kl = self.linesearch_eval(advantage, advantage)
                loss_improve = loss_before - loss_pi
                if logger:
                    logger.log(f"Expected Improvement: {expected_improve} Actual: {loss_improve}")
                if not np.isfinite(loss_pi):
                    if logger:
                        logger.log("WARNING: loss_pi not finite")
                elif loss_improve < 0:
                    if logger:
                        logger.log("INFO: did not improve improve <0")
                elif kl > target_kl:
                    if logger:
                        logger.log("INFO: violated KL constraint.")
                else:
                    acceptance_step = step + 1
                    if logger:
                        logger.log(f"Accept step at i={acceptance_step}")
                    final_kl = kl
                    break
                step_frac *= STEP_FRACTION
            else:
                if logger:
                    logger.log("INFO: no suitable step found...")
                step_direction = torch.zeros_like(step_direction)
                acceptance_step = 0
            self.theta_actor.copy_(theta_old + step_frac * step_direction)
        # actor.grad keeps d(loss_pi)/d(theta) = -grads: part of the critic fit's joint clip_grad_norm_
        self._set_stale_actor_grad(-grads)
        # Loss/Loss_actor: natural_pg logs the surrogate before the step (natural_pg.py:390), trpo the one of the
        # last line-search candidate (trpo.py:438)
        return {"alpha": float(alpha), "final_step_norm": float(torch.norm(step_direction)), "xHx": float(xHx),
                "gradient_norm": float(torch.norm(grads)), "H_inv_g": float(x.norm()),
                "acceptance_step": acceptance_step, "loss_actor": (loss_pi if line_search else loss_before),
                "kl": final_kl, "g": grads, "x": x}

    def _constrained_line_search(self, theta_old, step_direction, grads, optim_case, ep_costs, loss_reward_before,
                                 loss_cost_before, logger=None, max_steps: int = CPO_SEARCHING_STEPS):
        """Backtracking search shared by CPO (cpo.py:465-519, 15 candidates) and PCPO (pcpo.py:404-458, 200 candidates)."""
        target_kl = self.cfg["target_kl"]
        step_frac = 1.0
        expected_reward_improve = grads.dot(step_direction)
        kl = 0.0
        acceptance_step = 0
        for step in range(max_steps):
            self.theta_actor.copy_(theta_old + step_frac * step_direction)
            acceptance_step = step + 1
            loss_reward, loss_cost, kl = self.linesearch_eval()
            loss_reward_improve = loss_reward_before - loss_reward
            loss_cost_diff = loss_cost - loss_cost_before
            if logger:
                logger.log(f"Expected Improvement: {expected_reward_improve} Actual: {loss_reward_improve}")
            if not np.isfinite(kl):
                if logger:
                    logger.log("WARNING: KL not finite")
                continue
            if (loss_reward_improve < 0) if optim_case > 1 else False:
                if logger:
                    logger.log("INFO: did not improve improve <0")
            elif loss_cost_diff > max(-ep_costs, 0):
                if logger:
                    logger.log(f"INFO: no improve {loss_cost_diff} > {max(-ep_costs, 0)}")
            elif kl > target_kl:
                if logger:
                    logger.log(f"INFO: violated KL constraint {kl} at step {step + 1}.")
            else:
                if logger:
                    logger.log(f"Accept step at i={step + 1}")
                break
            step_frac *= STEP_FRACTION
        else:
            if logger:
                logger.log("INFO: no suitable step found...")
            step_direction = torch.zeros_like(step_direction)
            acceptance_step = 0
        return step_frac, step_direction, acceptance_step, kl

    def pcpo_update(self, ep_costs: float, logger=None) -> dict:
        """PCPO (reference safepo/single_agent/pcpo.py:352-470): reward step sqrt(2*delta/q) * H x projected onto
        the cost constraint, step = sqrt(2d/(q+1e-8)) * Hx - max(0, (sqrt(2d/q) r + c) / s) * p, then CPO's line
        search with optim_case fixed to 0."""
        target_kl = self.cfg["target_kl"]
        d = self.buffer.data
        theta_old = self.theta_actor.clone()
        self.snapshot_old_distribution()
        g_loss, mean_r = self.surrogate_grad(d["adv_r"], -1.0)
        loss_reward_before = -mean_r
        grads = -g_loss
        x = self.conjugate_gradients(grads)
        assert torch.isfinite(x).all(), "x is not finite"
        Hx = self.fvp(x)                               # the reference names this H_inv_g (pcpo.py:372)
        xHx = torch.dot(x, Hx)
        assert xHx.item() >= 0, "xHx is negative"
        alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
        b_grads, mean_c = self.surrogate_grad(d["adv_c"], 1.0)
        loss_cost_before = mean_c
        p = self.conjugate_gradients(b_grads)
        q = xHx
        r = grads.dot(p)
        s_ = b_grads.dot(p)
        step_direction = (torch.sqrt(2 * target_kl / (q + 1e-8)) * Hx
                          - torch.clamp_min((torch.sqrt(2 * target_kl / q) * r + ep_costs) / s_,
                                            torch.tensor(0.0, device=self.dev)) * p)
        step_frac, step_direction, acceptance_step, kl = self._constrained_line_search(
            theta_old, step_direction, grads, 0, ep_costs, loss_reward_before, loss_cost_before, logger,
            max_steps=PCPO_SEARCHING_STEPS)
        self.theta_actor.copy_(theta_old + step_frac * step_direction)
        self._set_stale_actor_grad(b_grads)
        return {"alpha": float(alpha), "final_step_norm": float(torch.norm(step_direction)), "xHx": float(xHx),
                "gradient_norm": float(torch.norm(grads)), "H_inv_g": float(x.norm()),
                "acceptance_step": acceptance_step, "loss_actor": loss_reward_before + loss_cost_before, "kl": kl,
                "case": 0, "g": grads, "b": b_grads, "x": x, "p": p, "step_direction": step_direction}

    # ---------------------------------------------------------------- critic fit on two workgroup pairs (one GPU)
    def _split_setup(self):
        """The critic fit steps through minibatches of 128 rows, which one workgroup per critic takes as two 64-column
        halves one after the other (the MFMA tile is 16 columns per wave, 4 waves).  On a single GPU the machinery that
        shards the step over ranks shards it over CUs instead: ONE launch whose four workgroups are two "ranks" (each
        with its own replica of the critics and 64 of every 128 rows), exchanging the gradient inside the step through a
        pair of exchange regions (spo_critic_fit_iter_split; no IPC, no second stream).  Same arithmetic as the mean over
        128 rows, ~15 us instead of ~20 us per step.  The four workgroups of one grid are co-resident by construction
        (rounds 1-2 used two launches on two streams, whose co-residency depended on HIP's stream-to-queue assignment);
        the exchange self-test in the same one-grid shape runs once as an assertion.
        Returns the state dict, or None when the form does not apply (then the one-launch form is used)."""
        if self._split is not None:
            return self._split or None
        self._split = False
        batch = int(self.cfg.get("batch_size", 0))
        mode = os.environ.get("SPO_CPO_SPLIT", "auto")
        if self.comm.world_size != 1 or batch != 128 or self.M % batch != 0 or mode == "0":
            return None
        # round 6: spo_critic_fit_iter runs the ROW-SPLIT kernel (csrc/update_rs.hip: four workgroups of 32 rows per critic, ONE
        # replica, one hand-off through the XCD's L2) where the shape fits -- faster than this two-replica form, which stays for
        # SPO_UPDATE_FORM < 3, for shapes the row-split kernel does not take, and on request (SPO_CPO_SPLIT=force: tests)
        if (mode != "force" and int(os.environ.get("SPO_UPDATE_FORM", "3")) >= 3
                and self.lib.spo_update_rs_supported(self.D, self.A, batch, 2)):
            return None
        import ctypes
        lib = self.lib
        regions = (ctypes.c_void_p * 8)()
        owns = []
        try:
            for r in range(2):
                own, handle = ctypes.c_void_p(), (ctypes.c_ubyte * 64)()
                _abi.check(lib.spo_p2p_alloc(ctypes.byref(own), handle), "spo_p2p_alloc")
                owns.append(own)
                regions[r] = own
        except _abi.SpoError as e:
            for own in owns:
                lib.spo_p2p_free(own)
            print(f"[cpo] split critic fit unavailable ({e}); using the one-launch form", file=sys.stderr)
            return None
        res = torch.zeros(4, dtype=torch.int32, device=self.dev)
        _abi.check(lib.spo_p2p_selftest_one_grid(regions, 0, 64, _abi.ptr(res), _abi.stream_ptr()), "spo_p2p_selftest_one_grid")
        got = res.tolist()
        if got != [0, 0, 0, 0]:
            for own in owns:
                lib.spo_p2p_free(own)
            # HIP does not promise that the workgroups of a plain launch are co-resident (a shared or busy GPU): degrade to
            # the one-launch form like the runtime timeout path of _critic_fit_split does, do not abort the training run
            print(f"[cpo] one-grid exchange self-test failed: {got} ({{wrong values, timeout}} per rank); using the "
                  "one-launch form of the critic fit", file=sys.stderr)
            return None
        th = self.policy.theta
        st = {"regions": regions, "owns": owns, "step": 64,
              "sync_ws": torch.zeros(32, dtype=torch.int64, device=self.dev),
              "theta1": torch.empty_like(th), "m1": torch.empty_like(th), "v1": torch.empty_like(th),
              "stale1": torch.empty_like(self.stale_sq)}
        self._split = st
        return st

    def __del__(self):
        st = getattr(self, "_split", None)
        if st:                       # release the two exchange regions of the split critic fit
            try:
                torch.cuda.synchronize(self.dev)
                for own in st["owns"]:
                    self.lib.spo_p2p_free(own)
            except Exception:        # noqa: BLE001 -- interpreter shutdown
                pass
            self._split = False

    def _critic_fit_split(self, st, perm_fn, cfg64, n_mb):
        """learning_iters passes with the one-grid split form; None if an exchange timed out (state restored)."""
        c, d, lib = self.cfg, self.buffer.data, self.lib
        th, m, v = self.policy.theta, self.adam_m, self.adam_v
        backup = (th.clone(), m.clone(), v.clone(), self.stale_sq.clone(), self.adam_step)
        th1, m1, v1, stale1 = st["theta1"], st["m1"], st["v1"], st["stale1"]
        th1.copy_(th); m1.copy_(m); v1.copy_(v); stale1.copy_(self.stale_sq)
        half = self.M // 2
        all_losses = []
        for it in range(c["learning_iters"]):
            perm = _abi.require_gpu_tensor(perm_fn(it), "perm", torch.int32).view(n_mb, 128)
            p0, p1 = perm[:, :64].contiguous().view(-1), perm[:, 64:].contiguous().view(-1)
            l0 = torch.empty((n_mb, 3), dtype=torch.float32, device=self.dev)
            l1 = torch.empty_like(l0)
            _abi.check(lib.spo_critic_fit_iter_split(
                _abi.ptr(th), _abi.ptr(m), _abi.ptr(v), _abi.ptr(th1), _abi.ptr(m1), _abi.ptr(v1), self.adam_step,
                _abi.ptr(d["obs"]), _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(p0), _abi.ptr(p1), half,
                cfg64, _abi.ptr(self.stale_sq), _abi.ptr(stale1), _abi.ptr(l0), _abi.ptr(l1), _abi.ptr(self.sync_ws),
                _abi.ptr(st["sync_ws"]), st["regions"], st["step"] & 0xFFFFFFFF, _abi.stream_ptr()), "spo_critic_fit_iter_split")
            st["step"] += n_mb
            self.adam_step += n_mb
            all_losses.append(((l0 + l1) * 0.5)[:, :2])
        code = (int(self.sync_ws[8].item()) | int(st["sync_ws"][8].item())) & 0xFFFFFFFF
        if code:
            # an exchange timed out (cannot happen while the four workgroups are resident): restore, one-launch form for good
            print(f"[cpo] split critic fit: exchange error {code}; falling back to the one-launch form", file=sys.stderr)
            self.sync_ws[8] = 0
            st["sync_ws"][8] = 0
            th.copy_(backup[0]); m.copy_(backup[1]); v.copy_(backup[2]); self.stale_sq.copy_(backup[3])
            self.adam_step = backup[4]
            for own in st["owns"]:
                lib.spo_p2p_free(own)
            self._split = False
            return None
        return all_losses

    def critic_fit(self, perm_fn=None):
        """cpo.py:534-571: learning_iters passes of minibatches (batch_size rows) over both critics."""
        c = self.cfg
        cfg = self._cfg_struct()
        d = self.buffer.data
        if perm_fn is None:
            perm_fn = lambda it: torch.randperm(self.M, device=self.dev).to(torch.int32)
        n_mb = (self.M + cfg.batch - 1) // cfg.batch
        st = self._split_setup()
        if st is not None:
            cfg64 = self._cfg_struct()
            cfg64.batch = 64
            done = self._critic_fit_split(st, perm_fn, cfg64, n_mb)
            if done is not None:
                means = torch.cat(done, 0).mean(0).tolist() if done else [float("nan")] * 2
                return {"loss_r": means[0], "loss_c": means[1], "losses": done}
        all_losses = []
        for it in range(c["learning_iters"]):
            perm = _abi.require_gpu_tensor(perm_fn(it), "perm", torch.int32)
            losses = torch.empty((n_mb, 3), dtype=torch.float32, device=self.dev)
            if self.comm.world_size > 1:
                px = self.p2p
                if px is None:
                    raise NotImplementedError("data-parallel CPO needs the in-kernel gradient exchange (peer-mapped "
                                              "regions) for the critic fit; it is unavailable on this node")
                _abi.check(self.lib.spo_critic_fit_iter_dp(
                    _abi.ptr(self.policy.theta), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), self.adam_step,
                    _abi.ptr(d["obs"]), _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(perm),
                    self.M, cfg, _abi.ptr(self.stale_sq), _abi.ptr(losses), _abi.ptr(self.sync_ws), px.rank, px.world,
                    px.regions, px.step & 0xFFFFFFFF, _abi.stream_ptr()), "spo_critic_fit_iter_dp")
                px.step += n_mb
                self.comm.all_reduce_sum_(losses)
                losses *= self._inv_world
            else:
                _abi.check(self.lib.spo_critic_fit_iter(
                    _abi.ptr(self.policy.theta), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), self.adam_step,
                    _abi.ptr(d["obs"]), _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(perm),
                    self.M, cfg, _abi.ptr(self.stale_sq), _abi.ptr(losses), _abi.ptr(self.sync_ws), _abi.stream_ptr()),
                    "spo_critic_fit_iter")
            self.adam_step += n_mb
            all_losses.append(losses[:, :2])
        self.check_sync_error()
        means = torch.cat(all_losses, 0).mean(0).tolist() if all_losses else [float("nan")] * 2
        return {"loss_r": means[0], "loss_c": means[1], "losses": all_losses}


class WideCPOEngine(_WideOps, CPOEngine):
    """CPOEngine for an ActorVCritic outside the envelope of the LDS-resident full-batch kernels (obs_dim > 64, act_dim > 16 or
    hidden_sizes other than [64, 64]): the reference's default sweep pairs cpo / pcpo / rcpo / trpo_lag with 72-88-dim Car /
    Doggo / Racecar observations and with HumanoidVelocity's 376 / 17 (single_agent/benchmark.py:5-44; model.py:131 takes any
    dims).  Same update code (policy_update / trust_region_update / pcpo_update are inherited unchanged); the three full-batch
    primitives run on the wide-network kernels in row chunks:
      surrogate gradient   spo_mlp_forward -> spo_wide_actor_loss(mode 1) -> spo_mlp_backward          (cpo.py:356-381)
      Fisher-vector prod.  spo_mlp_forward -> spo_mlp_jvp -> spo_wide_fvp_cotangent -> spo_mlp_backward (cpo.py:132-157;
                           J^T diag(1/sigma^2) J v / (M A) applied analytically, no double backward)
      line search sums     spo_mlp_forward -> spo_wide_linesearch_sums                                  (cpo.py:473-491)
    The critic fit keeps the persistent two-critic kernel whenever the CRITICS fit it (hidden [64, 64], obs_dim <= 128: their
    layout does not depend on act_dim), else runs minibatch by minibatch on the wide kernels with the actor's stale gradient
    kept in the flat gradient vector so the joint clip sees and rescales it (cpo.py:557).  Data-parallel like CPOEngine: the
    full-batch quantities are all-reduced means, the wide critic fit all-reduces the critics' flat gradient per minibatch step."""

    FAMILY = "cpo"          # what _WideOps._require_policy checks the policy against
    CHUNK = 65536           # rows per full-batch pass; raised in __init__ for narrow networks (activations stay under ~0.5 GB)

    def __init__(self, policy: ActorVCritic, num_envs: int, steps: int, config: dict, device, comm=None):
        super().__init__(policy, num_envs, steps, config, device, comm=comm)
        self._wide_init()
        w = self.wide
        assert self.Pa == w.A + w.Pa
        self._gtmp = torch.zeros(w.P, dtype=torch.float32, device=self.dev)
        # one pass of ~40 launches per chunk: a [64, 64] network at 376 observations takes the whole 524 288-row batch in one
        # chunk (activations 0.3 GB), [1024, 1024, 512] stays at 65 536 rows
        per_row = sum(policy.hidden_sizes) + policy.act_dim + policy.obs_dim
        self.CHUNK = max(self.CHUNK, min(self.M, (1 << 27) // max(per_row, 1)))
        self._critics_on_persistent_kernel = (list(policy.hidden_sizes) == [64, 64] and policy.obs_dim <= _abi.MAX_OBS)
        if not self._critics_on_persistent_kernel:
            self.p2p = None                       # (the in-kernel exchange belongs to the persistent critic fit)

    def _alloc_full_batch_workspaces(self) -> None:
        self.partial_ws = self.loss_ws = None           # the LDS-resident kernels' per-workgroup partial vectors: not used here

    def _feature_split_critic_fit_ok(self, cfg) -> bool:
        return (list(self.policy.hidden_sizes) == [64, 64] and self.comm.world_size == 1
                and os.environ.get("SPO_WIDE_KS", "1") != "0" and bool(self.lib.spo_critic_fit_ks_supported(self.D, int(cfg.batch))))

    def _set_stale_actor_grad(self, vec: torch.Tensor) -> None:
        super()._set_stale_actor_grad(vec)
        self.flat_grad[self.ls_off:].copy_(vec)         # the wide critic fit clips over (and rescales) the vector itself

    def _chunks(self):
        d = self.buffer.data
        obs, act = d["obs"].view(self.M, self.D), d["act"].view(self.M, self.A)
        for lo in range(0, self.M, self.CHUNK):
            hi = min(lo + self.CHUNK, self.M)
            yield lo, hi, obs[lo:hi], act[lo:hi]

    def _log_std(self):
        return self.policy.theta[self.ls_off:self.ls_off + self.A]

    def _surrogate_grad_local(self, adv: torch.Tensor, sign: float) -> torch.Tensor:
        w, lib, d = self.wide, self.lib, self.buffer.data
        adv, logp_old = adv.reshape(-1), d["log_prob"].view(-1)
        g = torch.zeros(self.Pa, dtype=torch.float32, device=self.dev)
        for k, (lo, hi, obs, act) in enumerate(self._chunks()):
            mu, ws = w.forward("a", obs, slot=2)
            d_mu = torch.empty((hi - lo, self.A), dtype=torch.float32, device=self.dev)
            _abi.check(lib.spo_wide_actor_loss(_abi.WIDE_ACTOR_SURR, _abi.ptr(mu), _abi.ptr(self._log_std()), _abi.ptr(act),
                                               _abi.ptr(logp_old[lo:hi]), _abi.ptr(adv[lo:hi]), None, None, hi - lo, self.M, self.A,
                                               float(sign), 0.0, _abi.ptr(d_mu), _abi.ptr(self.actor_sums), int(k > 0), None, None,
                                               _abi.ptr(self.loss_partials), self.loss_partials.numel(), _abi.stream_ptr()),
                       "spo_wide_actor_loss")
            w.backward("a", obs, ws, d_mu, self._gtmp)
            g[self.A:] += self._gtmp[w.off_a:]
        g[:self.A] = self.actor_sums[2:2 + self.A].to(torch.float32)
        self.loss_sum.copy_(self.actor_sums[0:1])
        return g

    def _fvp_local(self, v: torch.Tensor) -> torch.Tensor:
        w, lib = self.wide, self.lib
        tangent = v[self.A:].contiguous()
        out = torch.zeros(self.Pa, dtype=torch.float32, device=self.dev)
        for lo, hi, obs, _ in self._chunks():
            rows = hi - lo
            _, ws = w.forward("a", obs, slot=2)
            jv = torch.empty((rows, self.A), dtype=torch.float32, device=self.dev)
            sc = w.jvp_scratch(rows)
            _abi.check(lib.spo_mlp_jvp(_abi.ptr(w.theta_of("a")), w.net_a, _abi.ptr(tangent), _abi.ptr(obs), rows, _abi.ptr(ws),
                                       _abi.ptr(jv), _abi.ptr(sc), _abi.stream_ptr()), "spo_mlp_jvp")
            _abi.check(lib.spo_wide_fvp_cotangent(_abi.ptr(jv), _abi.ptr(self._log_std()), rows, self.M, self.A, _abi.ptr(jv),
                                                  _abi.stream_ptr()), "spo_wide_fvp_cotangent")
            w.backward("a", obs, ws, jv, self._gtmp)
            out[self.A:] += self._gtmp[w.off_a:]
        return out

    def _linesearch_sums_local(self, adv_a, adv_b) -> None:
        w, lib, d = self.wide, self.lib, self.buffer.data
        adv_a, adv_b, logp_old = adv_a.reshape(-1), adv_b.reshape(-1), d["log_prob"].view(-1)
        for k, (lo, hi, obs, act) in enumerate(self._chunks()):
            mu, _ = w.forward("a", obs, slot=2)
            _abi.check(lib.spo_wide_linesearch_sums(_abi.ptr(mu), _abi.ptr(self._log_std()), _abi.ptr(act), _abi.ptr(logp_old[lo:hi]),
                                                    _abi.ptr(adv_a[lo:hi]), _abi.ptr(adv_b[lo:hi]), _abi.ptr(self.mean_old[lo:hi]),
                                                    _abi.ptr(self.logstd_old), hi - lo, self.A, _abi.ptr(self.ls_partials),
                                                    self.ls_partials.numel(), _abi.ptr(self.ls_sums), int(k > 0), _abi.stream_ptr()),
                       "spo_wide_linesearch_sums")

    def _cfg_struct(self):
        cfg = super()._cfg_struct()
        if self._in_persistent_critic_fit:
            cfg.act_dim = min(cfg.act_dim, _abi.MAX_ACT)      # the two-critic kernel never touches the actor: its layout is act-free
        return cfg

    _in_persistent_critic_fit = False

    def critic_fit(self, perm_fn=None):
        """cpo.py:534-571."""
        if self._critics_on_persistent_kernel:
            self._in_persistent_critic_fit = True
            try:
                return CPOEngine.critic_fit(self, perm_fn)
            finally:
                self._in_persistent_critic_fit = False
        c, w, lib, d = self.cfg, self.wide, self.lib, self.buffer.data
        cfg = self._cfg_struct()
        if perm_fn is None:
            perm_fn = lambda it: torch.randperm(self.M, device=self.dev).to(torch.int32)
        n_mb = (self.M + cfg.batch - 1) // cfg.batch
        if self._feature_split_critic_fit_ok(cfg):
            # hidden [64, 64] critics with obs_dim <= 512 (HumanoidVelocity): every learning iteration is ONE launch of the
            # persistent feature-split kernel with two networks (csrc/update_ks.hip, round 5); it carries the NORM of the actor's
            # stale gradient like the LDS-resident critic fit, the vector is rescaled to match afterwards
            stale0 = self.stale_sq.clone()
            all_losses = []
            for it in range(c["learning_iters"]):
                perm = _abi.require_gpu_tensor(perm_fn(it), "perm", torch.int32)
                losses = torch.empty((n_mb, 3), dtype=torch.float32, device=self.dev)
                _abi.check(lib.spo_critic_fit_iter_ks(
                    _abi.ptr(self.policy.theta), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), self.adam_step,
                    _abi.ptr(d["obs"]), _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(perm),
                    self.M, cfg, _abi.ptr(self.stale_sq), _abi.ptr(losses), _abi.ptr(self.sync_ws), _abi.stream_ptr()),
                    "spo_critic_fit_iter_ks")
                self.adam_step += n_mb
                all_losses.append(losses[:, :2])
            if int(self.sync_ws[8].item()) & 0xFFFFFFFF:
                self.sync_ws[8] = 0
                raise _abi.SpoError("feature-split critic fit: inter-workgroup exchange timed out")
            self.flat_grad[self.ls_off:].mul_(torch.sqrt(self.stale_sq / stale0.clamp_min(1e-38)))
            means = torch.cat(all_losses, 0).mean(0).tolist() if all_losses else [float("nan")] * 2
            return {"loss_r": means[0], "loss_c": means[1], "losses": all_losses}
        obs_all = d["obs"].view(self.M, self.D)
        tr_all, tc_all = d["target_value_r"].view(-1), d["target_value_c"].view(-1)
        g, part, cap = self.flat_grad, self.loss_partials, self.loss_partials.numel()
        all_losses = []
        def step(idx, loss3, dev_clock):
            if w.rows_grad_ok(idx.numel(), critics_only=True):
                # round 6: gather + both critics' forward / MSE / backward in one launch split over 16-row groups (csrc/mlp_rows.hip),
                # then the fixed-order sum of the groups into g[:2 Pc] (the actor's stale gradient behind it is not touched)
                w.grad_rows(idx, obs_all, None, None, tr_all, tc_all, None, 0.0, g, loss3, critics_only=True)
            else:
                obs, tgt_r, tgt_c = w.gather_rows(idx, [obs_all, tr_all.view(-1, 1), tc_all.view(-1, 1)])
                tgt_r, tgt_c = tgt_r.view(-1), tgt_c.view(-1)
                n = obs.shape[0]
                (v_r, ws_r), (v_c, ws_c) = w.forward_multi("rc", obs, slot=1)
                d_vr = torch.empty(n, dtype=torch.float32, device=self.dev)
                d_vc = torch.empty_like(d_vr)
                _abi.check(lib.spo_wide_critic_loss(_abi.ptr(v_r), _abi.ptr(v_c), _abi.ptr(tgt_r), _abi.ptr(tgt_c), n, _abi.ptr(d_vr),
                                                    _abi.ptr(d_vc), _abi.ptr(loss3), _abi.ptr(part), cap, _abi.stream_ptr()),
                           "spo_wide_critic_loss")
                w.backward_multi("rc", obs, [ws_r, ws_c], [d_vr, d_vc], g)
            self._reduce_flat_grad(0, w.off_ls)       # data-parallel: the critics' gradient of the global minibatch
            if dev_clock:
                self._clip_adam_dev(cfg, 0, w.off_ls, 0, 1, loss3, loss3, idx if isinstance(idx, PermWindow) else None)
            else:
                _abi.check(lib.spo_wide_clip_adam_ex(_abi.ptr(self.policy.theta), _abi.ptr(g), _abi.ptr(self.adam_m),
                                                     _abi.ptr(self.adam_v), w.P, w.off_c, w.off_ls, w.off_ls, cfg, self.adam_step,
                                                     self.adam_step, 0, w.off_ls, 0, 1, _abi.ptr(loss3), _abi.ptr(self.scal4),
                                                     _abi.ptr(part), cap, _abi.stream_ptr()), "spo_wide_clip_adam_ex")
        graphed = 0 < cfg.batch <= self.graph_max_batch and n_mb > 2         # launch-bound regime: one captured graph per step
        key = ("critic_fit", self._graph_cfg_key(cfg))
        for it in range(c["learning_iters"]):
            perm = _abi.require_gpu_tensor(perm_fn(it), "perm", torch.int32).long()
            losses = torch.empty((n_mb, 3), dtype=torch.float32, device=self.dev)
            n_full = self.M // cfg.batch if graphed else 0
            if graphed:
                self._sync_pow4()
                self._graphed_pass(key, perm, cfg.batch, n_full, losses, lambda i_, l_: step(i_, l_, True))
                self.adam_step += n_full
            for k in range(n_full, n_mb):
                step(perm[k * cfg.batch:(k + 1) * cfg.batch], losses[k], False)
                self.adam_step += 1
            all_losses.append(self._mean_over_ranks_(losses)[:, :2])
        # keep the norm the persistent kernel would carry in step with the rescaled vector
        self.stale_sq.copy_(g[self.ls_off:].dot(g[self.ls_off:]).reshape(1))
        means = torch.cat(all_losses, 0).mean(0).tolist() if all_losses else [float("nan")] * 2
        return {"loss_r": means[0], "loss_c": means[1], "losses": all_losses}


def make_engine(policy, n_local, local_steps_per_epoch, config, device, comm=None):
    """CPOEngine on the LDS-resident kernels when the policy fits them, WideCPOEngine otherwise (any dims / widths)."""
    cls = CPOEngine if policy.kernels_supported("cpo") else WideCPOEngine
    return cls(policy, n_local, local_steps_per_epoch, config, device, comm=comm)


def _to_dev(x, dev):
    return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32, device=dev).contiguous()


def main(args, cfg_env=None, _update="cpo"):
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if args.device == "cpu":
        raise RuntimeError("this build runs the CPO hot path on a ROCm GPU only (--device cuda); no CPU fallback")
    comm = init_from_env()
    local_rank = int(os.environ.get("LOCAL_RANK", args.device_id))
    device = torch.device(f"cuda:{local_rank if comm.world_size > 1 else args.device_id}")
    torch.cuda.set_device(device)
    if args.task in isaac_gym_map:
        raise NotImplementedError("Isaac Gym tasks (isaac_gym_specific_cfg) are not part of this build")
    config = dict(default_cfg)
    config.update(getattr(args, "cfg_override", None) or {})
    require_equal_shards(args.num_envs, comm)
    _, n_local = shard_envs(args.num_envs, comm)          # one process per GPU: a contiguous shard of the envs each
    env, obs_space, act_space = make_sa_mujoco_env(num_envs=n_local, env_id=args.task, seed=args.seed + 1000 * comm.rank,
                                                   device=device, **(getattr(args, "env_kwargs", None) or {}))
    device_env = getattr(env, "is_device_env", False)
    steps_per_epoch = config.get("steps_per_epoch", args.steps_per_epoch)
    total_steps = config.get("total_steps", args.total_steps)
    local_steps_per_epoch = steps_per_epoch // args.num_envs
    epochs = total_steps // steps_per_epoch
    policy = ActorVCritic(obs_dim=obs_space.shape[0], act_dim=act_space.shape[0],
                          hidden_sizes=config["hidden_sizes"]).to(device)
    comm.broadcast_(policy.theta, 0)                       # identical replicas
    engine = make_engine(policy, n_local, local_steps_per_epoch, config, device, comm=comm)
    dict_args = dict(vars(args))
    dict_args.update(config)
    is_root = comm.rank == 0
    logger = EpochLogger(log_dir=args.log_dir if is_root else os.path.join(args.log_dir, f"rank{comm.rank}"),
                         seed=str(args.seed), verbose=is_root)
    logger.save_config(dict_args)
    logger.setup_torch_saver(policy.actor)
    logger.log("Start with training.")
    # device envs that normalise observations hand the normaliser over: statistics merge + normalisation then run inside
    # the collect kernel (spo_policy_step_norm), one pass over the raw observations
    rms = env.fuse_normalize(True) if hasattr(env, "fuse_normalize") else None
    obs, _ = env.reset()
    obs = _to_dev(obs, device)
    timings = []
    for epoch in range(epochs):
        rollout_start_time = time.time()
        for steps in range(local_steps_per_epoch):
            act = engine.collect_step(steps, obs, rms=rms)
            action = act if device_env else act.detach().squeeze().cpu().numpy()
            next_obs, reward, cost, terminated, truncated, info = env.step(action)
            final_obs = None
            if "final_observation" in info:
                fo = info["final_observation"]
                if not torch.is_tensor(fo):
                    fo = np.array([a if a is not None else np.zeros(obs.shape[-1]) for a in fo])
                final_obs = _to_dev(fo, device)
            next_obs = _to_dev(next_obs, device)
            engine.post_step(steps, next_obs, _to_dev(reward, device), _to_dev(cost, device),
                             _to_dev(terminated, device), _to_dev(truncated, device), final_obs, rms=rms)
            obs = next_obs
        engine.drain_episode_events(logger)
        torch.cuda.synchronize(device)
        rollout_end_time = time.time()
        eval_end_time = rollout_end_time

        # ---- update policy (cpo.py:350-532) and critics (:534-571)
        engine.buffer.compute_gae(None, comm)
        ep_costs = dp_epoch_stat(comm, logger, "Metrics/EpCost", device) - args.cost_limit
        out = engine.policy_update(ep_costs, logger) if _update == "cpo" else engine.pcpo_update(ep_costs, logger)
        logger.store(**{"Misc/Alpha": out["alpha"], "Misc/FinalStepNorm": out["final_step_norm"], "Misc/xHx": out["xHx"],
                        "Misc/gradient_norm": out["gradient_norm"], "Misc/H_inv_g": out["H_inv_g"],
                        "Misc/AcceptanceStep": out["acceptance_step"], "Loss/Loss_actor": out["loss_actor"],
                        "Train/KL": out["kl"]})
        fit = engine.critic_fit()
        engine.buffer.reset()
        logger.store(**{"Loss/Loss_reward_critic": fit["loss_r"], "Loss/Loss_cost_critic": fit["loss_c"]})
        torch.cuda.synchronize(device)
        update_end_time = time.time()
        timings.append((rollout_end_time - rollout_start_time, update_end_time - eval_end_time))
        if not logger.logged:
            logger.log_tabular("Metrics/EpRet")
            logger.log_tabular("Metrics/EpCost")
            logger.log_tabular("Metrics/EpLen")
            logger.log_tabular("Train/Epoch", epoch + 1)
            logger.log_tabular("Train/TotalSteps", (epoch + 1) * args.steps_per_epoch)
            logger.log_tabular("Train/KL")
            logger.log_tabular("Loss/Loss_reward_critic")
            logger.log_tabular("Loss/Loss_cost_critic")
            logger.log_tabular("Loss/Loss_actor")
            logger.log_tabular("Time/Rollout", rollout_end_time - rollout_start_time)
            logger.log_tabular("Time/Update", update_end_time - eval_end_time)
            logger.log_tabular("Time/Total", update_end_time - rollout_start_time)
            d = engine.buffer.data
            logger.log_tabular("Value/RewardAdv", d["adv_r"].mean().item())
            logger.log_tabular("Value/CostAdv", d["adv_c"].mean().item())
            logger.log_tabular("Misc/Alpha")
            logger.log_tabular("Misc/FinalStepNorm")
            logger.log_tabular("Misc/xHx")
            logger.log_tabular("Misc/gradient_norm")
            logger.log_tabular("Misc/H_inv_g")
            logger.log_tabular("Misc/AcceptanceStep")
            logger.dump_tabular()
            if (epoch + 1) % 100 == 0 or epoch == 0:
                logger.torch_save(itr=epoch)
                logger.save_state(state_dict={"Normalizer": getattr(env, "obs_rms", None)}, itr=epoch)
        else:
            for k in list(logger.epoch_dict):
                logger.epoch_dict[k] = []
    logger.close()
    return {"timings": timings, "policy": policy, "engine": engine}


if __name__ == "__main__":
    run_as_script(main, __file__)
