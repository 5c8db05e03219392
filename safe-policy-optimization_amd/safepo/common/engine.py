"""Device-resident engine for the single-agent hot path: collect -> GAE -> PPO-Lagrangian update.

Host-side orchestration of the HIP kernels behind the C ABI (include/safepo_hip.h).  It replaces
the three hot loops of the reference main():
  loop 1 (per-step Python loops over num_envs in buffer.store and the done-scan,
          ppo_lag.py:162-234)         -> spo_policy_step + spo_values + spo_boundary_step
  loop 2 (per-path Python scalar GAE, buffer.py:182-188) -> spo_gae_fused (+ adv statistics)
  loop 3 (learning_iters x N*T/64 torch minibatch steps, ppo_lag.py:297-336)
                                      -> spo_ppo_lag_update_iter (persistent kernel) and
                                         spo_actor_kl for the early-stop test (ppo_lag.py:338-348)
No step of the rollout synchronises with the host: episode statistics are appended to a device
event log and replayed on the host once per epoch, in the reference's (step, env) order.
"""
from __future__ import annotations

import os
from collections import deque

import numpy as np
import torch

from safepo import _abi
from safepo.common.buffer import VectorizedOnPolicyBuffer
from safepo.common.model import ActorVCritic
from safepo.common.wide import PermWindow
from safepo.parallel import Comm, dp_reduce_gradient_


def _to_dev(x, dev):
    return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32, device=dev).contiguous()


def deque_running_means(dq: deque, new: np.ndarray, want_means: bool = True) -> list:
    """for v in new: dq.append(v); out.append(np.mean(dq)) -- without the Python loop.  A full window's mean is numpy's mean
    of the same 50 contiguous float64 values (a row of a sliding-window view: the same pairwise routine on the same run of
    values), so the results are those of the loop bit for bit; the windows that are still filling up go through the loop."""
    W = dq.maxlen
    new = np.asarray(new, dtype=np.float64)
    out = []
    if want_means and new.size:
        seq = np.concatenate([np.asarray(dq, dtype=np.float64), new])
        p = len(dq)
        k0 = min(max(W - 1 - p, 0), new.size)            # new[k] with k >= k0 sees a full window
        out = [np.mean(seq[:p + k + 1]) for k in range(k0)]
        if k0 < new.size:
            sw = np.lib.stride_tricks.sliding_window_view(seq, W)
            out.extend(np.mean(sw[p + k0 - (W - 1):p + new.size - (W - 1)], axis=1))
    dq.extend(new[-W:] if new.size > W else new)
    return out


class _Space:
    def __init__(self, dim):
        self.shape = (int(dim),)


class PPOLagEngine:
    """Owns the dense buffer, optimiser state and scratch for one GPU (one env shard)."""

    def __init__(self, policy: ActorVCritic, num_envs: int, steps: int, config: dict, device,
                 comm: Comm | None = None, lr: float = 3e-4, critic_lr: float | None = None):
        self._require_policy(policy)
        self.policy, self.N, self.T = policy, int(num_envs), int(steps)
        self.D, self.A = policy.obs_dim, policy.act_dim
        self.cfg = config
        self.dev = torch.device(device)
        self.comm = comm or Comm()
        self.lib = _abi.load()
        N, T, D, A = self.N, self.T, self.D, self.A
        self.buffer = VectorizedOnPolicyBuffer(_Space(D), _Space(A), size=T, num_envs=N, device=self.dev,
                                               gamma=config["gamma"], lam=config.get("lam", 0.95),
                                               lam_c=config.get("lam_c", 0.95))
        f32 = dict(dtype=torch.float32, device=self.dev)
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.act_out = torch.empty((N, A), **f32)
        self.logp = torch.empty(N, **f32)
        self.v_r, self.v_c = torch.empty(N, **f32), torch.empty(N, **f32)
        self.vnext_r, self.vnext_c = torch.zeros(N, **f32), torch.zeros(N, **f32)
        self.vfinal_r, self.vfinal_c = torch.zeros(N, **f32), torch.zeros(N, **f32)
        self.ep_ret, self.ep_cost, self.ep_len = torch.zeros(N, **f64), torch.zeros(N, **f64), torch.zeros(N, **f64)
        self.events_cap = N * T
        self.events = torch.zeros((self.events_cap, 4), **f64)
        # running count of logged episodes per step (spo_boundary_step_fold_mb): the kernel of step t reads [t], writes [t + 1]
        self.events_prefix = torch.zeros(T + 1, dtype=torch.int32, device=self.dev)
        self._events_last_t, self._events_drained = -1, 0
        self._events_pending = False              # steps were taken since the last drain_episode_events()
        self._rollout_graphs = {}                 # rollout_epoch: captured epochs by (env, observation tensor, normaliser)
        # optimiser state (flat, same order as policy.theta)
        P = policy.theta.numel()
        self.adam_m, self.adam_v = torch.zeros(P, **f32), torch.zeros(P, **f32)
        self.adam_step = 0
        self.adam_step_actor_extra = 0        # actor-only optimiser steps (CUP's second stage)
        self.std_old = torch.empty(A, **f32)
        self.lr_actor0, self.lr_critic = lr, (lr if critic_lr is None else critic_lr)
        self.lr_factor = 1.0
        self.M = N * T
        self.mean_old = torch.empty((self.M, A), **f32)
        self.logstd_old = torch.empty(A, **f32)
        self.kl_partials = torch.zeros(1024, **f64)
        self.kl_sum = torch.zeros(1, **f64)
        self.sync_ws = torch.zeros(32, dtype=torch.int64, device=self.dev)
        self.flat_grad = torch.zeros(P, **f32)
        self.losses3 = torch.zeros(3, **f32)
        self._losses = None
        # data-parallel: in-kernel exchange over peer-mapped regions when available (every rank must hold the same
        # number of rows), else the RCCL form
        self.p2p = None
        if self.comm.world_size > 1:
            from safepo.parallel import PeerExchange
            rows = torch.tensor([float(self.M), -float(self.M)], device=self.dev)
            self.comm.all_reduce_max_(rows)
            if rows[0].item() == -rows[1].item():
                self.p2p = PeerExchange.try_create(self.comm, self.dev)
                self._autotune_exchange()
        self.rew_deque, self.cost_deque, self.len_deque = deque(maxlen=50), deque(maxlen=50), deque(maxlen=50)

    FUSED_POST_STEP = True      # post_step: spo_values_boundary_step_fold (the wide-network engines take the two-launch form)

    def _autotune_exchange(self) -> None:
        """Which form of the per-minibatch exchange this job uses is MEASURED on the topology it runs on (PeerExchange.autotune)
        unless the environment names one (SPO_P2P_ALGO / SPO_P2P_A2A / SPO_P2P_HELPER) or SPO_P2P_AUTOTUNE=0; if the kernel /
        RCCL / kernel form is the fastest, the peer regions are released and that form runs."""
        px = self.p2p
        if px is None or os.environ.get("SPO_P2P_AUTOTUNE", "1") == "0":
            return
        if any(os.environ.get(k) not in (None, "", "0") for k in ("SPO_P2P_ALGO", "SPO_P2P_A2A", "SPO_P2P_HELPER")):
            return
        if not self.policy.kernels_supported("ppo"):
            return
        self.exchange_autotune = px.autotune(self.D, self.A)
        if px.prefer_rccl:
            px.close()
            self.p2p = None

    def _require_policy(self, policy) -> None:
        policy._require_kernels()

    def _values_into(self, obs: torch.Tensor, out_r: torch.Tensor, out_c: torch.Tensor) -> None:
        """(v_r, v_c) of both critics for bootstrap values (ppo_lag.py:201-215)."""
        _abi.check(self.lib.spo_values(_abi.ptr(self.policy.theta), _abi.ptr(obs), _abi.ptr(out_r), _abi.ptr(out_c), self.N,
                                       self.D, self.A, _abi.stream_ptr()), "spo_values")

    # ------------------------------------------------------------------ collect
    def collect_step(self, t: int, obs: torch.Tensor, eps: torch.Tensor | None = None,
                     deterministic: bool = False, rms=None) -> torch.Tensor:
        """policy.step(obs) + buffer.store(obs, act, value_r, value_c, log_prob) for step t
        (ppo_lag.py:163-164,187-195) in one kernel.  Returns the sampled action [N, A] (device).
        `rms`: a DeviceObsNormalizer in fused mode (env.fuse_normalize()): when its `pending` flag says `obs` is still raw,
        the statistics merge and the normalisation (wrappers.py:42-49) run inside this step (spo_policy_step_norm) and
        `obs` is normalised in place."""
        b = self.buffer
        assert t == b.ptr and t < self.T, "Buffer overflow"
        obs = _abi.require_gpu_tensor(obs, "obs", torch.float32)
        if not deterministic and eps is None:
            eps = torch.randn((self.N, self.A), device=self.dev, dtype=torch.float32)
        d = b.data
        if rms is not None and rms.pending:
            rms.pending = False
            _abi.check(self.lib.spo_policy_step_norm(
                _abi.ptr(self.policy.theta), _abi.ptr(obs), _abi.ptr(rms.state), int(rms.update_enabled), _abi.ptr(eps),
                _abi.ptr(self.act_out), _abi.ptr(self.logp), _abi.ptr(self.v_r), _abi.ptr(self.v_c), _abi.ptr(d["obs"]),
                _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]), _abi.ptr(d["value_r"]), _abi.ptr(d["value_c"]), self.N, self.T,
                t, self.D, self.A, _abi.stream_ptr()), "spo_policy_step_norm")
            return self.act_out
        _abi.check(self.lib.spo_policy_step(
            _abi.ptr(self.policy.theta), _abi.ptr(obs), _abi.ptr(eps), _abi.ptr(self.act_out), _abi.ptr(self.logp),
            _abi.ptr(self.v_r), _abi.ptr(self.v_c), _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]),
            _abi.ptr(d["value_r"]), _abi.ptr(d["value_c"]), self.N, self.T, t, self.D, self.A, _abi.stream_ptr()),
            "spo_policy_step")
        return self.act_out

    def post_step(self, t: int, next_obs, reward, cost, terminated, truncated, final_obs=None, rms=None) -> None:
        """Everything after env.step for step t (ppo_lag.py:168-234): reward/cost store, episode
        accumulators, boundary flags, bootstrap values, finish_path marks.
        `rms` (fused normaliser, see collect_step): at the epoch end the bootstrap values are taken from `next_obs`, which the
        reference's wrapper has already normalised -- a still-raw `next_obs` is merged and normalised here, once."""
        b, st = self.buffer, _abi.stream_ptr()
        epoch_end = t >= self.T - 1
        if epoch_end and rms is not None and rms.pending:
            rms.pending = False
            rms.normalize_(_abi.require_gpu_tensor(next_obs, "next_obs", torch.float32), update=True)
        tens = [_abi.require_gpu_tensor(x, n, torch.float32) for x, n in
                ((reward, "reward"), (cost, "cost"), (terminated, "terminated"), (truncated, "truncated"))]
        if epoch_end:
            next_obs = _abi.require_gpu_tensor(next_obs, "next_obs", torch.float32)
            self._values_into(next_obs, self.vnext_r, self.vnext_c)
        d = b.data
        if t == 0:
            if self._events_pending:
                # the log restarts at events_prefix[0] == 0 with every epoch: episodes of the previous epoch that were never
                # drained are about to be overwritten (every loop in this tree drains once per epoch)
                import warnings
                warnings.warn("PPOLagEngine.post_step: the previous epoch's episode log was not drained "
                              "(drain_episode_events() must be called once per epoch); its episodes are dropped", RuntimeWarning)
            self._events_drained = 0                  # a new epoch: the log restarts at events_prefix[0] == 0
        self._events_pending = True
        self._events_last_t = t
        if final_obs is not None:
            final_obs = _abi.require_gpu_tensor(final_obs, "final_observation", torch.float32)
            if self.FUSED_POST_STEP:
                # critics on the final observations + the boundary logic that consumes their values: one launch
                _abi.check(self.lib.spo_values_boundary_step_fold(
                    _abi.ptr(self.policy.theta), _abi.ptr(final_obs), _abi.ptr(self.vfinal_r), _abi.ptr(self.vfinal_c), self.D, self.A,
                    _abi.ptr(tens[0]), _abi.ptr(tens[1]), _abi.ptr(tens[2]), _abi.ptr(tens[3]),
                    _abi.ptr(self.vnext_r), _abi.ptr(self.vnext_c),
                    _abi.ptr(d["reward"]), _abi.ptr(d["cost"]), _abi.ptr(b.seg_end), _abi.ptr(b.boot_r), _abi.ptr(b.boot_c),
                    _abi.ptr(self.ep_ret), _abi.ptr(self.ep_cost), _abi.ptr(self.ep_len), _abi.ptr(self.events),
                    _abi.ptr(self.events_prefix), self.events_cap, self.N, self.T, t, int(epoch_end),
                    _abi.ptr(b.reward_fold), _abi.ptr(b.cost_fold), float(b._gamma), st), "spo_values_boundary_step_fold")
                if b._fold_cols == t:
                    b._fold_cols = t + 1
                b.advance()
                return
            self._values_into(final_obs, self.vfinal_r, self.vfinal_c)
        _abi.check(self.lib.spo_boundary_step_fold_mb(
            _abi.ptr(tens[0]), _abi.ptr(tens[1]), _abi.ptr(tens[2]), _abi.ptr(tens[3]),
            _abi.ptr(self.vnext_r), _abi.ptr(self.vnext_c), _abi.ptr(self.vfinal_r), _abi.ptr(self.vfinal_c),
            _abi.ptr(d["reward"]), _abi.ptr(d["cost"]), _abi.ptr(b.seg_end), _abi.ptr(b.boot_r), _abi.ptr(b.boot_c),
            _abi.ptr(self.ep_ret), _abi.ptr(self.ep_cost), _abi.ptr(self.ep_len), _abi.ptr(self.events),
            _abi.ptr(self.events_prefix), self.events_cap, self.N, self.T, t, int(epoch_end),
            _abi.ptr(b.reward_fold), _abi.ptr(b.cost_fold), float(b._gamma), st), "spo_boundary_step_fold_mb")
        if b._fold_cols == t:
            b._fold_cols = t + 1          # column t of this epoch carries its folded bootstrap
        b.advance()

    # ------------------------------------------------------------------ one epoch of collect steps
    def _rollout_step(self, t: int, env, obs, rms, eps=None):
        """collect_step -> env.step -> post_step for step t (ppo_lag.py:162-234).  Host envs: numpy in, numpy out."""
        act = self.collect_step(t, obs, eps=eps, rms=rms)
        device_env = getattr(env, "is_device_env", False)
        next_obs, reward, cost, terminated, truncated, info = env.step(act if device_env else act.detach().squeeze().cpu().numpy())
        final_obs = None
        if "final_observation" in info:
            fo = info["final_observation"]
            if not torch.is_tensor(fo):
                fo = np.array([a if a is not None else np.zeros(self.D) for a in fo])
            final_obs = _to_dev(fo, self.dev)
        next_obs = _to_dev(next_obs, self.dev)
        self.post_step(t, next_obs, _to_dev(reward, self.dev), _to_dev(cost, self.dev), _to_dev(terminated, self.dev),
                       _to_dev(truncated, self.dev), final_obs, rms=rms)
        return next_obs

    def rollout_epoch(self, env, obs, rms=None):
        """The T collect steps of one epoch; returns the observation the next epoch starts from.  (Episode statistics:
        drain_episode_events afterwards.)
        A device env that declares `graph_safe` (its step is a fixed launch sequence on fixed tensors, its step counter lives on
        the device: SynthDeviceEnv) has the WHOLE epoch -- the noise draw, then per step: policy step, env step, bootstrap values,
        boundary/fold; normaliser merges where the loop has them -- captured into one HIP graph and replayed: the eager loop is
        bound by ~9 launches x ~12 us of host work per step, the replay by the device.  The step index and the epoch-end flag
        are baked into the captured kernel arguments; every array the kernels touch (parameters, buffer, normaliser state,
        event log, env tensors) is updated in place between epochs, so a replay sees the current contents.  The captured
        sequence is the steady-state one (the epoch starts from an observation the previous epoch's last step normalised); an
        epoch that starts from a raw observation (the first after reset()) runs eagerly, and the first epoch of an engine
        captures the graph on its way out, so that the capture falls into a warm-up epoch.  SPO_ROLLOUT_GRAPH=0 keeps the eager loop.  Host envs (numpy, PCIe
        every step) always run it."""
        T, b = self.T, self.buffer
        steady = not (rms is not None and rms.pending) and b._fold_cols == 0
        use_graph = getattr(env, "graph_safe", False) and os.environ.get("SPO_ROLLOUT_GRAPH", "1") != "0"
        if use_graph:
            obs = _abi.require_gpu_tensor(obs, "obs", torch.float32)
            # everything the captured launches bake in: the tensors (env, observation, normaliser) and the env / normaliser
            # settings that are kernel ARGUMENTS (a reset(seed=...), a changed p_term or a frozen normaliser gets its own graph);
            # the cache entry holds env and rms, so their ids cannot be recycled while the graph lives
            key = (id(env), obs.data_ptr(), id(rms), getattr(env, "seed", None), getattr(env, "p_term", None),
                   getattr(env, "p_cost", None), getattr(env, "trunc_len", None), None if rms is None else bool(rms.update_enabled))
            graphs = self._rollout_graphs
        if not (use_graph and steady and key in graphs):
            # (also the first steady-state epoch runs eagerly: kernels with lazy set-up must have run once before a capture)
            eps_all = self._epoch_noise()
            for t in range(T):
                obs = self._rollout_step(t, env, obs, rms, eps_all[t])
            if use_graph and key not in graphs and torch.is_tensor(obs) and obs.data_ptr() == key[1]:
                if len(graphs) >= 4:                   # (settings that keep changing: do not pile up graphs and their pools)
                    graphs.clear()
                graphs[key] = self._capture_rollout(env, obs, rms) + ((env, rms),)
            return obs
        assert b.ptr == 0, "rollout_epoch starts on an empty buffer"
        g, post = graphs[key][:2]
        env.begin_epoch_base()                     # device step base := the epoch's first step (an ordinary launch before the replay)
        g.replay()
        # what T eager steps leave behind on the host side
        b.ptr = T
        b.ptr_list = [T] * b.num_envs
        b._fold_cols = post["fold_cols"]
        if rms is not None:
            rms.pending = post["pending"]
        env.step_count += T
        if self._events_pending:
            import warnings
            warnings.warn("PPOLagEngine.rollout_epoch: the previous epoch's episode log was not drained; its episodes are dropped",
                          RuntimeWarning)
        self._events_last_t, self._events_drained, self._events_pending = T - 1, 0, True
        return post["obs"]

    def _epoch_noise(self) -> torch.Tensor:
        """The standard normals of the epoch's T policy steps in one draw ([T, N, A]; the reference draws them step by step
        from the host generator, model.py:163 rsample) -- one generator launch per epoch instead of one per step."""
        return torch.randn((self.T, self.N, self.A), device=self.dev, dtype=torch.float32)

    def _capture_rollout(self, env, obs, rms):
        """Capture T steady-state steps starting from `obs`.  Nothing executes during a capture, but the host-side bookkeeping
        of the steps does: it runs on a scratch copy of that state (empty buffer, normalised observation, step offsets 1..T)
        which is put back afterwards."""
        b = self.buffer
        saved = (b.ptr, b.ptr_list, b._fold_cols, None if rms is None else rms.pending, env.step_count,
                 self._events_last_t, self._events_drained, self._events_pending)
        b.ptr, b.ptr_list, b._fold_cols = 0, [0] * b.num_envs, 0
        self._events_pending = False
        if rms is not None:
            rms.pending = False
        env.begin_epoch_base()
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        try:
            # (thread_local: another thread of the process -- the RCCL watchdog of a data-parallel run polling its events -- must
            #  not invalidate the capture; this thread only launches kernels inside it)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                cur = obs
                eps_all = self._epoch_noise()
                for t in range(self.T):
                    cur = self._rollout_step(t, env, cur, rms, eps_all[t])
            post = {"fold_cols": b._fold_cols, "pending": None if rms is None else rms.pending, "obs": cur}
            wide = getattr(self, "wide", None)
            if wide is not None:                       # workspaces created by the eager epoch and addressed by the captured kernels
                post["keep"] = list(wide._ws.values()) + list(wide._scratch.values())
        finally:
            b.ptr, b.ptr_list, b._fold_cols = saved[0], saved[1], saved[2]
            if rms is not None:
                rms.pending = saved[3]
            env.step_count = saved[4]
            self._events_last_t, self._events_drained, self._events_pending = saved[5], saved[6], saved[7]
        return g, post

    def drain_episode_events(self, logger=None):
        """Replay finished episodes on the host in the reference's (step, env) order
        (ppo_lag.py:216-230: deques of 50, running means stored per finished episode).
        One device->host copy per epoch.  Returns the number of finished episodes."""
        if self._events_last_t < 0:
            return 0
        self._events_pending = False
        hi = int(self.events_prefix[self._events_last_t + 1].item())
        if hi > self.events_cap:
            raise _abi.SpoError(f"episode event log overflow ({hi} > {self.events_cap})")
        lo, self._events_drained = self._events_drained, hi
        n = hi - lo
        if not n:
            return 0
        ev = self.events[lo:hi].cpu().numpy()
        # the reference appends each finished episode to deques of 50 and stores the deques' means per episode
        # (ppo_lag.py:216-230); here per column in one pass (8 192 episodes per epoch at 4 096 envs: a Python loop with three
        # np.mean calls per episode was ~0.1 s per epoch, more than the device side of the rollout)
        for dq, col, key in ((self.rew_deque, 1, "Metrics/EpRet"), (self.cost_deque, 2, "Metrics/EpCost"),
                             (self.len_deque, 3, "Metrics/EpLen")):
            means = deque_running_means(dq, ev[:, col], want_means=logger is not None)
            if logger is not None and hasattr(logger, "epoch_dict"):
                logger.epoch_dict.setdefault(key, []).extend(means)          # == logger.store(**{key: m}) for m in means
            elif logger is not None:
                for m in means:
                    logger.store(**{key: m})
        if logger is not None:
            logger.logged = False
        return n

    # ------------------------------------------------------------------ update
    def _cfg_struct(self) -> _abi.PpoCfg:
        c = self.cfg
        M = self.M
        batch = c.get("batch_size", max(M // c.get("num_mini_batch", 1), 1))
        # data-parallel batch semantics (SURVEY.md 8(e) "Partitioning"): "local" (default) = every rank takes batch_size rows
        # of its shard per step (global batch = batch_size x world: weak scaling, a different optimisation trajectory);
        # "global" = batch_size is the GLOBAL minibatch and every rank takes batch_size / world rows -- the reference's
        # arithmetic up to the order of the sums (cfg key dp_batch or SPO_DP_BATCH)
        if self.comm.world_size > 1 and str(c.get("dp_batch", os.environ.get("SPO_DP_BATCH", "local"))) == "global":
            if batch % self.comm.world_size:
                raise _abi.SpoError(f"dp_batch=global needs batch_size ({batch}) divisible by the world size "
                                    f"({self.comm.world_size})")
            batch //= self.comm.world_size
        return _abi.PpoCfg(obs_dim=self.D, act_dim=self.A, batch=int(batch),
                           use_critic_norm=int(c.get("use_critic_norm", True)),
                           use_value_coefficient=int(c.get("use_value_coefficient", False)),
                           clip=float(c.get("clip", 0.2)), max_grad_norm=float(c["max_grad_norm"]),
                           lr_actor=float(self.lr_actor0 * self.lr_factor), lr_critic=float(self.lr_critic),
                           beta1=0.9, beta2=0.999, adam_eps=1e-8, l2_coef=0.001)

    def snapshot_old_distribution(self) -> None:
        """old_distribution = policy.actor(data["obs"]) (ppo_lag.py:277)."""
        obs = self.buffer.data["obs"]
        _abi.check(self.lib.spo_actor_mean(_abi.ptr(self.policy.theta), _abi.ptr(obs), _abi.ptr(self.mean_old),
                                           self.M, self.D, self.A, _abi.stream_ptr()), "spo_actor_mean")
        off = self.policy.log_std_offset
        self.logstd_old.copy_(self.policy.theta[off:off + self.A])
        torch.exp(self.logstd_old, out=self.std_old)      # Normal.stddev of the snapshot (focops.py:283, cup.py:358)

    def kl_launch(self) -> None:
        """Enqueue the full-batch KL(old || new) of the early-stop test (ppo_lag.py:338-345); kl_read() returns it."""
        obs = self.buffer.data["obs"]
        _abi.check(self.lib.spo_actor_kl(_abi.ptr(self.policy.theta), _abi.ptr(obs), _abi.ptr(self.mean_old),
                                         _abi.ptr(self.logstd_old), _abi.ptr(self.kl_partials),
                                         self.kl_partials.numel(), _abi.ptr(self.kl_sum), self.M, self.D, self.A,
                                         _abi.stream_ptr()), "spo_actor_kl")
        self.comm.all_reduce_sum_(self.kl_sum)

    def kl_read(self) -> float:
        return float(self.kl_sum.item()) / float(self.M * self.comm.world_size)

    def kl_to_old(self) -> float:
        """KL(old || new).sum(-1).mean() over the (global) batch (ppo_lag.py:338-345)."""
        self.kl_launch()
        return self.kl_read()

    def learning_iter(self, perm: torch.Tensor) -> torch.Tensor:
        """All minibatches of one pass over the data (ppo_lag.py:298-336).  `perm`: int32 device
        permutation of [0, M).  Returns per-minibatch losses [n_mb, 3] (device)."""
        cfg = self._cfg_struct()
        perm = _abi.require_gpu_tensor(perm, "perm", torch.int32)
        d, b = self.buffer.data, self.buffer
        M = self.M
        n_mb = (M + cfg.batch - 1) // cfg.batch
        losses = torch.empty((n_mb, 3), dtype=torch.float32, device=self.dev)
        st = _abi.stream_ptr()
        th = self.policy.theta
        import os
        if self.comm.world_size > 1 and self.p2p is not None:
            px = self.p2p
            _abi.check(self.lib.spo_ppo_lag_update_iter_dp(
                _abi.ptr(th), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), self.adam_step, _abi.ptr(d["obs"]),
                _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]), _abi.ptr(d["target_value_r"]),
                _abi.ptr(d["target_value_c"]), _abi.ptr(b.adv_mix), _abi.ptr(perm), M, cfg, _abi.ptr(losses),
                _abi.ptr(self.sync_ws), px.rank, px.world, px.regions, px.step & 0xFFFFFFFF, st),
                "spo_ppo_lag_update_iter_dp")
            px.step += n_mb
            self.adam_step += n_mb
            self.comm.all_reduce_sum_(losses)
            losses *= 1.0 / self.comm.world_size
        elif self.comm.world_size == 1 and os.environ.get("SPO_FORCE_DP", "0") != "1":
            _abi.check(self.lib.spo_ppo_lag_update_iter(
                _abi.ptr(th), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), self.adam_step, _abi.ptr(d["obs"]),
                _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]), _abi.ptr(d["target_value_r"]),
                _abi.ptr(d["target_value_c"]), _abi.ptr(b.adv_mix), _abi.ptr(perm), M, cfg, _abi.ptr(losses),
                _abi.ptr(self.sync_ws), st), "spo_ppo_lag_update_iter")
            self.adam_step += n_mb
        else:
            # data-parallel: local minibatch gradient -> all-reduce (RCCL) -> identical clip+Adam on every rank.
            # One C call per step: it applies step k and enqueues the gradient kernel of step k+1.
            lib, fg = self.lib, self.flat_grad
            p_th, p_m, p_v, p_fg = _abi.ptr(th), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), _abi.ptr(fg)
            p_obs, p_act, p_lp = _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"])
            p_tr, p_tc, p_adv = _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(b.adv_mix)
            p_perm, p_loss = perm.data_ptr(), losses.data_ptr()
            scale = 1.0 / self.comm.world_size
            n0 = min(cfg.batch, M)
            _abi.check(lib.spo_ppo_lag_grad(p_th, p_obs, p_act, p_lp, p_tr, p_tc, p_adv, p_perm, n0, n0, cfg, p_fg,
                                            p_loss, st), "spo_ppo_lag_grad")
            reduce_ = self.comm.all_reduce_sum_
            for k in range(n_mb):
                reduce_(fg)
                nxt = k + 1
                if nxt < n_mb:
                    lo = nxt * cfg.batch
                    rc = lib.spo_clip_adam_then_grad(p_th, p_m, p_v, p_fg, self.adam_step, scale, p_obs, p_act, p_lp,
                                                     p_tr, p_tc, p_adv, p_perm + 4 * lo, min(cfg.batch, M - lo), cfg,
                                                     p_loss + 12 * nxt, st)
                else:
                    rc = lib.spo_clip_adam_then_grad(p_th, p_m, p_v, p_fg, self.adam_step, scale, p_obs, p_act, p_lp,
                                                     p_tr, p_tc, p_adv, None, 0, cfg, None, st)
                if rc:
                    _abi.check(rc, "spo_clip_adam_then_grad")
                self.adam_step += 1
            self.comm.all_reduce_sum_(losses)
            losses *= 1.0 / self.comm.world_size
        return losses

    def learning_iter_ex(self, perm: torch.Tensor, adv: torch.Tensor, actor_loss: int = 0,
                         kl_bound: float = float("inf"), pg_coef: float = 0.0, actor_only: bool = False) -> torch.Tensor:
        """One pass over the data on the persistent kernel with the FOCOPS / CUP options of spo_update_iter_ex
        (include/safepo_hip.h): KL-penalty actor loss against the last snapshot_old_distribution(), actor-only
        optimisation, separate optimiser step counts.  Single GPU only."""
        if self.comm.world_size != 1:
            raise NotImplementedError("focops / cup run on one GPU in this build (no data-parallel form of the "
                                      "KL-penalty minibatch step)")
        cfg = self._cfg_struct()
        perm = _abi.require_gpu_tensor(perm, "perm", torch.int32)
        adv = _abi.require_gpu_tensor(adv, "adv", torch.float32)
        d = self.buffer.data
        M = self.M
        n_mb = (M + cfg.batch - 1) // cfg.batch
        losses = torch.full((n_mb, 3), float("nan"), dtype=torch.float32, device=self.dev)
        _abi.check(self.lib.spo_update_iter_ex(
            _abi.ptr(self.policy.theta), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), self.adam_step,
            self.adam_step + self.adam_step_actor_extra, _abi.ptr(d["obs"]), _abi.ptr(d["act"]),
            _abi.ptr(d["log_prob"]), _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(adv),
            _abi.ptr(perm), M, cfg, int(actor_loss), _abi.ptr(self.mean_old), _abi.ptr(self.std_old),
            float(kl_bound), float(pg_coef), int(actor_only), _abi.ptr(losses), _abi.ptr(self.sync_ws),
            _abi.stream_ptr()), "spo_update_iter_ex")
        if actor_only:
            self.adam_step_actor_extra += n_mb
        else:
            self.adam_step += n_mb
        return losses

    def _kl_stopped_loop(self, perm_fn, it0: int, run_iter):
        c = self.cfg
        all_losses, stop_iter, kl = [], 0, 1.0
        for it in range(c["learning_iters"]):
            all_losses.append(run_iter(perm_fn(it0 + it)))
            kl = self.kl_to_old()
            stop_iter += 1
            if kl > c["target_kl"]:
                break
        return all_losses, stop_iter, kl

    def update_focops(self, lagrangian_multiplier: float, perm_fn=None, focops_lam: float = 1.5):
        """FOCOPS epoch update (safepo/single_agent/focops.py:280-366): advantage (adv_r - nu*adv_c)/(nu+1), per-sample
        KL to the pre-update policy as a penalty, masked where it exceeds target_kl."""
        self.buffer.compute_gae(lagrangian_multiplier, self.comm)
        self.snapshot_old_distribution()
        if perm_fn is None:
            perm_fn = lambda it: torch.randperm(self.M, device=self.dev).to(torch.int32)
        adv = self.buffer.adv_mix
        losses, stop_iter, kl = self._kl_stopped_loop(perm_fn, 0, lambda perm: self.learning_iter_ex(
            perm, adv, _abi.ACTOR_LOSS_KL_PENALTY, self.cfg["target_kl"], 1.0 / focops_lam))
        self.check_sync_error()
        self.buffer.reset()
        means = torch.cat(losses, 0).mean(0).tolist()
        return {"stop_iter": stop_iter, "kl": kl, "loss_r": means[0], "loss_c": means[1], "loss_pi": means[2],
                "losses": losses}

    def update_cup(self, lagrangian_multiplier: float, perm_fn=None, cup_lambda: float = 0.95):
        """CUP epoch update (safepo/single_agent/cup.py:280-400): a PPO stage on adv_r, then an actor-only stage
        minimising  nu*coef*ratio*adv_c + KL(new || policy after stage one)."""
        self.buffer.compute_gae(0.0, self.comm)            # lambda 0: adv_mix == adv_r exactly (cup.py:285)
        self.snapshot_old_distribution()
        if perm_fn is None:
            perm_fn = lambda it: torch.randperm(self.M, device=self.dev).to(torch.int32)
        adv_r = self.buffer.adv_mix
        losses, stop_iter, kl = self._kl_stopped_loop(perm_fn, 0, lambda perm: self.learning_iter_ex(
            perm, adv_r, _abi.ACTOR_LOSS_CLIP))
        self.snapshot_old_distribution()                   # cup.py:355-358
        gamma = self.cfg["gamma"]
        coef = (1 - gamma * cup_lambda) / (1 - gamma)
        adv_c = self.buffer.data["adv_c"]
        losses2, stop_iter2, kl2 = self._kl_stopped_loop(perm_fn, stop_iter, lambda perm: self.learning_iter_ex(
            perm, adv_c, _abi.ACTOR_LOSS_KL_PENALTY, float("inf"), -float(lagrangian_multiplier) * coef, True))
        self.check_sync_error()
        self.buffer.reset()
        means = torch.cat(losses, 0).mean(0).tolist()
        return {"stop_iter": stop_iter, "second_stage_stop_iter": stop_iter2, "kl": kl2, "kl_first_stage": kl,
                "loss_r": means[0], "loss_c": means[1], "loss_pi": means[2], "losses": losses,
                "second_stage_losses": losses2}

    def check_sync_error(self):
        """Raise if an update kernel gave up on an exchange.  With the in-kernel data-parallel exchange the code is
        max-reduced over the ranks first, so every rank raises together (a rank that alone went on to the next
        collective would hang the job) and a caller can fall back to the RCCL form on all ranks at once."""
        if self.comm.world_size > 1 and self.p2p is not None:
            code_t = self.sync_ws[8:9].clone()
            self.comm.all_reduce_max_(code_t)
            code = int(code_t.item()) & 0xFFFFFFFF
        else:
            code = int(self.sync_ws[8].item()) & 0xFFFFFFFF
        if code:
            self.sync_ws[8] = 0           # sticky on the device (set by any launch since the last check): cleared here
        if code == 2:
            raise _abi.SpoError("update kernel: a peer rank never answered the in-kernel gradient exchange "
                                "(set SPO_P2P=0 to use the RCCL form)")
        if code:
            raise _abi.SpoError("update kernel: inter-workgroup exchange timed out (the workgroups of the persistent launch were not "
                                "co-resident; the row-split kernel leaves theta and the optimiser state as they were -- "
                                "SPO_UPDATE_FORM=2 selects the one-workgroup-per-network kernel)")

    def drop_peer_exchange(self):
        """Leave the in-kernel gradient exchange for the RCCL form of the minibatch step (after a peer timeout): close the
        regions, clear the error word and make the replicas identical again from rank 0 (parameters, Adam moments)."""
        if self.p2p is not None:
            self.p2p.close()
            self.p2p = None
        self.sync_ws.zero_()
        for t in (self.policy.theta, self.adam_m, self.adam_v):
            self.comm.broadcast_(t, 0)

    def update(self, lagrangian_multiplier: float, perm_fn=None):
        """GAE + statistics + mix, then the PPO-Lag update with KL early stopping
        (ppo_lag.py:275-349).  Returns dict(stop_iter, kl, loss means)."""
        c = self.cfg
        self.buffer.compute_gae(lagrangian_multiplier, self.comm)
        self.snapshot_old_distribution()
        if perm_fn is None:
            perm_fn = lambda it: torch.randperm(self.M, device=self.dev).to(torch.int32)
        all_losses = []
        stop_iter, kl = 0, 1.0
        # The host reads the KL back after every pass (the early-stop decision, ppo_lag.py:346-348).  The next pass's shuffle
        # is enqueued BEFORE that read, so the device generates it while the host waits and the next persistent launch
        # follows the KL kernels without a gap for the shuffle's own launches.
        n_it = c["learning_iters"]
        perm = perm_fn(0) if n_it > 0 else None
        for it in range(n_it):
            all_losses.append(self.learning_iter(perm))
            self.kl_launch()
            perm = perm_fn(it + 1) if it + 1 < n_it else None
            kl = self.kl_read()
            stop_iter += 1
            if kl > c["target_kl"]:
                break
        self.check_sync_error()
        self.buffer.reset()
        if all_losses:
            means = torch.cat(all_losses, 0).mean(0).tolist()
        else:
            means = [float("nan")] * 3
        return {"stop_iter": stop_iter, "kl": kl, "loss_r": means[0], "loss_c": means[1], "loss_pi": means[2],
                "losses": all_losses}


class _WideOps:
    """Engine pieces shared by the wide-network engines (WidePPOLagEngine here, WideCPOEngine in single_agent/cpo.py): the
    policy step, bootstrap values, the old-distribution snapshot and the full-batch KL on the wide-network kernels
    (safepo.common.wide), for any (obs_dim, act_dim <= 64, hidden_sizes).  Mixed in FRONT of PPOLagEngine / CPOEngine."""

    FAMILY = "ppo"
    FUSED_POST_STEP = False

    def _require_policy(self, policy) -> None:
        if policy.kernels_supported(self.FAMILY):
            raise ValueError(f"this shape runs on the persistent kernels (obs {policy.obs_dim}, act {policy.act_dim}, hidden "
                             f"{policy.hidden_sizes}): use the plain engine")
        if policy.act_dim > _abi.WIDE_MAX_ACT:
            raise _abi.SpoError(f"act_dim {policy.act_dim} outside [1, {_abi.WIDE_MAX_ACT}] (wide-network path)")
        policy.wide          # builds the layout; raises on unsupported depths

    def _wide_init(self) -> None:
        self.wide = self.policy.wide
        f32 = dict(dtype=torch.float32, device=self.dev)
        # spo_wide_ppo_loss: 256 x (3 + SPO_WIDE_MAX_ACT) + 512 + (2 + SPO_WIDE_MAX_ACT) doubles; spo_wide_clip_adam: <= 1024 x 3
        self.loss_partials = torch.zeros(256 * (3 + _abi.WIDE_MAX_ACT) + 1024 + 3 * 1024, dtype=torch.float64, device=self.dev)
        self.actor_sums = torch.zeros(2 + _abi.WIDE_MAX_ACT, dtype=torch.float64, device=self.dev)
        self.scal4 = torch.zeros(4, **f32)
        # device-resident optimiser clocks {beta1^t, beta2^t} of the critics' and the actor's optimisers (spo_wide_clip_adam_dev):
        # a minibatch step then has no host argument that changes between steps and is replayed from ONE captured HIP graph
        # ... and [4], [5] = the actor's / the critics' learning rate of this epoch (the LinearLR factor changes per epoch: baked into
        # the captured cfg it made every epoch capture a new graph)
        self.pow4 = torch.ones(6, dtype=torch.float64, device=self.dev)
        self._step_graphs = {}
        self.graph_max_batch = int(os.environ.get("SPO_WIDE_GRAPH_MAX_BATCH", "2048"))     # 0: never (every launch eager)
        if self.comm.world_size > 1:
            # data-parallel: the flat gradient of all three networks is all-reduced between the backward pass
            # and the joint clip (ppo_lag.py:325: clip_grad_norm_ on the reduced gradient), a collective per minibatch step --
            # steps launch eagerly (a host collective cannot sit inside a captured graph of this process)
            self.graph_max_batch = 0

    def _mean_over_ranks_(self, losses: torch.Tensor) -> torch.Tensor:
        """Per-step losses of the GLOBAL minibatches (the critics' L2 terms are the same on every rank)."""
        if self.comm.world_size > 1:
            self.comm.all_reduce_sum_(losses)
            losses.mul_(1.0 / self.comm.world_size)
        return losses

    def _reduce_flat_grad(self, lo: int = 0, hi: int | None = None) -> None:
        """Mean over the ranks of flat_grad[lo:hi] (world 1: nothing)."""
        if self.comm.world_size > 1:
            g = self.flat_grad[lo:hi]
            self.comm.all_reduce_sum_(g)
            g.mul_(1.0 / self.comm.world_size)

    # ------------------------------------------------------------------ graph-replayed minibatch steps
    def _sync_pow4(self) -> None:
        """Device clocks from the host step counts (start of a pass, after an eager step)."""
        b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))
        tc, ta = self.adam_step, self.adam_step + self.adam_step_actor_extra
        c = self._cfg_struct()                    # (c_float fields: the float32 values the host-clock kernels receive)
        self.pow4.copy_(torch.tensor([b1 ** tc, b2 ** tc, b1 ** ta, b2 ** ta, float(c.lr_actor), float(c.lr_critic)], dtype=torch.float64))

    @staticmethod
    def _graph_cfg_key(cfg) -> bytes:
        """The cfg struct's bytes without the learning rates (they live on the device, self.pow4[4:6])."""
        k = type(cfg).from_buffer_copy(bytes(cfg))
        k.lr_actor = 0.0
        k.lr_critic = 0.0
        return bytes(k)

    def _graphed_pass(self, key, perm: torch.Tensor, batch: int, n_full: int, losses: torch.Tensor, body) -> None:
        """The first `n_full` (whole) minibatches of the permutation `perm` as `n_full` replays of ONE captured HIP graph of
        `body(window, loss_static)` -- the launch sequence of a minibatch step with the optimiser clocks, the position in the
        permutation and the loss log on the device (safepo.common.wide.PermWindow, spo_gather_rows_at,
        spo_wide_clip_adam_dev_log): a replay takes no host copy in or out.  `key` carries everything baked into the captured
        kernel arguments (the cfg struct's bytes: learning rates change per epoch; the loss options; the batch).  First use: one
        eager run for the lazy per-kernel set-up (its effects on parameters / moments / clocks / cursor undone), then the
        capture.  losses[:n_full] receives the steps' losses."""
        from safepo.common.wide import PermWindow
        ent = self._step_graphs.get(key)
        if ent is None or ent[1].perm.numel() < perm.numel():
            if len(self._step_graphs) >= 6:
                self._step_graphs.clear()
            win = PermWindow(perm.numel(), batch, self.dev)
            loss_static = torch.full((3,), float("nan"), dtype=torch.float32, device=self.dev)
            win.load(perm)
            state = (self.policy.theta, self.adam_m, self.adam_v, self.pow4, self.flat_grad, win.cursor)
            snap = [t.clone() for t in state]
            body(win, loss_static)
            for t, b in zip(state, snap):
                t.copy_(b)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body(win, loss_static)
            # a second capture of `unroll` consecutive steps (round 6): a graph launch costs ~8 us of GPU idle time between the last
            # kernel of one replay and the first of the next -- a quarter of a 3-kernel step -- and nothing in a step's launches
            # depends on the host, so eight steps are one launch (SPO_WIDE_GRAPH_UNROLL=1: one step per launch, as rounds 4-5)
            unroll = max(1, int(os.environ.get("SPO_WIDE_GRAPH_UNROLL", "8")))
            g_many = None
            if unroll > 1:
                g_many = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_many, capture_error_mode="thread_local"):
                    for _ in range(unroll):
                        body(win, loss_static)
            # everything the captured kernels address stays alive with the graph: the static loss buffer, and the workspaces the
            # warm-up run created outside the capture (the caches of safepo.common.wide evict when they grow)
            keep = [loss_static, g_many, unroll] + list(self.wide._ws.values()) + list(self.wide._scratch.values())
            ent = self._step_graphs[key] = (g, win, keep)
        g, win, keep = ent
        g_many, unroll = keep[1], keep[2]
        win.load(perm)
        done = 0
        if g_many is not None:
            for _ in range(n_full // unroll):
                g_many.replay()
            done = (n_full // unroll) * unroll
        for _ in range(n_full - done):
            g.replay()
        losses[:n_full].copy_(win.loss_log[:n_full])

    def _clip_adam_dev(self, cfg, lo, hi, norm0, scale_rest, losses_out, log_src, window) -> None:
        """spo_wide_clip_adam_dev(_log): the joint clip + Adam with device-resident optimiser clocks; `window` (a PermWindow: the
        step is being captured / replayed) adds the loss log and the cursor advance."""
        w, part = self.wide, self.loss_partials
        _abi.check(self.lib.spo_wide_clip_adam_dev_log(
            _abi.ptr(self.policy.theta), _abi.ptr(self.flat_grad), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), w.P, w.off_c, w.off_ls,
            w.off_ls, cfg, _abi.ptr(self.pow4), lo, hi, norm0, scale_rest, None if losses_out is None else _abi.ptr(losses_out),
            _abi.ptr(self.scal4), _abi.ptr(part), part.numel(), None if window is None else _abi.ptr(window.loss_log),
            None if window is None else _abi.ptr(log_src), None if window is None else _abi.ptr(window.cursor),
            0 if window is None else window.n, _abi.stream_ptr()), "spo_wide_clip_adam_dev_log")

    def _values_into(self, obs, out_r, out_c) -> None:
        v_r, v_c = self.wide.values(obs)
        out_r.copy_(v_r); out_c.copy_(v_c)

    def collect_step(self, t: int, obs: torch.Tensor, eps: torch.Tensor | None = None,
                     deterministic: bool = False, rms=None) -> torch.Tensor:
        b = self.buffer
        assert t == b.ptr and t < self.T, "Buffer overflow"
        obs = _abi.require_gpu_tensor(obs, "obs", torch.float32)
        if rms is not None and rms.pending:
            rms.pending = False
            rms.normalize_(obs, update=True)
        if not deterministic and eps is None:
            eps = torch.randn((self.N, self.A), device=self.dev, dtype=torch.float32)
        act, logp, v_r, v_c = self.wide.step(obs, None if deterministic else eps)
        d = b.data
        d["obs"][:, t].copy_(obs); d["act"][:, t].copy_(act); d["log_prob"][:, t].copy_(logp)
        d["value_r"][:, t].copy_(v_r); d["value_c"][:, t].copy_(v_c)
        self.act_out.copy_(act); self.logp.copy_(logp); self.v_r.copy_(v_r); self.v_c.copy_(v_c)
        return self.act_out

    def snapshot_old_distribution(self) -> None:
        obs = self.buffer.data["obs"].view(self.M, self.D)
        self.wide.actor_mean(obs, out=self.mean_old)
        off = self.policy.log_std_offset
        self.logstd_old.copy_(self.policy.theta[off:off + self.A])
        torch.exp(self.logstd_old, out=self.std_old)

    def kl_launch(self) -> None:
        obs = self.buffer.data["obs"].view(self.M, self.D)
        off = self.policy.log_std_offset
        ls_new = self.policy.theta[off:off + self.A]
        from safepo.common.wide import KL_CHUNK
        for k, lo in enumerate(range(0, self.M, KL_CHUNK)):
            mu, _ = self.wide.forward("a", obs[lo:lo + KL_CHUNK])
            _abi.check(self.lib.spo_gauss_kl_sum(_abi.ptr(self.mean_old[lo:lo + mu.shape[0]]), _abi.ptr(self.logstd_old), _abi.ptr(mu),
                                                 _abi.ptr(ls_new), mu.shape[0], self.A, _abi.ptr(self.kl_partials),
                                                 self.kl_partials.numel(), _abi.ptr(self.kl_sum), int(k > 0), _abi.stream_ptr()),
                       "spo_gauss_kl_sum")
        self.comm.all_reduce_sum_(self.kl_sum)

    def kl_read(self) -> float:
        return float(self.kl_sum.item()) / float(self.M * self.comm.world_size)

    def check_sync_error(self):
        return None


class WidePPOLagEngine(_WideOps, PPOLagEngine):
    """The same epoch for an ActorVCritic outside the persistent kernels' envelope -- hidden_sizes other than [64, 64] (reference
    model.py:131; the isaac_gym_specific_cfg regime of ppo_lag.py:54-65), obs_dim > 128 or act_dim > 16 (HumanoidVelocity:
    376 / 17): collect, boundary logic, GAE, statistics and the KL early stop are shared with PPOLagEngine; the policy step, the
    bootstrap values, the full-batch actor evaluation and the minibatch step (clipped surrogate, and the KL-penalty loss of
    FOCOPS / CUP) run on the wide-network kernels (safepo.common.wide).  Data-parallel: env shards as PPOLagEngine, the flat
    gradient all-reduced (RCCL) per minibatch step before the joint clip."""

    def __init__(self, policy: ActorVCritic, num_envs: int, steps: int, config: dict, device,
                 comm: Comm | None = None, lr: float = 3e-4, critic_lr: float | None = None):
        super().__init__(policy, num_envs, steps, config, device, comm=comm, lr=lr, critic_lr=critic_lr)
        self._wide_init()
        if self.p2p is not None:                  # (the in-kernel exchange belongs to the persistent kernels: release its regions)
            self.p2p.close()
        self.p2p = None

    def _gather(self, idx, adv_all=None, extra=()):
        """The minibatch rows of obs, act, log_prob, both value targets, the advantage (adv_all; default the mixed one) and any
        `extra` [M, w] arrays: one launch (the reference's DataLoader batch, ppo_lag.py:298-305)."""
        d = self.buffer.data
        adv_all = self.buffer.adv_mix if adv_all is None else adv_all
        M = self.M
        out = self.wide.gather_rows(idx, [d["obs"].view(M, self.D), d["act"].view(M, self.A), d["log_prob"].view(M, 1),
                                          d["target_value_r"].view(M, 1), d["target_value_c"].view(M, 1), adv_all.view(M, 1)]
                                    + [e.view(M, -1) for e in extra])
        return [out[0], out[1], out[2].view(-1), out[3].view(-1), out[4].view(-1), out[5].view(-1)] + out[6:]

    def minibatch_step(self, idx: torch.Tensor, losses_out: torch.Tensor, dev_clock: bool = False, cfg=None) -> None:
        """ppo_lag.py:306-329 on the rows `idx` (int64 device indices into the flat buffer).  dev_clock: optimiser clocks from
        self.pow4 on the device (the graph-replayed form; the caller advances self.adam_step)."""
        w, lib, st = self.wide, self.lib, _abi.stream_ptr
        cfg = self._cfg_struct() if cfg is None else cfg
        g = self.flat_grad
        off_ls = w.off_ls
        if w.rows_grad_ok(idx.numel()):
            # round 6: gather + forward + loss + backward of the three networks in ONE launch split over the rows (csrc/mlp_rows.hip)
            d, M = self.buffer.data, self.M
            # one GPU, device-resident clocks (the replayed step): the group sum rides in the optimiser's first pass and the clip
            # coefficient in its second -- gradient launch + two optimiser launches (SPO_WIDE_ROWS_FUSED=0: sum, norm, coefficient
            # and Adam as the four launches of the other paths; same numbers)
            fused = dev_clock and self.comm.world_size == 1 and os.environ.get("SPO_WIDE_ROWS_FUSED", "1") != "0"
            parts = w.grad_rows(idx, d["obs"].view(M, self.D), d["act"].view(M, self.A), d["log_prob"].view(M),
                                d["target_value_r"].view(M), d["target_value_c"].view(M), self.buffer.adv_mix.view(M), float(cfg.clip),
                                g, losses_out, reduce=not fused)
            if fused:
                win = idx if isinstance(idx, PermWindow) else None
                part = self.loss_partials
                _abi.check(lib.spo_wide_rows_clip_adam_dev_log(
                    _abi.ptr(parts), idx.numel(), _abi.ptr(self.policy.theta), _abi.ptr(g), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v),
                    w.P, w.off_c, w.off_ls, w.off_ls, cfg, _abi.ptr(self.pow4), _abi.ptr(losses_out), _abi.ptr(self.scal4), _abi.ptr(part),
                    part.numel(), None if win is None else _abi.ptr(win.loss_log), None if win is None else _abi.ptr(win.cursor),
                    0 if win is None else win.n, st()), "spo_wide_rows_clip_adam_dev_log")
                return
        else:
            obs, act, logp_old, tgt_r, tgt_c, adv = self._gather(idx)
            n = obs.shape[0]
            (v_r, ws_r), (v_c, ws_c), (mu, ws_a) = w.forward_multi("rca", obs, slot=1)
            d_vr = torch.empty(n, dtype=torch.float32, device=self.dev)
            d_vc, d_mu = torch.empty_like(d_vr), torch.empty((n, self.A), dtype=torch.float32, device=self.dev)
            _abi.check(lib.spo_wide_ppo_loss(_abi.ptr(v_r), _abi.ptr(v_c), _abi.ptr(mu), _abi.ptr(self.policy.theta[off_ls:]), _abi.ptr(act),
                                             _abi.ptr(logp_old), _abi.ptr(adv), _abi.ptr(tgt_r), _abi.ptr(tgt_c), n, self.A, float(cfg.clip),
                                             _abi.ptr(d_vr), _abi.ptr(d_vc), _abi.ptr(d_mu), _abi.ptr(g[off_ls:]), _abi.ptr(losses_out),
                                             _abi.ptr(self.loss_partials), self.loss_partials.numel(), st()), "spo_wide_ppo_loss")
            w.backward_multi("rca", obs, [ws_r, ws_c, ws_a], [d_vr, d_vc, d_mu], g)
        self._reduce_flat_grad()
        if dev_clock:
            self._clip_adam_dev(cfg, 0, w.P, 0, 0, losses_out, losses_out, idx if isinstance(idx, PermWindow) else None)
            return
        _abi.check(lib.spo_wide_clip_adam(_abi.ptr(self.policy.theta), _abi.ptr(g), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), w.P,
                                          w.off_c, w.off_ls, w.off_ls, cfg, self.adam_step, _abi.ptr(losses_out), _abi.ptr(self.scal4),
                                          _abi.ptr(self.loss_partials), self.loss_partials.numel(), st()), "spo_wide_clip_adam")
        self.adam_step += 1

    def _feature_split_kernel_ok(self, cfg) -> bool:
        """hidden [64, 64] with obs_dim <= 512 / act_dim <= 32 (HumanoidVelocity's 376 / 17) at minibatches of <= 64 rows on one
        GPU: the persistent feature-split kernel (csrc/update_ks.hip, round 5) instead of the launch-per-layer wide step.
        SPO_WIDE_KS=0 keeps the wide step."""
        return (list(self.policy.hidden_sizes) == [64, 64] and self.comm.world_size == 1
                and os.environ.get("SPO_WIDE_KS", "1") != "0" and bool(self.lib.spo_ks_supported(self.D, self.A, int(cfg.batch))))

    def _feature_split_grad_ok(self, cfg) -> bool:
        """The same dims under data parallelism (world_size > 1): the feature-split kernel computes one minibatch's gradient
        per launch (spo_ppo_lag_grad_ks) and the optimiser step stays outside, behind the all-reduce.  SPO_WIDE_KS=0: the
        launch-per-layer step."""
        return (list(self.policy.hidden_sizes) == [64, 64] and self.comm.world_size > 1
                and os.environ.get("SPO_WIDE_KS", "1") != "0" and bool(self.lib.spo_ks_supported(self.D, self.A, int(cfg.batch))))

    def check_sync_error(self):
        code = int(self.sync_ws[8].item()) & 0xFFFFFFFF
        if code:
            self.sync_ws[8] = 0
            raise _abi.SpoError("feature-split update kernel: inter-workgroup exchange timed out")

    def learning_iter(self, perm: torch.Tensor) -> torch.Tensor:
        cfg = self._cfg_struct()
        perm = _abi.require_gpu_tensor(perm, "perm", torch.int32)
        M = self.M
        n_mb = (M + cfg.batch - 1) // cfg.batch
        losses = torch.empty((n_mb, 3), dtype=torch.float32, device=self.dev)
        if self._feature_split_kernel_ok(cfg):
            d, b = self.buffer.data, self.buffer
            _abi.check(self.lib.spo_ppo_lag_update_iter_ks(
                _abi.ptr(self.policy.theta), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), self.adam_step, _abi.ptr(d["obs"]),
                _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]), _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]),
                _abi.ptr(b.adv_mix), _abi.ptr(perm), M, cfg, _abi.ptr(losses), _abi.ptr(self.sync_ws), _abi.stream_ptr()),
                "spo_ppo_lag_update_iter_ks")
            self.adam_step += n_mb
            return losses
        if self._feature_split_grad_ok(cfg):
            # data-parallel at the feature-split kernel's dims (round 6): per minibatch step ONE launch for the forward / loss /
            # backward of the three networks (spo_ppo_lag_grad_ks), the all-reduce of the flat gradient, the joint clip + Adam
            d, b, w, g = self.buffer.data, self.buffer, self.wide, self.flat_grad
            for k in range(n_mb):
                idx = perm[k * cfg.batch:(k + 1) * cfg.batch]
                _abi.check(self.lib.spo_ppo_lag_grad_ks(
                    _abi.ptr(self.policy.theta), _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]),
                    _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(b.adv_mix), _abi.ptr(idx), idx.numel(),
                    cfg, _abi.ptr(g), _abi.ptr(losses[k]), _abi.ptr(self.sync_ws), _abi.stream_ptr()), "spo_ppo_lag_grad_ks")
                self._reduce_flat_grad()
                _abi.check(self.lib.spo_wide_clip_adam(
                    _abi.ptr(self.policy.theta), _abi.ptr(g), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), w.P, w.off_c, w.off_ls,
                    w.off_ls, cfg, self.adam_step, _abi.ptr(losses[k]), _abi.ptr(self.scal4), _abi.ptr(self.loss_partials),
                    self.loss_partials.numel(), _abi.stream_ptr()), "spo_wide_clip_adam")
                self.adam_step += 1
            return self._mean_over_ranks_(losses)
        perm = perm.long()
        graphed = 0 < cfg.batch <= self.graph_max_batch and n_mb > 2
        n_full = M // cfg.batch if graphed else 0
        if graphed:
            self._sync_pow4()
            self._graphed_pass(("ppo", self._graph_cfg_key(cfg)), perm, cfg.batch, n_full, losses,
                               lambda i_, l_: self.minibatch_step(i_, l_, dev_clock=True, cfg=cfg))
            self.adam_step += n_full
        for k in range(n_full, n_mb):               # (eager: everything without graphs; else the ragged last minibatch, host clocks)
            self.minibatch_step(perm[k * cfg.batch:(k + 1) * cfg.batch], losses[k], cfg=cfg)
        return self._mean_over_ranks_(losses)

    def minibatch_step_ex(self, idx, adv_all, losses_out, actor_loss, kl_bound, pg_coef, actor_only, dev_clock: bool = False,
                          cfg=None) -> None:
        """One FOCOPS minibatch step (focops.py:312-347) or one step of CUP's actor-only second stage (cup.py:370-386) on the
        wide kernels: spo_update_iter_ex's semantics (include/safepo_hip.h), one minibatch."""
        w, lib, st = self.wide, self.lib, _abi.stream_ptr
        cfg = self._cfg_struct() if cfg is None else cfg
        klpen = actor_loss == _abi.ACTOR_LOSS_KL_PENALTY
        got = self._gather(idx, adv_all, extra=(self.mean_old,) if klpen else ())
        obs, act, logp_old, tgt_r, tgt_c, adv = got[:6]
        n = obs.shape[0]
        g, off_ls, A = self.flat_grad, w.off_ls, self.A
        part, cap = self.loss_partials, self.loss_partials.numel()
        nets = "a" if actor_only else "rca"
        fw = w.forward_multi(nets, obs, slot=1)
        mu, ws_a = fw[-1]
        wss, d_outs = [ws for _, ws in fw], []
        if not actor_only:
            (v_r, _), (v_c, _) = fw[0], fw[1]
            d_vr = torch.empty(n, dtype=torch.float32, device=self.dev)
            d_vc = torch.empty_like(d_vr)
            _abi.check(lib.spo_wide_critic_loss(_abi.ptr(v_r), _abi.ptr(v_c), _abi.ptr(tgt_r), _abi.ptr(tgt_c), n, _abi.ptr(d_vr),
                                                _abi.ptr(d_vc), _abi.ptr(losses_out), _abi.ptr(part), cap, st()), "spo_wide_critic_loss")
            d_outs = [d_vr, d_vc]
        d_mu = torch.empty((n, A), dtype=torch.float32, device=self.dev)
        if klpen:
            old_mean = got[6]
            mode, p0, p1, om, os_ = _abi.WIDE_ACTOR_KLPEN, float(kl_bound), float(pg_coef), _abi.ptr(old_mean), _abi.ptr(self.std_old)
        else:
            mode, p0, p1, om, os_ = _abi.WIDE_ACTOR_CLIP, float(cfg.clip), 0.0, None, None
        _abi.check(lib.spo_wide_actor_loss(mode, _abi.ptr(mu), _abi.ptr(self.policy.theta[off_ls:]), _abi.ptr(act), _abi.ptr(logp_old),
                                           _abi.ptr(adv), om, os_, n, n, A, p0, p1, _abi.ptr(d_mu), _abi.ptr(self.actor_sums), 0,
                                           _abi.ptr(losses_out[2:]), _abi.ptr(g[off_ls:]), _abi.ptr(part), cap, st()),
                   "spo_wide_actor_loss")
        w.backward_multi(nets, obs, wss, d_outs + [d_mu], g)
        self._reduce_flat_grad(off_ls if actor_only else 0)
        step_c, step_a = self.adam_step, self.adam_step + self.adam_step_actor_extra
        lo, norm0 = (off_ls, off_ls) if actor_only else (0, 0)
        if dev_clock:
            self._clip_adam_dev(cfg, lo, w.P, norm0, 0, None if actor_only else losses_out, losses_out,
                                idx if isinstance(idx, PermWindow) else None)
            return
        _abi.check(lib.spo_wide_clip_adam_ex(_abi.ptr(self.policy.theta), _abi.ptr(g), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), w.P,
                                             w.off_c, w.off_ls, w.off_ls, cfg, step_c, step_a, lo, w.P, norm0, 0,
                                             None if actor_only else _abi.ptr(losses_out), _abi.ptr(self.scal4), _abi.ptr(part), cap,
                                             st()), "spo_wide_clip_adam_ex")

    def learning_iter_ex(self, perm: torch.Tensor, adv: torch.Tensor, actor_loss: int = 0,
                         kl_bound: float = float("inf"), pg_coef: float = 0.0, actor_only: bool = False) -> torch.Tensor:
        if self.comm.world_size != 1:            # (as PPOLagEngine.learning_iter_ex: the indicator fraction of the KL-penalty loss
            raise NotImplementedError("focops / cup run on one GPU in this build (no data-parallel form of the "   # couples the
                                      "KL-penalty minibatch step)")                                             # global minibatch)
        cfg = self._cfg_struct()
        adv = _abi.require_gpu_tensor(adv, "adv", torch.float32)
        M = self.M
        n_mb = (M + cfg.batch - 1) // cfg.batch
        losses = torch.full((n_mb, 3), float("nan"), dtype=torch.float32, device=self.dev)
        if self._feature_split_kernel_ok(cfg):
            # FOCOPS / CUP at HumanoidVelocity-class dims: one launch of the persistent feature-split kernel (round 5)
            perm = _abi.require_gpu_tensor(perm, "perm", torch.int32)
            d = self.buffer.data
            _abi.check(self.lib.spo_update_iter_ex_ks(
                _abi.ptr(self.policy.theta), _abi.ptr(self.adam_m), _abi.ptr(self.adam_v), self.adam_step,
                self.adam_step + self.adam_step_actor_extra, _abi.ptr(d["obs"]), _abi.ptr(d["act"]),
                _abi.ptr(d["log_prob"]), _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(adv),
                _abi.ptr(perm), M, cfg, int(actor_loss), _abi.ptr(self.mean_old), _abi.ptr(self.std_old),
                float(kl_bound), float(pg_coef), int(actor_only), _abi.ptr(losses), _abi.ptr(self.sync_ws),
                _abi.stream_ptr()), "spo_update_iter_ex_ks")
            if actor_only:
                self.adam_step_actor_extra += n_mb
            else:
                self.adam_step += n_mb
            return losses
        perm = _abi.require_gpu_tensor(perm, "perm", torch.int32).long()
        graphed = 0 < cfg.batch <= self.graph_max_batch and n_mb > 2
        n_full = M // cfg.batch if graphed else 0
        if graphed:
            self._sync_pow4()
            key = ("ex", self._graph_cfg_key(cfg), int(actor_loss), float(kl_bound), float(pg_coef), bool(actor_only), adv.data_ptr())
            self._graphed_pass(key, perm, cfg.batch, n_full, losses,
                               lambda i_, l_: self.minibatch_step_ex(i_, adv, l_, actor_loss, kl_bound, pg_coef, actor_only,
                                                                     dev_clock=True, cfg=cfg))
            if actor_only:
                self.adam_step_actor_extra += n_full
            else:
                self.adam_step += n_full
        for k in range(n_full, n_mb):
            self.minibatch_step_ex(perm[k * cfg.batch:(k + 1) * cfg.batch], adv, losses[k], actor_loss, kl_bound, pg_coef, actor_only,
                                   cfg=cfg)
            if actor_only:
                self.adam_step_actor_extra += 1
            else:
                self.adam_step += 1
        return self._mean_over_ranks_(losses)
