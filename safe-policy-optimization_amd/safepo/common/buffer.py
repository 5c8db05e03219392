"""VectorizedOnPolicyBuffer: dense device-resident on-policy buffer + fused HIP GAE.

Same constructor / store / finish_path / get surface as the reference
(safepo/common/buffer.py:24-164), different storage: instead of a Python list of `num_envs` dicts
of per-env tensors and a Python reverse loop per path, the data lives in dense [num_envs, T]
tensors in HBM (env-major: flat row = env*T + t, the order reference get() concatenates in,
buffer.py:149-153), path ends are a u8 mask `seg_end[N,T]` plus bootstrap values `boot_r/boot_c`,
and ALL paths (reward and cost) are scanned by one kernel launch at get() time
(spo_gae_fused, csrc/gae.hip).  get() returns zero-copy flattened views.
"""
from __future__ import annotations

import torch

from safepo import _abi

SCALAR_KEYS = ("reward", "cost", "done", "value_r", "value_c", "adv_r", "adv_c", "target_value_r",
               "target_value_c", "log_prob")


class VectorizedOnPolicyBuffer:
    def __init__(self, obs_space, act_space, size: int, gamma: float = 0.99, lam: float = 0.95,
                 lam_c: float = 0.95, standardized_adv_r: bool = True, standardized_adv_c: bool = True,
                 device="cpu", num_envs: int = 1) -> None:
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _abi.SpoError(
                "VectorizedOnPolicyBuffer (MI355X) stores its data in HBM and computes GAE with a HIP kernel; "
                f"device={device!r} is not a GPU (no CPU fallback)")
        self._lib = _abi.load()
        self.num_envs, self.size = int(num_envs), int(size)
        self.obs_dim = int(obs_space.shape[0]) if len(obs_space.shape) else 1
        self.act_dim = int(act_space.shape[0]) if len(act_space.shape) else 1
        N, T = self.num_envs, self.size
        f32 = dict(dtype=torch.float32, device=dev)
        self.data = {"obs": torch.zeros((N, T, *obs_space.shape), **f32),
                     "act": torch.zeros((N, T, *act_space.shape), **f32)}
        for k in SCALAR_KEYS:
            self.data[k] = torch.zeros((N, T), **f32)
        self.seg_end = torch.zeros((N, T), dtype=torch.uint8, device=dev)
        self.boot_r = torch.zeros((N, T), **f32)
        self.boot_c = torch.zeros((N, T), **f32)
        self.adv_mix = torch.zeros((N, T), **f32)
        # reward / cost with gamma * bootstrap folded in at path ends (written by spo_boundary_step_fold, one column per
        # collect step); the scan reads these instead of reward/cost + boot_r/boot_c when every column of the epoch
        # came from that kernel (self._fold_cols == size)
        self.reward_fold = torch.zeros((N, T), **f32)
        self.cost_fold = torch.zeros((N, T), **f32)
        self._fold_cols = 0
        self._partials = torch.zeros((max(self._lib.spo_gae_num_blocks(N, T), 1), _abi.GAE_PARTIAL_STRIDE), dtype=torch.float64,
                                     device=dev)
        self.sums = torch.zeros(4, dtype=torch.float64, device=dev)
        self.stats = torch.zeros(3, **f32)
        self._gamma, self._lam, self._lam_c = gamma, lam, lam_c
        self._standardized_adv_r, self._standardized_adv_c = standardized_adv_r, standardized_adv_c
        self._device = dev
        self.ptr = 0
        self.ptr_list = [0] * N                 # kept for API parity (all entries equal self.ptr)
        self.path_start_idx_list = [0] * N

    # ------------------------------------------------------------------ reference API
    def store(self, **data: torch.Tensor) -> None:
        """Write one vector step (buffer.py:84-95).  The fused collect kernel writes the same slots
        directly (safepo.common.engine); this method is the API-compatible path."""
        assert self.ptr < self.size, "Buffer overflow"
        if self.ptr == 0:
            # first store of an epoch: the path marks of the previous epoch must not cut this epoch's paths (the
            # engine path never gets here -- spo_boundary_step rewrites every slot of a column)
            self.clear_boundaries()
        for key, value in data.items():
            self.data[key][:, self.ptr] = torch.as_tensor(value, dtype=torch.float32, device=self._device)
        self._fold_cols = -1          # a column written through the API has no folded bootstrap: scan the plain arrays
        self.advance()

    def advance(self) -> None:
        self.ptr += 1
        self.ptr_list = [self.ptr] * self.num_envs

    def finish_path(self, last_value_r: torch.Tensor | None = None, last_value_c: torch.Tensor | None = None,
                    idx: int = 0) -> None:
        """Mark the end of env `idx`'s current path and remember its bootstrap values
        (buffer.py:97-140).  The scan itself is deferred to get()."""
        if self.ptr == 0 or self.path_start_idx_list[idx] >= self.ptr:
            return
        t = self.ptr - 1
        # the marks below change seg_end / boot_* only: reward_fold / cost_fold of an engine-collected epoch would be stale
        # (ADVICE r02), so the epoch falls back to the unfolded scan (reward/cost + boot_r/boot_c, bit-identical results)
        self._fold_cols = -1
        self.seg_end[idx, t] = 1
        self.boot_r[idx, t] = 0.0 if last_value_r is None else torch.as_tensor(last_value_r).reshape(-1)[0]
        self.boot_c[idx, t] = 0.0 if last_value_c is None else torch.as_tensor(last_value_c).reshape(-1)[0]
        self.path_start_idx_list[idx] = self.ptr

    def compute_gae(self, lagrangian_multiplier: float | None = None, comm=None, force_unfolded: bool = False) -> None:
        """All paths, reward + cost, one launch; then the get() statistics (buffer.py:154-160) and,
        if a multiplier is given, the PPO-Lag advantage mix (ppo_lag.py:280-281) into self.adv_mix.
        `comm`: optional safepo.parallel.Comm -- statistics are all-reduced over the env shards.
        Folded form: when every column of the epoch was written by the engine's boundary kernel, the scan reads
        reward_fold / cost_fold (reward + gamma * bootstrap at path ends) and no bootstrap arrays.  store() and
        finish_path() invalidate it; a caller that edits data["reward"] / data["cost"] in place after an engine-collected
        epoch (reward shaping) passes force_unfolded=True so the edit is what the scan sees."""
        d, lib, st = self.data, self._lib, _abi.stream_ptr()
        N, T = self.num_envs, self.size
        folded = self._fold_cols == T and self.ptr == T and not force_unfolded
        rew, cst = (self.reward_fold, self.cost_fold) if folded else (d["reward"], d["cost"])
        boot_r, boot_c = (None, None) if folded else (self.boot_r, self.boot_c)
        self.last_scan_folded = folded
        self._scan_args = (_abi.ptr(rew), _abi.ptr(cst), _abi.ptr(d["value_r"]), _abi.ptr(d["value_c"]),
                           _abi.ptr(self.seg_end), _abi.ptr(boot_r), _abi.ptr(boot_c), _abi.ptr(d["adv_r"]),
                           _abi.ptr(d["adv_c"]), _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]),
                           _abi.ptr(self._partials), N, T, self._gamma, self._lam, self._lam_c)
        def launch_scan():
            return lib.spo_gae_fused(*self._scan_args, _abi.stream_ptr())
        self._launch_scan = launch_scan
        _abi.check(launch_scan(), "spo_gae_fused")
        _abi.check(lib.spo_adv_reduce(_abi.ptr(self._partials), self._partials.shape[0], _abi.ptr(self.sums), st),
                   "spo_adv_reduce")
        if comm is not None and comm.world_size > 1:
            comm.all_reduce_sum_(self.sums)
        lam = 0.0 if lagrangian_multiplier is None else float(lagrangian_multiplier)
        _abi.check(lib.spo_adv_apply(
            _abi.ptr(d["adv_r"]), _abi.ptr(d["adv_c"]),
            _abi.ptr(self.adv_mix) if lagrangian_multiplier is not None else None,
            _abi.ptr(self.sums), N * T, lam, int(self._standardized_adv_r), int(self._standardized_adv_c),
            _abi.ptr(self.stats), st), "spo_adv_apply")

    def get(self, lagrangian_multiplier: float | None = None, comm=None) -> dict[str, torch.Tensor]:
        """Flattened [N*T, ...] views in env-major order with adv_r standardised and adv_c centred
        (buffer.py:142-164); resets the write pointer.  Views alias the buffer: they are valid
        until the next epoch's stores."""
        self.compute_gae(lagrangian_multiplier, comm)
        M = self.num_envs * self.size
        out = {k: v.reshape(M, *v.shape[2:]) for k, v in self.data.items()}
        if lagrangian_multiplier is not None:
            out["advantage"] = self.adv_mix.reshape(M)
        self.reset()
        return out

    def reset(self) -> None:
        self.ptr = 0
        self._fold_cols = 0
        self.ptr_list = [0] * self.num_envs
        self.path_start_idx_list = [0] * self.num_envs

    def clear_boundaries(self) -> None:
        self.seg_end.zero_()
        self.boot_r.zero_()
        self.boot_c.zero_()

    def time_scan(self, reps: int = 20) -> float:
        """Average GPU time (seconds) of one spo_gae_fused launch on the current buffer contents:
        `reps` launches captured in a HIP graph and replayed between two HIP events, so neither host
        launch cost nor event resolution pollutes a ~5 us kernel.  The scan is idempotent.
        Falls back to an eager loop if graph capture is unavailable."""
        launch = self._launch_scan
        torch.cuda.synchronize(self._device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            side = torch.cuda.Stream(device=self._device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                _abi.check(launch(), "spo_gae_fused")
                torch.cuda.synchronize(self._device)
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    for _ in range(reps):
                        launch()
                graph.replay()
                torch.cuda.synchronize(self._device)
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize(self._device)
        except Exception:
            torch.cuda.synchronize(self._device)
            e0.record()
            for _ in range(reps):
                launch()
            e1.record()
            torch.cuda.synchronize(self._device)
        return e0.elapsed_time(e1) * 1e-3 / reps


    def time_scan_dispatches(self, reps: int = 50, warm: int = 200):
        """Per-dispatch GPU time (seconds, one entry per launch) of spo_gae_fused on the current buffer contents: every
        dispatch carries its own start / stop events (spo_gae_fused_timed), i.e. the timestamps of the dispatch packet
        itself -- the per-dispatch duration rocprofv3 --kernel-trace reports, unlike the graph average of time_scan,
        in which the command processor overlaps one dispatch's end-of-kernel tail with the next one's start."""
        import ctypes
        args = self._scan_args
        out = (ctypes.c_float * (warm + reps))()
        torch.cuda.synchronize(self._device)
        # `warm` untimed-in-effect dispatches first (~0.9 ms): for the first few hundred microseconds after the GPU switches
        # from a light load (the persistent update kernel keeps 3 CUs busy for seconds; the rollout is tiny kernels) to
        # full-chip launches, scattered dispatches take 2-6x the steady 4.3 us (power-management transient: the series is
        # in tools/gae_after_update.py / SPO_BENCH_DUMP_DISPATCHES); a second batch right after is flat.  The steady state
        # is the kernel's figure; the one scan of a real epoch costs 4-30 us of a 3.6 s epoch either way
        _abi.check(self._lib.spo_gae_fused_timed(*args, warm + reps, out, _abi.stream_ptr()), "spo_gae_fused_timed")
        return [float(x) * 1e-6 for x in out][warm:]


class SeparatedReplayBuffer:
    """Per-agent multi-agent buffer with the reference's attributes and time-major shapes
    (safepo/common/buffer.py:209-465): share_obs/obs [T+1, N, ...], value_preds/cost_preds/returns/cost_returns/
    masks [T+1, N, 1], rewards/costs/actions/action_log_probs/factor [T, N, ...].  Storage is device memory;
    compute_returns / compute_cost_returns (buffer.py:356-384) run on the fused kernel spo_ma_gae (each writes only its own
    output, like the reference; compute_returns_and_cost_returns does both in one launch) -- bit-identical to the
    reference's fp32 Python loop."""

    def __init__(self, config, obs_space, share_obs_space, act_space):
        self.episode_length = config["episode_length"]
        self.n_rollout_threads = config["n_rollout_threads"]
        # recurrent policies are not built: the rnn_states* tensors keep their place in the API with a hidden dim of 1
        # (the reference allocates [T+1, N, recurrent_N, hidden_size] zeros for each of the three, buffer.py:239-244)
        self.rnn_hidden_size = 1
        self.recurrent_N = config["recurrent_N"]
        self.gamma, self.gae_lambda = config["gamma"], config["gae_lambda"]
        self._use_gae, self._use_popart = config["use_gae"], config["use_popart"]
        self._use_valuenorm = config["use_valuenorm"]
        self._use_proper_time_limits = config["use_proper_time_limits"]
        self.device = torch.device(config.get("device", "cuda:0"))
        self.algo = config["algorithm_name"]
        if self.device.type != "cuda":
            raise _abi.SpoError("SeparatedReplayBuffer (MI355X) lives in HBM; device must be a GPU (no CPU fallback)")
        self._lib = _abi.load()
        T, N = self.episode_length, self.n_rollout_threads
        obs_shape, share_obs_shape = tuple(obs_space.shape), tuple(share_obs_space.shape)
        act_dim = act_space.shape[0]
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=self.device)
        self.aver_episode_costs = z(T + 1, N, *obs_shape)
        self.share_obs, self.obs = z(T + 1, N, *share_obs_shape), z(T + 1, N, *obs_shape)
        self.rnn_states = z(T + 1, N, self.recurrent_N, self.rnn_hidden_size)
        self.rnn_states_critic, self.rnn_states_cost = torch.zeros_like(self.rnn_states), torch.zeros_like(self.rnn_states)
        self.value_preds, self.returns = z(T + 1, N, 1), z(T + 1, N, 1)
        self.cost_preds, self.cost_returns = z(T + 1, N, 1), z(T + 1, N, 1)
        self.available_actions = None
        self.actions, self.action_log_probs = z(T, N, act_dim), z(T, N, act_dim)
        self.rewards, self.costs = z(T, N, 1), z(T, N, 1)
        self.masks = torch.ones(T + 1, N, 1, dtype=torch.float32, device=self.device)
        self.bad_masks, self.active_masks = torch.ones_like(self.masks), torch.ones_like(self.masks)
        self.factor = torch.ones(T, N, 1, dtype=torch.float32, device=self.device)
        self.step = 0

    def update_factor(self, factor):
        self.factor.copy_(factor)

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None, cost_adv=None, perm=None):
        """buffer.py:386-465: `num_mini_batch` index sets of a random permutation of the T*N rows; yields the reference's
        18-tuple (mappolag / macpo form).  Row gathers are device index_selects; `perm` may carry the permutation."""
        T, N = self.rewards.shape[0:2]
        batch_size = N * T
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch, (N, T, num_mini_batch)
            mini_batch_size = batch_size // num_mini_batch
        # the shuffle is drawn on the device (a 524 288-element CPU randperm costs ~10 ms per learning iteration)
        one_batch = perm is None and (mini_batch_size == batch_size)
        rand = (torch.arange(1, device=self.device) if one_batch else torch.randperm(batch_size, device=self.device)) \
            if perm is None else torch.as_tensor(perm).to(self.device)
        flat = lambda t: t.reshape(-1, *t.shape[2:])
        share_obs, obs = flat(self.share_obs[:-1]), flat(self.obs[:-1])
        rnn, rnn_c, rnn_k = flat(self.rnn_states[:-1]), flat(self.rnn_states_critic[:-1]), flat(self.rnn_states_cost[:-1])
        actions, logp = flat(self.actions), flat(self.action_log_probs)
        value_preds, returns = self.value_preds[:-1].reshape(-1, 1), self.returns[:-1].reshape(-1, 1)
        cost_preds, cost_returns = self.cost_preds[:-1].reshape(-1, 1), self.cost_returns[:-1].reshape(-1, 1)
        masks, active = self.masks[:-1].reshape(-1, 1), self.active_masks[:-1].reshape(-1, 1)
        factor = self.factor.reshape(-1, self.factor.shape[-1])
        advantages = advantages.reshape(-1, 1)
        cost_adv = cost_adv.reshape(-1, 1) if cost_adv is not None else None
        # One minibatch = the whole buffer (the reference default): every loss is a mean over all rows, so the shuffle only
        # changes the summation order.  Without a caller-supplied permutation the rows are handed over in place -- no
        # 0.5 GB gather per learning iteration.
        whole = perm is None and mini_batch_size == batch_size
        for i in range(num_mini_batch if num_mini_batch is not None else batch_size // mini_batch_size):
            idx = rand[i * mini_batch_size:(i + 1) * mini_batch_size]
            g = (lambda t: t) if whole else (lambda t: t.index_select(0, idx))
            head = (g(share_obs), g(obs), g(rnn), g(rnn_c), g(actions), g(value_preds), g(returns), g(masks), g(active),
                    g(logp), g(advantages), None, g(factor))
            if self.algo not in ("mappolag", "macpo"):           # buffer.py:462-464: the 13-tuple of happo / mappo
                yield head
                continue
            yield head + (g(cost_preds), g(cost_returns), g(rnn_k), g(cost_adv) if cost_adv is not None else None,
                          self.aver_episode_costs)

    def return_aver_insert(self, aver_episode_costs):
        self.aver_episode_costs = aver_episode_costs.clone()

    def insert(self, share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds, rewards,
               masks, bad_masks=None, active_masks=None, available_actions=None, costs=None, cost_preds=None,
               rnn_states_cost=None, done_episodes_costs_aver=None, aver_episode_costs=0):
        s = self.step
        self.share_obs[s + 1].copy_(share_obs)
        self.obs[s + 1].copy_(obs)
        # recurrent policies are not built: the three rnn_states tensors are, and stay, zeros (no per-step copies)
        self.actions[s].copy_(actions)
        self.action_log_probs[s].copy_(action_log_probs)
        self.value_preds[s].copy_(value_preds)
        self.rewards[s].copy_(rewards)
        self.masks[s + 1].copy_(masks)
        if bad_masks is not None:
            self.bad_masks[s + 1].copy_(bad_masks)
        if active_masks is not None:
            self.active_masks[s + 1].copy_(active_masks)
        if costs is not None:
            self.costs[s].copy_(costs)
        if cost_preds is not None:
            self.cost_preds[s].copy_(cost_preds)
        self.step = (s + 1) % self.episode_length

    def after_update(self):
        for name in ("share_obs", "obs", "masks", "bad_masks", "active_masks"):
            t = getattr(self, name)
            t[0].copy_(t[-1])

    def compute_returns_and_cost_returns(self, next_value, next_cost, value_normalizer=None, cost_normalizer=None,
                                         _returns_out=None, _cost_returns_out=None):
        """Both GAE recurrences in one launch (the reference's compute() calls them back to back per agent,
        mappolag.py:583-597)."""
        self.value_preds[-1] = next_value
        self.cost_preds[-1] = next_cost
        sd_r, mu_r = value_normalizer.denorm_scalars() if value_normalizer is not None else (1.0, 0.0)
        sd_c, mu_c = cost_normalizer.denorm_scalars() if cost_normalizer is not None else (1.0, 0.0)
        ret = self.returns if _returns_out is None else _returns_out
        cret = self.cost_returns if _cost_returns_out is None else _cost_returns_out
        _abi.check(self._lib.spo_ma_gae(
            _abi.ptr(self.rewards), _abi.ptr(self.costs), _abi.ptr(self.value_preds), _abi.ptr(self.cost_preds),
            _abi.ptr(self.masks), _abi.ptr(ret), _abi.ptr(cret), self.episode_length,
            self.n_rollout_threads, self.gamma, self.gae_lambda, sd_r, mu_r, sd_c, mu_c, _abi.stream_ptr()),
            "spo_ma_gae")

    def _other_side_scratch(self):
        # the fused kernel always runs both recurrences; the side that was not asked for lands here, so that (like the
        # reference, buffer.py:356-384) compute_returns touches only `returns` and compute_cost_returns only `cost_returns`
        if getattr(self, "_gae_scratch", None) is None:
            self._gae_scratch = torch.empty_like(self.returns)
        return self._gae_scratch

    def compute_returns(self, next_value, value_normalizer=None):
        """buffer.py:356-377."""
        self.compute_returns_and_cost_returns(next_value, self.cost_preds[-1].clone(), value_normalizer, None,
                                              _cost_returns_out=self._other_side_scratch())

    def compute_cost_returns(self, next_cost, value_normalizer=None):
        """buffer.py:379-384."""
        self.compute_returns_and_cost_returns(self.value_preds[-1].clone(), next_cost, None, value_normalizer,
                                              _returns_out=self._other_side_scratch())
