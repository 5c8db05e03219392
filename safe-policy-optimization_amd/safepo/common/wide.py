"""ActorVCritic outside the persistent kernels' envelope (hidden_sizes other than [64, 64], obs_dim > 128, act_dim > 16): host
side of the wide-network kernels.

The reference builds its MLPs for any `hidden_sizes` (safepo/common/model.py:30-48,131) and selects
`[1024, 1024, 512]` with minibatches of steps_per_epoch // 4 rows for Isaac Gym tasks
(isaac_gym_specific_cfg, safepo/single_agent/ppo_lag.py:54-65).  The persistent kernels of csrc/update.hip are built
around a 64-wide network living in one CU's LDS; every other shape runs here, as a sequence of launches on the in-tree
fp32 MFMA GEMM kernels (csrc/ma_net.hip: spo_mlp_forward / spo_mlp_backward / spo_wide_ppo_loss / spo_wide_clip_adam).
Same flat parameter vector, same state_dict keys, same optimiser semantics (critic L2 terms, joint clip_grad_norm_,
three Adam optimisers with the actor's own learning rate, ppo_lag.py:306-329).
"""
from __future__ import annotations

import torch

from safepo import _abi

KL_CHUNK = 65536        # rows per full-batch actor evaluation (activations of [1024, 1024, 512]: 0.7 GB per chunk)


class PermWindow:
    """The rows perm[cursor : cursor + n] of a device-resident permutation, cursor (int64[1]) on the device too: what a
    minibatch step replayed from a HIP graph gathers -- spo_gather_rows_at reads the window, spo_wide_clip_adam_dev_log stores
    the step's losses at loss_log[cursor / n] and moves the cursor on."""

    def __init__(self, capacity: int, n: int, device):
        self.perm = torch.zeros(capacity, dtype=torch.int64, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=device)
        self.loss_log = torch.full(((capacity + n - 1) // n, 3), float("nan"), dtype=torch.float32, device=device)
        self.n = int(n)

    def numel(self) -> int:
        return self.n

    def load(self, perm: torch.Tensor) -> None:
        self.perm[:perm.numel()].copy_(perm)
        self.cursor.zero_()


class WideNets:
    """The three networks of an ActorVCritic as views of its flat parameter vector, with cached workspaces."""

    def __init__(self, policy):
        self.policy = policy
        self.theta_key = (policy.theta.data_ptr(), policy.theta.device)      # ActorVCritic.wide rebuilds when theta moved
        self.lib = _abi.load()
        D, A, hs = policy.obs_dim, policy.act_dim, list(policy.hidden_sizes)
        self.D, self.A = D, A
        self.net_c = _abi.MlpNet.of([D] + hs + [1])
        self.net_a = _abi.MlpNet.of([D] + hs + [A])
        self.Pc = int(self.lib.spo_mlp_param_count(self.net_c))
        self.Pa = int(self.lib.spo_mlp_param_count(self.net_a))
        if self.Pc < 0 or self.Pa < 0:
            _abi.check(-2, "spo_mlp_param_count")
        # policy.parameters() order (model.py:131-135): reward critic, cost critic, actor.log_std, actor.mean.*
        self.off_r, self.off_c, self.off_ls, self.off_a = 0, self.Pc, 2 * self.Pc, 2 * self.Pc + A
        self.P = 2 * self.Pc + A + self.Pa
        assert self.P == policy.theta.numel(), (self.P, policy.theta.numel())
        self._ws, self._scratch = {}, {}

    # ------------------------------------------------------------------ pieces
    def theta_of(self, which):
        th = self.policy.theta
        return {"r": th[self.off_r:], "c": th[self.off_c:], "a": th[self.off_a:]}[which]

    def net_of(self, which):
        return self.net_a if which == "a" else self.net_c

    def _workspace(self, which, rows, slot=0):
        key = (which, rows, slot)
        ws = self._ws.get(key)
        if ws is None:
            n = int(self.lib.spo_mlp_workspace_floats(self.net_of(which), rows))
            ws = torch.empty(n, dtype=torch.float32, device=self.policy.theta.device)
            if len(self._ws) > 24:
                self._ws.clear()
            self._ws[key] = ws
        return ws

    def forward(self, which, x, slot=0):
        """Output [rows, out] (a view of the workspace, valid until the next forward with the same key) and the workspace."""
        rows = x.shape[0]
        ws = self._workspace(which, rows, slot)
        _abi.check(self.lib.spo_mlp_forward(_abi.ptr(self.theta_of(which)), self.net_of(which), _abi.ptr(x), rows, _abi.ptr(ws),
                                            _abi.stream_ptr()), "spo_mlp_forward")
        out_dim = self.A if which == "a" else 1
        return ws[ws.numel() - rows * out_dim:].view(rows, out_dim), ws

    def backward(self, which, x, ws, d_out, grad_flat):
        rows = x.shape[0]
        net = self.net_of(which)
        key = (which, rows)
        sc = self._scratch.get(key)
        if sc is None:
            n = int(self.lib.spo_mlp_backward_scratch_floats(net, rows))
            sc = torch.empty(n, dtype=torch.float32, device=x.device)
            if len(self._scratch) > 12:
                self._scratch.clear()
            self._scratch[key] = sc
        off = {"r": self.off_r, "c": self.off_c, "a": self.off_a}[which]
        _abi.check(self.lib.spo_mlp_backward(_abi.ptr(self.theta_of(which)), net, _abi.ptr(x), rows, _abi.ptr(ws), _abi.ptr(d_out),
                                             _abi.ptr(grad_flat[off:]), _abi.ptr(sc), _abi.stream_ptr()), "spo_mlp_backward")

    # ------------------------------------------------------------------ several networks, one launch each way (small row counts)
    def _bwd_scratch(self, which, rows):
        key = (which, rows)
        sc = self._scratch.get(key)
        if sc is None:
            n = int(self.lib.spo_mlp_backward_scratch_floats(self.net_of(which), rows))
            sc = torch.empty(max(n, 1), dtype=torch.float32, device=self.policy.theta.device)
            if len(self._scratch) > 12:
                self._scratch.clear()
            self._scratch[key] = sc
        return sc

    def forward_multi(self, whichs, x, slot=0):
        """[(output, workspace)] of the networks `whichs` ("r", "c", "a") on the same rows `x`: spo_mlp_forward_multi -- one
        launch for all of them at minibatch-sized row counts (csrc/mlp_small.hip), else a call per network."""
        import ctypes
        rows, n = x.shape[0], len(whichs)
        wss = [self._workspace(w, rows, slot) for w in whichs]
        vp = ctypes.c_void_p * n
        nets = (ctypes.POINTER(_abi.MlpNet) * n)(*[ctypes.pointer(self.net_of(w)) for w in whichs])
        _abi.check(self.lib.spo_mlp_forward_multi(n, vp(*[_abi.ptr(self.theta_of(w)) for w in whichs]), nets, vp(*[_abi.ptr(x)] * n), rows,
                                                  vp(*[_abi.ptr(ws) for ws in wss]), _abi.stream_ptr()), "spo_mlp_forward_multi")
        outs = []
        for w, ws in zip(whichs, wss):
            od = self.A if w == "a" else 1
            outs.append((ws[ws.numel() - rows * od:].view(rows, od), ws))
        return outs

    def backward_multi(self, whichs, x, wss, d_outs, grad_flat):
        """spo_mlp_backward_multi: the flat gradients of the networks `whichs` into their slices of `grad_flat`."""
        import ctypes
        rows, n = x.shape[0], len(whichs)
        off = {"r": self.off_r, "c": self.off_c, "a": self.off_a}
        vp = ctypes.c_void_p * n
        nets = (ctypes.POINTER(_abi.MlpNet) * n)(*[ctypes.pointer(self.net_of(w)) for w in whichs])
        _abi.check(self.lib.spo_mlp_backward_multi(
            n, vp(*[_abi.ptr(self.theta_of(w)) for w in whichs]), nets, vp(*[_abi.ptr(x)] * n), rows, vp(*[_abi.ptr(ws) for ws in wss]),
            vp(*[_abi.ptr(d) for d in d_outs]), vp(*[_abi.ptr(grad_flat[off[w]:]) for w in whichs]),
            vp(*[_abi.ptr(self._bwd_scratch(w, rows)) for w in whichs]), _abi.stream_ptr()), "spo_mlp_backward_multi")

    # ------------------------------------------------------------------ a whole minibatch gradient in one launch (round 6)
    def rows_grad_ok(self, rows, critics_only=False) -> bool:
        """csrc/mlp_rows.hip holds this shape: <= 256 rows, every level's images in one CU's LDS (SPO_WIDE_ROWS=0: never)."""
        return bool(self.lib.spo_wide_grad_rows_supported(self.net_c, None if critics_only else self.net_a, int(rows)))

    def grad_rows(self, idx, obs, act, logp_old, tgt_r, tgt_c, adv, clip, grad_flat, losses_out, critics_only=False, reduce=True):
        """ppo_lag.py:298-324 on the rows `idx` of the full [M, .] arrays (int64 device indices, or a PermWindow: the rows
        perm[cursor : cursor + n] with the cursor on the device): spo_wide_ppo_grad_rows (networks x ceil(n / 16) workgroups, one
        launch) + spo_wide_reduce_parts -> the flat data gradient in grad_flat[:n_params] and the data losses in losses_out.
        reduce=False: the row groups' partial gradients only (returned: spo_wide_rows_clip_adam_dev_log sums them itself)."""
        win = idx if isinstance(idx, PermWindow) else None
        n = idx.numel()
        P = 2 * self.Pc if critics_only else self.P
        key = ("parts", n, critics_only)
        parts = self._scratch.get(key)
        if parts is None:
            parts = torch.zeros(int(self.lib.spo_wide_grad_rows_part_floats(P, n)), dtype=torch.float32, device=self.policy.theta.device)
            self._scratch[key] = parts
        _abi.check(self.lib.spo_wide_ppo_grad_rows(
            _abi.ptr(self.policy.theta), self.net_c, None if critics_only else self.net_a, _abi.ptr(obs),
            None if critics_only else _abi.ptr(act), None if critics_only else _abi.ptr(logp_old), _abi.ptr(tgt_r), _abi.ptr(tgt_c),
            None if critics_only else _abi.ptr(adv), _abi.ptr(win.perm if win else idx), _abi.ptr(win.cursor) if win else None, n,
            float(clip), _abi.ptr(parts), _abi.stream_ptr()), "spo_wide_ppo_grad_rows")
        if reduce:
            _abi.check(self.lib.spo_wide_reduce_parts(_abi.ptr(parts), n, P, 2 if critics_only else 3, _abi.ptr(grad_flat),
                                                      _abi.ptr(losses_out), _abi.stream_ptr()), "spo_wide_reduce_parts")
        return parts

    def gather_rows(self, idx, srcs):
        """[src[idx] for src in srcs] (row-major float32 arrays, int64 device indices) in one launch (spo_gather_rows).  `idx` may
        be a PermWindow: the rows perm[cursor : cursor + n] with the cursor on the device (a replayed minibatch step)."""
        import ctypes
        win = idx if isinstance(idx, PermWindow) else None
        n, k = idx.numel(), len(srcs)
        dev = self.policy.theta.device
        srcs = [s_.view(s_.shape[0], -1) for s_ in srcs]
        dsts = [torch.empty((n, s_.shape[1]), dtype=torch.float32, device=dev) for s_ in srcs]
        vp = ctypes.c_void_p * k
        _abi.check(self.lib.spo_gather_rows_at(k, vp(*[_abi.ptr(s_) for s_ in srcs]), (ctypes.c_int * k)(*[s_.shape[1] for s_ in srcs]),
                                               vp(*[_abi.ptr(d) for d in dsts]), _abi.ptr(win.perm if win else idx),
                                               _abi.ptr(win.cursor) if win else None, n, _abi.stream_ptr()), "spo_gather_rows_at")
        return dsts

    def jvp_scratch(self, rows):
        key = ("jvp", rows)
        sc = self._scratch.get(key)
        if sc is None:
            sc = torch.empty(int(self.lib.spo_mlp_jvp_scratch_floats(self.net_a, rows)), dtype=torch.float32,
                             device=self.policy.theta.device)
            self._scratch[key] = sc
        return sc

    # ------------------------------------------------------------------ reference API pieces
    def values(self, obs):
        (v_r, _), (v_c, _) = self.forward_multi("rc", obs)
        return v_r.reshape(-1).clone(), v_c.reshape(-1).clone()

    def actor_mean(self, obs, out=None):
        """mean of policy.actor(obs) for any number of rows (chunked)."""
        M = obs.shape[0]
        out = torch.empty((M, self.A), dtype=torch.float32, device=obs.device) if out is None else out
        for lo in range(0, M, KL_CHUNK):
            mu, _ = self.forward("a", obs[lo:lo + KL_CHUNK])
            out[lo:lo + mu.shape[0]].copy_(mu)
        return out

    def step(self, obs, eps):
        """ActorVCritic.step (model.py:149-170): (act, logp, v_r, v_c); eps None = deterministic."""
        n = obs.shape[0]
        (v_r, _), (v_c, _), (mu, _) = self.forward_multi("rca", obs)      # (one launch for a handful of envs, csrc/mlp_small.hip)
        act = torch.empty((n, self.A), dtype=torch.float32, device=obs.device)
        logp = torch.empty(n, dtype=torch.float32, device=obs.device)
        ls = self.policy.theta[self.off_ls:self.off_ls + self.A]
        _abi.check(self.lib.spo_gauss_sample(_abi.ptr(mu), _abi.ptr(ls), _abi.ptr(eps), _abi.ptr(act), _abi.ptr(logp), n, self.A,
                                             _abi.stream_ptr()), "spo_gauss_sample")
        return act, logp, v_r.reshape(-1).clone(), v_c.reshape(-1).clone()
