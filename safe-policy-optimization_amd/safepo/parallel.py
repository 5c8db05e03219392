"""Data-parallel plumbing: one process per GPU, env shards, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference has no distributed layer at all.  What the sharded path has
to exchange: per epoch the advantage statistics (4 doubles) and the EpCost mean;
per minibatch step the flat gradient of all three networks (24 850 fp32 = 99 KB, latency-bound);
per learning iteration the KL sum (1 double) so every rank stops at the same iteration."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.distributed as dist


class Comm:
    """Thin wrapper so single-process runs need no process group."""

    def __init__(self, group=None):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = group
        self.world_size = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

    @classmethod
    def single(cls) -> "Comm":
        """A communicator of one rank, whatever process group exists (reference runs inside a distributed test)."""
        c = cls.__new__(cls)
        c.enabled, c.group, c.world_size, c.rank = False, None, 1, 0
        return c

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_max_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world_size > 1:
            dist.broadcast(t, src=src, group=self.group)
        return t

    def barrier(self):
        if self.world_size > 1:
            dist.barrier(group=self.group)


def init_from_env(backend: str | None = None) -> Comm:
    """Initialise torch.distributed from torchrun-style env vars (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        # gloo announces its connections on STDOUT from C++ ("[Gloo] Rank 0 is connected to ..."): a caller that prints a
        # machine-readable line (bench.py) must not find that in its output -- the descriptor is pointed at stderr meanwhile
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
            if dist.is_initialized():
                dist.barrier()                                      # (the connections are made lazily: force them now)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return Comm()


def shard_envs(num_envs_global: int, comm: Comm) -> tuple[int, int]:
    """Contiguous env shard [start, start+count) of this rank (rank k owns envs k*N/W..)."""
    w, r = comm.world_size, comm.rank
    base, rem = divmod(num_envs_global, w)
    count = base + (1 if r < rem else 0)
    start = r * base + min(r, rem)
    return start, count


def require_equal_shards(num_envs_global: int, comm: Comm) -> None:
    """The data-parallel update assumes every rank holds the same number of rows: the per-minibatch exchange runs
    ceil(M_local / batch) rounds per rank (unequal counts would leave collectives unmatched and hang the job) and the
    global means (KL, CPO surrogates, gradient scale 1/world) weight every rank equally."""
    if comm.world_size > 1 and num_envs_global % comm.world_size != 0:
        raise ValueError(f"--num-envs {num_envs_global} is not divisible by the number of ranks ({comm.world_size}): "
                         "data-parallel runs need equal env shards")


def dp_reduce_gradient_(comm: Comm, flat_grad: torch.Tensor) -> float:
    """All-reduce(sum) the flat minibatch gradient of all three networks in place and return the
    scale (1/world_size) that turns the sum of per-rank MEAN-loss gradients into the gradient of the
    mean over the global minibatch (the L2 terms are identical on every rank, so they average to
    themselves).  The joint clip_grad_norm_ and Adam then run identically on every rank."""
    comm.all_reduce_sum_(flat_grad)
    return 1.0 / comm.world_size


def dp_mean_scalar(comm: Comm, value: float, device=None) -> float:
    """Mean over ranks of a host scalar (EpCost for the Lagrange update, ppo_lag.py:272)."""
    if comm.world_size == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    comm.all_reduce_sum_(t)
    return float(t.item()) / comm.world_size


def dp_epoch_stat(comm: Comm, logger, key: str, device=None) -> float:
    """logger.get_stats(key) for a data-parallel job (ppo_lag.py:272, logger.py:369-373): the mean over ALL values stored
    this epoch on ANY rank -- (sum, count) are all-reduced, so a rank that finished no episode contributes nothing instead
    of a NaN / 0.0 placeholder.  Keeps the reference's two quirks job-wide: 0.0 until the key has been in a dumped header
    on some rank, NaN when it has been but no rank stored a value this epoch."""
    if comm.world_size == 1:
        return float(logger.get_stats(key))
    vals = logger.epoch_dict.get(key, []) if key in logger.log_headers else []
    t = torch.tensor([float(np.sum(vals)) if len(vals) else 0.0, float(len(vals)), 1.0 if key in logger.log_headers else 0.0],
                     dtype=torch.float64, device=device)
    comm.all_reduce_sum_(t)
    s, n, seen = (float(x) for x in t.tolist())
    if seen == 0.0:
        return 0.0
    return s / n if n > 0 else float("nan")


def adv_stats_from_sums(sums: torch.Tensor):
    """(mean_r, unbiased std_r, mean_c) from all-reduced [sum r, sum r^2, sum c, n] (buffer.py:154-160)."""
    s = sums.double().cpu()
    n = float(s[3])
    mean_r = float(s[0]) / n
    var = max((float(s[1]) - float(s[0]) ** 2 / n) / (n - 1.0), 0.0)
    return mean_r, var ** 0.5, float(s[2]) / n


class PeerExchange:
    """Exchange regions of the in-kernel data-parallel update (include/safepo_hip.h section (e), csrc/update.hip
    xr_allreduce): one uncached device region per rank, mapped into every peer through an IPC handle, so the
    persistent update kernel can push / reduce the per-minibatch gradient over xGMI inside the step instead of
    leaving the kernel for an RCCL call 327 680 times per epoch.  `step` is the tag base (optimiser steps taken
    through the regions), advanced by the caller identically on every rank."""

    MAX_WORLD = 8

    def __init__(self, comm: Comm, device: torch.device):
        import ctypes
        from safepo import _abi
        self._abi, self._ct = _abi, ctypes
        self.lib = _abi.load()
        self.comm, self.device = comm, device
        self.world, self.rank = comm.world_size, comm.rank
        if not (2 <= self.world <= self.MAX_WORLD):
            raise _abi.SpoError(f"PeerExchange: world size {self.world} outside [2, {self.MAX_WORLD}]")
        self.step = 0
        self.own = ctypes.c_void_p()
        self.regions = (ctypes.c_void_p * self.MAX_WORLD)()
        self._opened = []
        handle = (ctypes.c_ubyte * 64)()
        ok = 1
        try:
            _abi.check(self.lib.spo_p2p_alloc(ctypes.byref(self.own), handle), "spo_p2p_alloc")
        except _abi.SpoError as e:
            ok, self._why = 0, str(e)
        # handles travel through the job's own process group (GPU tensors for nccl, CPU tensors for gloo)
        coll_dev = device if dist.get_backend(comm.group) == "nccl" else torch.device("cpu")
        mine = torch.tensor(list(bytes(handle)) + [ok], dtype=torch.uint8, device=coll_dev)
        gathered = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(gathered, mine, group=comm.group)
        gathered = [g.cpu() for g in gathered]
        if all(int(g[64]) == 1 for g in gathered):
            for r in range(self.world):
                if r == self.rank:
                    self.regions[r] = self.own
                    continue
                buf = (ctypes.c_ubyte * 64)(*gathered[r][:64].tolist())
                ptr = ctypes.c_void_p()
                try:
                    _abi.check(self.lib.spo_p2p_open(buf, ctypes.byref(ptr)), "spo_p2p_open")
                    self.regions[r] = ptr
                    self._opened.append(ptr)
                except _abi.SpoError as e:
                    ok, self._why = 0, str(e)
                    break
        else:
            ok = 0
        flag = torch.tensor([float(ok)], device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=comm.group)
        self.ok = bool(flag.item() == 1.0)
        self._result = torch.zeros(2, dtype=torch.int32, device=device)

    def selftest(self, iters: int = 200) -> bool:
        """Run the exchange protocol on known patterns on every rank; True only if every rank saw every value right."""
        if not self.ok:
            return False
        _abi = self._abi
        import time
        torch.cuda.synchronize(self.device)
        self.comm.barrier()                      # the ranks enter together, so the wall time below is the exchange itself
        t0 = time.perf_counter()
        _abi.check(self.lib.spo_p2p_selftest(self.rank, self.world, self.regions, self.step & 0xFFFFFFFF, iters,
                                             _abi.ptr(self._result), _abi.stream_ptr()), "spo_p2p_selftest")
        self.step += iters
        bad, timeout = self._result.tolist()
        self.last_selftest_s = time.perf_counter() - t0
        # an exchange that works but crawls (peers time-sliced on a shared GPU: one scheduler quantum per hand-off) is as
        # useless as one that fails: `iters` rounds take a few milliseconds when every rank's kernel is resident
        crawling = self.last_selftest_s > max(2.0, 0.01 * iters)
        coll_dev = self.device if dist.get_backend(self.comm.group) == "nccl" else torch.device("cpu")
        flag = torch.tensor([1.0 if (bad == 0 and timeout == 0 and not crawling) else 0.0], device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.comm.group)
        self.ok = bool(flag.item() == 1.0)
        self.last_selftest = (bad, timeout)
        return self.ok

    FORM_NAMES = {0: "two-phase (packed reduce-scatter + all-gather, four-wave kernel)",
                  1: "recursive doubling of packed words (four-wave kernel)",
                  2: "one-shot all-to-all with flags (helper waves)",
                  3: "recursive doubling of packed words (helper waves)",
                  4: "row-split kernel, tagged words behind the row groups' L2 hand-off: one-hand-off all-to-all at 2 ranks, "
                     "reduce-scatter + all-gather (two hand-offs, every poll one batch) from 4 ranks on"}

    def autotune(self, obs_dim: int, act_dim: int, steps: int = 192, with_rccl: bool = True) -> dict:
        """Start-up selection of the per-minibatch exchange on THIS topology: every form of the in-kernel
        exchange that exists at this world size runs `steps` real minibatch steps of the persistent update kernel (64 rows,
        this policy shape, synthetic rows, scratch parameters) -- and the kernel / RCCL all-reduce / kernel form a few --,
        the slowest rank's time per step is taken (max-reduce), and the fastest form is pinned on every rank
        (spo_p2p_select_form).  One-GPU loopback cannot rank forms whose cost is link parallelism (the one-shot all-to-all
        uses all 7 xGMI links at once there and one memory system here), so the ranking is measured where the job runs.
        Returns {form name: us per step} (inf: the form timed out) with "chosen"; self.prefer_rccl says whether the RCCL form won."""
        import time
        _abi, lib, dev, comm = self._abi, self.lib, self.device, self.comm
        D, A, B = int(obs_dim), int(act_dim), 64
        M = steps * B
        P = int(lib.spo_param_count(D, A))
        g = torch.Generator(device=dev).manual_seed(4321 + self.rank)
        f32 = dict(dtype=torch.float32, device=dev)
        theta0 = torch.randn(P, generator=g, **f32) * 0.1
        comm.broadcast_(theta0, 0)                          # replicas start identical, as in a job
        obs, act = torch.randn(M, D, generator=g, **f32), torch.randn(M, A, generator=g, **f32)
        logp, adv = torch.full((M,), -float(A), **f32), torch.randn(M, generator=g, **f32)
        tgt_r, tgt_c = torch.randn(M, generator=g, **f32), torch.rand(M, generator=g, **f32)
        perm = torch.arange(M, dtype=torch.int32, device=dev)
        losses = torch.empty((steps, 3), **f32)
        sync_ws = torch.zeros(32, dtype=torch.int64, device=dev)
        cfg = _abi.PpoCfg(obs_dim=D, act_dim=A, batch=B, use_critic_norm=1, use_value_coefficient=0, clip=0.2, max_grad_norm=40.0,
                          lr_actor=3e-4, lr_critic=3e-4, beta1=0.9, beta2=0.999, adam_eps=1e-8, l2_coef=0.001)
        coll_dev = dev if dist.get_backend(comm.group) == "nccl" else torch.device("cpu")

        def slowest(us: float) -> float:
            t = torch.tensor([us if us == us and us != float("inf") else 1e30], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=comm.group)
            v = float(t.item())
            return float("inf") if v >= 1e29 else v

        table = {}
        only = os.environ.get("SPO_P2P_AUTOTUNE_FORMS")          # (development: restrict the table, e.g. "1,4")
        for form in (1, 0, 2, 3, 4):
            if not lib.spo_p2p_form_valid(form, self.world) or (only and str(form) not in only.split(",")):
                continue
            _abi.check(lib.spo_p2p_select_form(form), "spo_p2p_select_form")
            us, failed = float("inf"), False
            for rep in range(2):                            # first launch: lazy per-kernel set-up (untimed); an error in EITHER disqualifies
                theta, m, v = theta0.clone(), torch.zeros(P, **f32), torch.zeros(P, **f32)
                torch.cuda.synchronize(dev)
                comm.barrier()
                t0 = time.perf_counter()
                rc = lib.spo_ppo_lag_update_iter_dp(_abi.ptr(theta), _abi.ptr(m), _abi.ptr(v), 0, _abi.ptr(obs), _abi.ptr(act), _abi.ptr(logp),
                                                    _abi.ptr(tgt_r), _abi.ptr(tgt_c), _abi.ptr(adv), _abi.ptr(perm), M, cfg, _abi.ptr(losses),
                                                    _abi.ptr(sync_ws), self.rank, self.world, self.regions, self.step & 0xFFFFFFFF,
                                                    _abi.stream_ptr())
                self.step += steps
                torch.cuda.synchronize(dev)
                us = (time.perf_counter() - t0) / steps * 1e6
                err = int(sync_ws[8].item()) & 0xFFFFFFFF
                if rc or err:                               # a peer never answered inside the bounded waits: not a candidate
                    sync_ws[8] = 0
                    failed = True
            table[form] = slowest(float("inf") if failed else us)
        rccl_us = None
        if with_rccl:
            # (the same treatment as the in-kernel forms -- an untimed warm-up pass for the lazy RCCL / kernel set-up, then a
            # timed pass of comparable length)
            n_warm, n_r = 16, min(steps, 96)
            theta, m, v = theta0.clone(), torch.zeros(P, **f32), torch.zeros(P, **f32)
            fg, l3 = torch.zeros(P, **f32), torch.zeros(3, **f32)
            idx = perm[:B]

            def rccl_steps(n, k0):
                for k in range(k0, k0 + n):
                    lib.spo_ppo_lag_grad(_abi.ptr(theta), _abi.ptr(obs), _abi.ptr(act), _abi.ptr(logp), _abi.ptr(tgt_r), _abi.ptr(tgt_c),
                                         _abi.ptr(adv), _abi.ptr(idx), B, B, cfg, _abi.ptr(fg), _abi.ptr(l3), _abi.stream_ptr())
                    comm.all_reduce_sum_(fg)
                    lib.spo_clip_adam(_abi.ptr(theta), _abi.ptr(m), _abi.ptr(v), _abi.ptr(fg), k, 1.0 / self.world, cfg, _abi.stream_ptr())
            rccl_steps(n_warm, 0)
            torch.cuda.synchronize(dev)
            comm.barrier()
            t0 = time.perf_counter()
            rccl_steps(n_r, n_warm)
            torch.cuda.synchronize(dev)
            rccl_us = slowest((time.perf_counter() - t0) / n_r * 1e6)
        live = {f: u for f, u in table.items() if u != float("inf")}
        best = min(live, key=lambda f: (live[f], f)) if live else None
        # Reproducibility: the forms add the ranks' gradients in different orders, so WHICH form runs decides the bits of a
        # seeded run.  The deterministic default policy (csrc/update.hip xr_form: the row-split form at 2 / 4 / 8 ranks; on the four-wave kernel doubling at 2 / 4 ranks, packed two-phase otherwise)
        # is kept unless the measured winner beats it by more than `margin` (timing noise does not flip the choice between two
        # near-equal forms); the choice is logged on rank 0 and lands in the run's config through engine.exchange_autotune.
        _abi.check(lib.spo_p2p_select_form(-1), "spo_p2p_select_form")
        default_form = int(lib.spo_p2p_current_form(self.world))
        margin = float(os.environ.get("SPO_P2P_AUTOTUNE_MARGIN", "0.05"))
        if best is not None and default_form in live and live[default_form] <= live[best] * (1.0 + margin):
            best = default_form
        _abi.check(lib.spo_p2p_select_form(-1 if best is None else best), "spo_p2p_select_form")
        self.form = best
        self.prefer_rccl = best is None or (rccl_us is not None and rccl_us < live[best])
        backend = dist.get_backend(comm.group)
        out = {self.FORM_NAMES[f]: (round(u, 2) if u != float("inf") else None) for f, u in table.items()}
        if rccl_us is not None:
            out[f"kernel / {'RCCL' if backend == 'nccl' else backend} all-reduce / kernel"] = round(rccl_us, 2)
        out["chosen"] = ("kernel / all-reduce / kernel" if self.prefer_rccl else self.FORM_NAMES[best])
        out["unit"] = f"us per 64-row minibatch step, slowest rank, {steps} steps per in-kernel form"
        out["default_form"] = self.FORM_NAMES.get(default_form, str(default_form))
        out["margin"] = margin
        self.autotune_table = out
        if self.rank == 0:
            import sys
            print(f"[safepo] per-minibatch gradient exchange: {out['chosen']} (start-up auto-tune over {len(table)} in-kernel form(s)"
                  f"{' + the all-reduce form' if rccl_us is not None else ''}; SPO_P2P_AUTOTUNE=0 keeps the default policy)", file=sys.stderr)
        return out

    def close(self) -> None:
        for ptr in self._opened:
            self.lib.spo_p2p_close(ptr)
        self._opened = []
        if self.own:
            self.lib.spo_p2p_free(self.own)
            self.own = self._ct.c_void_p()
        self.ok = False

    @classmethod
    def try_create(cls, comm: Comm, device, verbose: bool = True):
        """Regions + self-test, or None (with a note on stderr) when peer mapping is not available: the caller then
        uses the kernel / RCCL all-reduce / kernel form of the step."""
        import sys
        if comm.world_size < 2 or comm.world_size > cls.MAX_WORLD or os.environ.get("SPO_P2P", "1") == "0":
            return None
        px = cls(comm, torch.device(device))
        px.form, px.prefer_rccl, px.autotune_table = None, False, None
        if px.ok and px.selftest():
            return px
        if verbose and comm.rank == 0:
            print(f"[safepo] in-kernel gradient exchange unavailable ({getattr(px, '_why', getattr(px, 'last_selftest', ''))}); "
                  "using the RCCL all-reduce form of the minibatch step", file=sys.stderr)
        px.close()
        return None
