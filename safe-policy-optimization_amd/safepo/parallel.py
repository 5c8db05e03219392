"""Data-parallel plumbing: one process per GPU, env shards, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference has no distributed layer at all.  What the sharded path has
to exchange: per epoch the advantage statistics (4 doubles) and the EpCost mean;
per minibatch step the flat gradient of all three networks (24 850 fp32 = 99 KB, latency-bound);
per learning iteration the KL sum (1 double) so every rank stops at the same iteration."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class Comm:
    """Thin wrapper so single-process runs need no process group."""

    def __init__(self, group=None):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = group
        self.world_size = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

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
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
    return Comm()


def shard_envs(num_envs_global: int, comm: Comm) -> tuple[int, int]:
    """Contiguous env shard [start, start+count) of this rank (rank k owns envs k*N/W..)."""
    w, r = comm.world_size, comm.rank
    base, rem = divmod(num_envs_global, w)
    count = base + (1 if r < rem else 0)
    start = r * base + min(r, rem)
    return start, count


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


def adv_stats_from_sums(sums: torch.Tensor):
    """(mean_r, unbiased std_r, mean_c) from all-reduced [sum r, sum r^2, sum c, n] (buffer.py:154-160)."""
    s = sums.double().cpu()
    n = float(s[3])
    mean_r = float(s[0]) / n
    var = max((float(s[1]) - float(s[0]) ** 2 / n) / (n - 1.0), 0.0)
    return mean_r, var ** 0.5, float(s[2]) / n
