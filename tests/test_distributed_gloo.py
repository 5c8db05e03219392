"""world_size-2 gloo test (CPU) of the data-parallel plumbing used by the sharded path: env sharding,
replica broadcast, advantage-statistics merge, per-minibatch flat-gradient all-reduce + identical
clip/Adam on every rank, KL-sum agreement.  Local gradients come from the CPU oracle (checker);
on the GPU the same Comm calls wrap spo_ppo_lag_grad / spo_clip_adam (safepo.common.engine)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _set_flat_grads(policy, flat):
    i = 0
    for p in policy.parameters():
        n = p.numel()
        p.grad = flat[i:i + n].view(p.shape).clone()
        i += n


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle import restatement as R
    from safepo import parallel as P
    comm = P.init_from_env(backend="gloo")
    assert (comm.world_size, comm.rank) == (world, rank)
    res = {}
    # --- env sharding
    res["shard"] = P.shard_envs(7, comm)
    # --- replicas start identical after broadcasting rank 0's flat parameter vector
    torch.manual_seed(100 + rank)
    pol = R.OraclePolicy(12, 3)
    theta = R.flat_params(pol).clone()
    comm.broadcast_(theta, 0)
    i = 0
    for p in pol.parameters():
        p.data.copy_(theta[i:i + p.numel()].view(p.shape)); i += p.numel()
    # --- advantage statistics over env shards == statistics of the global buffer
    g = torch.Generator().manual_seed(5)
    adv_all = torch.randn(8, 16, generator=g) * 3 + 1.5
    cadv_all = torch.rand(8, 16, generator=g)
    lo, cnt = P.shard_envs(8, comm)
    a, c = adv_all[lo:lo + cnt].double(), cadv_all[lo:lo + cnt].double()
    sums = torch.tensor([a.sum(), (a * a).sum(), c.sum(), float(a.numel())], dtype=torch.float64)
    comm.all_reduce_sum_(sums)
    res["adv_stats"] = P.adv_stats_from_sums(sums)
    # --- per-minibatch gradient all-reduce; 3 steps; global batch 64 = 2 x 32 local rows
    M, D, A = 192, 12, 3
    obs, act = torch.randn(M, D, generator=g), torch.randn(M, A, generator=g)
    logp = -3.0 + 0.1 * torch.randn(M, generator=g)
    tr, tc, adv = torch.randn(M, generator=g), torch.rand(M, generator=g), torch.randn(M, generator=g)
    upd = R.PPOLagUpdater(pol, epochs=1, max_grad_norm=0.7)         # clip active
    for step in range(3):
        rows = torch.arange(64 * step + 32 * rank, 64 * step + 32 * rank + 32)
        upd.opt_r.zero_grad(); upd.opt_c.zero_grad(); upd.opt_a.zero_grad()
        total, *_ = R.ppo_lag_losses(pol, obs[rows], act[rows], logp[rows], tr[rows], tc[rows], adv[rows])
        total.backward()
        flat = R.flat_grads(pol).clone()
        scale = P.dp_reduce_gradient_(comm, flat)
        _set_flat_grads(pol, flat * scale)
        torch.nn.utils.clip_grad_norm_(pol.parameters(), 0.7)
        upd.opt_r.step(); upd.opt_c.step(); upd.opt_a.step()
    res["theta"] = R.flat_params(pol).numpy().copy()
    # --- KL: sum over shards / global count; every rank takes the same early-stop decision
    kl_local = torch.tensor([0.3 * (rank + 1) * 96], dtype=torch.float64)
    comm.all_reduce_sum_(kl_local)
    res["kl"] = float(kl_local.item()) / (96 * world)
    res["ep_cost"] = P.dp_mean_scalar(comm, 10.0 * (rank + 1))
    out[rank] = res
    dist.destroy_process_group()


def test_two_rank_data_parallel_matches_single_process():
    for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import restatement as R
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["shard"] == (0, 4) and r1["shard"] == (4, 3)
    # replicas bit-identical after three all-reduced steps
    assert np.array_equal(r0["theta"], r1["theta"])
    # single-process reference: same data, global minibatches of 64, same init as rank 0
    torch.manual_seed(100)
    pol = R.OraclePolicy(12, 3)
    g = torch.Generator().manual_seed(5)
    adv_all = torch.randn(8, 16, generator=g) * 3 + 1.5
    cadv_all = torch.rand(8, 16, generator=g)
    M, D, A = 192, 12, 3
    obs, act = torch.randn(M, D, generator=g), torch.randn(M, A, generator=g)
    logp = -3.0 + 0.1 * torch.randn(M, generator=g)
    tr, tc, adv = torch.randn(M, generator=g), torch.rand(M, generator=g), torch.randn(M, generator=g)
    upd = R.PPOLagUpdater(pol, epochs=1, max_grad_norm=0.7)
    for step in range(3):
        rows = torch.arange(64 * step, 64 * step + 64)
        upd.minibatch_step(obs[rows], act[rows], logp[rows], tr[rows], tc[rows], adv[rows])
    np.testing.assert_allclose(r0["theta"], R.flat_params(pol).numpy(), rtol=2e-5, atol=2e-7)
    mean_r, std_r, mean_c = r0["adv_stats"]
    assert mean_r == pytest.approx(float(adv_all.double().mean()), rel=1e-12)
    assert std_r == pytest.approx(float(adv_all.double().std()), rel=1e-10)
    assert mean_c == pytest.approx(float(cadv_all.double().mean()), rel=1e-12)
    assert r0["adv_stats"] == r1["adv_stats"]
    assert r0["kl"] == r1["kl"] == pytest.approx(0.45)
    assert r0["ep_cost"] == r1["ep_cost"] == 15.0
