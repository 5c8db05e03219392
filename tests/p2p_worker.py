"""Worker of tests/test_gpu_parity.py::test_in_kernel_gradient_exchange_two_ranks_one_gpu (not a test module).

Launched as 2 ranks by torch.distributed.run on ONE GPU: both processes use cuda:0, map each other's exchange region
through the IPC handle, and run the data-parallel persistent update kernel (spo_ppo_lag_update_iter_dp).  Host
collectives go through gloo (RCCL refuses two ranks on one device).  Checks, written as JSON by rank 0:
replicas bit-identical; parameters equal to the kernel / all-reduce / kernel form of the same steps; self-test clean."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(out_path: str, M: int, iters: int):
    from safepo import parallel as P
    from safepo.common.engine import PPOLagEngine
    from safepo.common.model import ActorVCritic
    comm = P.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    rank, world = comm.rank, comm.world_size
    D, A = 60, 8
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": iters,
           "max_grad_norm": 40.0}
    g = torch.Generator().manual_seed(1234 + rank)          # every rank has its own shard of rows
    obs, act = torch.randn(M, D, generator=g), torch.randn(M, A, generator=g)
    logp = -A * 0.9 - 0.5 * (act ** 2).sum(-1) + 0.1 * torch.randn(M, generator=g)
    tgt_r, tgt_c, adv = torch.randn(M, generator=g), torch.rand(M, generator=g), torch.randn(M, generator=g)
    perms = [torch.randperm(M, generator=g).to(torch.int32).to(dev) for _ in range(iters)]

    def fresh_engine(use_p2p: bool):
        torch.manual_seed(7)
        pol = ActorVCritic(D, A).to(dev)
        os.environ["SPO_P2P"] = "1" if use_p2p else "0"
        eng = PPOLagEngine(pol, 1, M, cfg, dev, comm=comm)
        b = eng.buffer
        b.data["obs"].copy_(obs.view(1, M, D)); b.data["act"].copy_(act.view(1, M, A))
        b.data["log_prob"].copy_(logp.view(1, M)); b.data["target_value_r"].copy_(tgt_r.view(1, M))
        b.data["target_value_c"].copy_(tgt_c.view(1, M)); b.adv_mix.copy_(adv.view(1, M))
        return pol, eng

    res = {"world": world, "M": M, "iters": iters}
    pol, eng = fresh_engine(True)
    res["p2p_created"] = eng.p2p is not None
    if eng.p2p is not None:
        res["selftest"] = list(eng.p2p.last_selftest)
        theta0 = pol.theta.detach().cpu().clone()
        losses = []
        torch.cuda.synchronize()
        comm.barrier()
        t0 = time.perf_counter()
        for it in range(iters):
            losses.append(eng.learning_iter(perms[it]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.check_sync_error()
        n_steps = iters * ((M + 63) // 64)
        res["p2p_us_per_step"] = 1e6 * dt / n_steps
        theta_p2p = pol.theta.detach().cpu()
        loss_p2p = torch.cat(losses, 0).cpu()
        gathered = [torch.empty_like(theta_p2p) for _ in range(world)]
        dist.all_gather(gathered, theta_p2p)
        res["replicas_identical"] = all(torch.equal(gathered[0], x) for x in gathered[1:])
        res["finite"] = bool(torch.isfinite(theta_p2p).all())
        res["moved"] = float((theta_p2p - theta0).abs().max())
        # a peer timeout seen by ONE rank must raise on EVERY rank (max-reduced error word), after which the engine drops
        # to the RCCL form with the replicas made identical again
        from safepo._abi import SpoError
        if rank == world - 1:
            eng.sync_ws[8] = 2
        try:
            eng.check_sync_error()
            raised = 0.0
        except SpoError:
            raised = 1.0
        flag = torch.tensor([raised])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res["timeout_raises_on_every_rank"] = bool(flag.item() == 1.0)
        eng.drop_peer_exchange()
        assert eng.p2p is None
        eng.learning_iter(perms[0])
        eng.check_sync_error()
        th_fb = pol.theta.detach().cpu()
        gathered = [torch.empty_like(th_fb) for _ in range(world)]
        dist.all_gather(gathered, th_fb)
        res["fallback_replicas_identical"] = all(torch.equal(gathered[0], x) for x in gathered[1:]) and bool(torch.isfinite(th_fb).all())
        # the same steps through kernel / all-reduce / kernel
        pol2, eng2 = fresh_engine(False)
        assert eng2.p2p is None
        losses2 = []
        torch.cuda.synchronize()
        comm.barrier()
        t0 = time.perf_counter()
        for it in range(iters):
            losses2.append(eng2.learning_iter(perms[it]))
        torch.cuda.synchronize()
        res["allreduce_us_per_step"] = 1e6 * (time.perf_counter() - t0) / n_steps
        theta_ar = pol2.theta.detach().cpu()
        diff = (theta_p2p - theta_ar).abs()
        res["max_abs_diff_vs_allreduce_form"] = float(diff.max())
        res["frac_outside_1e-5"] = float((diff > 1e-5 + 1e-4 * theta_ar.abs()).float().mean())
        res["loss_max_abs_diff"] = float((loss_p2p - torch.cat(losses2, 0).cpu()).abs().max())
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)
    comm.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 2)
