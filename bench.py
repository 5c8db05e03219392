"""bench.py -- env-steps/sec (collect + GAE + update) of the PPO-Lagrangian hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one epoch of BASELINE.json config[1] per GPU: 4096 synthetic envs x 128 steps
(obs 60, act 8), device-resident env, collect -> fused reward/cost GAE -> 40 learning iterations
of 8192 minibatches of 64 (default_cfg of the reference, ppo_lag.py:45-52; KL early stopping
disabled so the work per step is fixed).  N > 1: one process per GPU, 4096 envs per
rank (weak scaling); the per-minibatch gradient all-reduce runs inside the persistent update kernel over
IPC-mapped peer regions (xGMI), or through RCCL between kernels when peer mapping is unavailable.
Prints ONE JSON line (rank 0) with the driver's fields plus `roofline` (GAE scan kernel, HBM
bound, 33 algorithmic bytes per (env, step)) and `cpu_baseline` (oracle port of the reference
loop timed on host cores, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GAE_BYTES_PER_ELEM = 33.0       # 4 f32 in + 1 u8 mask + 4 f32 out
GAE_REPS = 100                   # back-to-back launches per timed epoch (event timing cannot resolve one ~10 us launch)


def cpu_baseline(sample_envs: int, T: int, threads: int = 4):
    """Oracle port of the reference epoch (oracle/restatement.ppo_lag_epoch_port: per-env Python
    store loop, per-path Python GAE, DataLoader minibatches, torch CPU) on a bounded sample of the
    same workload: `sample_envs` envs x T steps, full default_cfg (batch 64, 40 iterations)."""
    from collections import deque
    from oracle import restatement as R
    from oracle.synth_env import SynthEnv
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    cfg = {"gamma": 0.99, "target_kl": float("inf"), "batch_size": 64, "learning_iters": 40}
    env = SynthEnv(sample_envs, 60, 8, seed=0, p_term=0.0, trunc_len=64)
    pol = R.OraclePolicy(60, 8)
    upd = R.PPOLagUpdater(pol, epochs=1)
    lag = R.OracleLagrange(25.0, 0.001, 0.035)
    stats = R.StatsLog()
    obs, _ = env.reset()
    obs = torch.as_tensor(obs)
    timers = {}
    t0 = time.time()
    R.ppo_lag_epoch_port(env, pol, upd, lag, obs, sample_envs, T, stats,
                         (deque(maxlen=50), deque(maxlen=50), deque(maxlen=50)),
                         (np.zeros(sample_envs), np.zeros(sample_envs), np.zeros(sample_envs)), cfg, timers)
    wall = time.time() - t0
    steps = sample_envs * T
    return {"value": steps / (timers["rollout"] + timers["update"]), "unit": "env-steps/s", "cores": threads,
            "kind": "port",
            "sample": f"1 epoch of {sample_envs} envs x {T} steps (={steps} env-steps), batch 64, 40 learning iters, "
                      f"torch CPU {threads} threads; rollout {timers['rollout']:.2f}s update {timers['update']:.2f}s wall {wall:.2f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs PER GPU")
    ap.add_argument("--num-steps", type=int, default=128)
    ap.add_argument("--learning-iters", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-envs", type=int, default=96)   # ~15 s of CPU work on the GPU box host
    ap.add_argument("--stream-envs", type=int, default=262144, help="extra GAE roofline point that streams from HBM")
    ap.add_argument("--algo", choices=["ppo_lag", "cpo"], default="ppo_lag",
                    help="cpo = BASELINE config 3 (not the headline metric; single GPU)")
    a = ap.parse_args()

    from safepo import _abi
    from safepo.common.engine import PPOLagEngine
    from safepo.common.env import SynthDeviceEnv
    from safepo.common.model import ActorVCritic
    from safepo.parallel import init_from_env

    # SPO_BENCH_ONE_GPU=1 (development aid): all ranks share cuda:0 with gloo for the host collectives, to exercise the
    # N > 1 code path -- including the in-kernel exchange through IPC-mapped regions -- on a single-GPU box.
    one_gpu = os.environ.get("SPO_BENCH_ONE_GPU", "0") == "1"
    comm = init_from_env(backend="gloo" if one_gpu else None)
    world = comm.world_size
    assert world == max(a.gpus, 1) or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    local_rank = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    N, T, D, A = a.num_envs, a.num_steps, 60, 8
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": float("inf"), "batch_size": 64,
           "learning_iters": a.learning_iters, "max_grad_norm": 40.0}
    policy = ActorVCritic(D, A).to(dev)
    comm.broadcast_(policy.theta, 0)
    if a.algo == "cpo":
        from safepo.single_agent.cpo import CPOEngine, default_cfg as cpo_cfg
        cfg = dict(cpo_cfg)
        eng = CPOEngine(policy, N, T, cfg, dev, comm=comm)
    else:
        eng = PPOLagEngine(policy, N, T, cfg, dev, comm=comm)
    env = SynthDeviceEnv(N, D, A, seed=1234 + comm.rank, p_term=0.0, p_cost=0.1, trunc_len=64, device=dev)
    obs, _ = env.reset()
    lam = 0.001
    gae_events = []

    def epoch(timed: bool):
        nonlocal obs
        t0 = time.time()
        for t in range(T):
            act = eng.collect_step(t, obs)
            nobs, rew, cost, term, trunc, info = env.step(act)
            eng.post_step(t, nobs, rew, cost, term, trunc, info["final_observation"])
            obs = nobs
        n_ep = eng.drain_episode_events(None)
        torch.cuda.synchronize(dev)
        t1 = time.time()
        if a.algo == "cpo":
            eng.buffer.compute_gae(None, comm)
            pu = eng.policy_update(-1.0)
            fit = eng.critic_fit()
            eng.buffer.reset()
            out = {"stop_iter": pu["acceptance_step"], "kl": pu["kl"]}
        else:
            out = eng.update(lam)
        if timed:   # GAE scan of THIS epoch's buffer: graph of GAE_REPS launches between HIP events (~0.5 ms)
            gae_events.append(eng.buffer.time_scan(GAE_REPS))
        torch.cuda.synchronize(dev)
        t2 = time.time()
        return t1 - t0, t2 - t1, out, n_ep

    # N > 1: the in-kernel gradient exchange has a self-test at start-up; should a peer still time out in a full epoch
    # (bounded spins, the error is max-reduced so every rank sees it), all ranks drop to the RCCL form together and the
    # warm-up starts over.  One guard epoch runs even with --warmup 0 so the timed region never hits this first.
    guard = max(a.warmup, 1) if (world > 1 and getattr(eng, "p2p", None) is not None) else a.warmup
    done_w = 0
    while done_w < guard:
        try:
            if os.environ.pop("SPO_BENCH_INJECT_PEER_TIMEOUT", "0") == "1" and comm.rank == world - 1:
                eng.sync_ws[8] = 2          # development aid: exercise the fallback below on one rank's error word
            epoch(False)
            done_w += 1
        except _abi.SpoError as e:
            if world == 1 or getattr(eng, "p2p", None) is None:
                raise
            if comm.rank == 0:
                print(f"[bench] {e}; falling back to the RCCL form of the minibatch step", file=sys.stderr)
            eng.drop_peer_exchange()
            eng.buffer.reset()
            obs, _ = env.reset()
            done_w = 0
    comm.barrier()
    torch.cuda.synchronize(dev)
    t_start = time.time()
    roll = upd = 0.0
    last = None
    for _ in range(a.steps):
        r, u, last, n_ep = epoch(True)
        roll += r
        upd += u
    comm.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.time() - t_start
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    comm.all_reduce_max_(tmax)
    elapsed = float(tmax.item())

    if comm.rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    total_env_steps = world * N * T * a.steps
    value = total_env_steps / elapsed
    n_mb = (N * T + 63) // 64
    if a.algo == "cpo":
        n_mb = (N * T + 127) // 128
    gae_ms = gae_events
    gae_avg_s = (sum(gae_ms) / len(gae_ms)) if gae_ms else float("nan")
    seg_ends = int(eng.buffer.seg_end.sum().item())
    gae_bytes = GAE_BYTES_PER_ELEM * N * T + 8.0 * seg_ends
    achieved = gae_bytes / gae_avg_s / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_gae_pmc.json")
    if os.path.exists(pmc_path):
        try:
            # PMC counters were collected (separate rocprofv3 passes) for the 4096 x 128 launch only
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch_n4096") if (N, T) == (4096, 128) else None
        except Exception:
            traffic = None
    roofline = {"kernel": "gae_kernel<4,32> (spo_gae_fused)", "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_launch": gae_bytes, "avg_launch_us": round(gae_avg_s * 1e6, 2), "launches_timed": len(gae_ms) * GAE_REPS,
                "note": "HIP events around a hipGraph of 100 back-to-back launches, once per timed epoch, inside the timed region "
                        "(includes the ~1.5 us kernel boundary of each launch)"}

    # extra roofline points (untimed, after the run): 32 768 envs (138 MB) and a buffer that cannot sit in
    # the 256 MiB Infinity Cache
    def scan_point(Ns, reps):
        from safepo.common.buffer import VectorizedOnPolicyBuffer
        from safepo.common.engine import _Space
        big = VectorizedOnPolicyBuffer(_Space(1), _Space(1), size=T, num_envs=Ns, device=dev)
        for k in ("reward", "cost", "value_r", "value_c"):
            big.data[k].normal_()
        big.seg_end[:, T - 1] = 1
        big.seg_end[:, T // 2 - 1] = 1
        big.compute_gae(None)
        t_s = big.time_scan(reps)
        b = GAE_BYTES_PER_ELEM * Ns * T + 8.0 * 2 * Ns
        del big
        return {"num_envs": Ns, "bytes_per_launch": b, "avg_launch_us": round(t_s * 1e6, 1),
                "achieved": round(b / t_s / 1e9, 1), "unit": "GB/s", "frac": round(b / t_s / 1e9 / HBM_PEAK_GBS, 4)}
    stream, mid = None, None
    try:
        mid = scan_point(32768, 20)
        stream = scan_point(a.stream_envs, 10)
    except Exception as e:  # pragma: no cover
        stream = {"error": str(e)[:200]}

    # the reference-faithful epoch (KL early stopping at default_cfg's target_kl = 0.02), untimed extra
    faithful = None
    if world == 1 and a.algo == "ppo_lag":
        cfg["target_kl"] = 0.02
        r_f, u_f, out_f, _ = epoch(False)
        cfg["target_kl"] = float("inf")
        faithful = {"target_kl": 0.02, "stop_iter": out_f["stop_iter"], "kl": out_f["kl"], "s_per_epoch": round(r_f + u_f, 4),
                    "env_steps_per_s": round(N * T / (r_f + u_f), 1)}

    cpu = None
    if world == 1 and not a.no_cpu_baseline and a.algo == "ppo_lag":
        cpu = cpu_baseline(a.cpu_sample_envs, T)

    line = {
        "metric": "env-steps/sec (collect+GAE+update) at num_envs=4096, 1/2/4/8 GPU",
        "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"ppo_lag synthetic env (obs=60, act=8), num_envs={N} per GPU, num_steps={T}, "
                                f"default_cfg batch 64 x {a.learning_iters} learning iters (target_kl=inf: all iters run), "
                                f"device-resident env") if a.algo == "ppo_lag" else
                               (f"cpo synthetic env (obs=60, act=8), num_envs={N}, num_steps={T}, default_cfg "
                                f"(15 CG iters, 33 FVPs, line search, critic fit batch 128 x 10 iters), device-resident env"),
                   "global_envs": world * N,
                   "parallelism": (f"dp{world} over num_envs, per-minibatch gradient all-reduce "
                                   + ("inside the persistent update kernel over IPC-mapped peer regions (xGMI)"
                                      if getattr(eng, "p2p", None) is not None else "via RCCL between kernels"))
                   if world > 1 else "single GPU",
                   "minibatch_steps_per_epoch": n_mb * a.learning_iters},
        "roofline": roofline,
        "roofline_32768_envs": mid,
        "roofline_hbm_streaming": stream,
        "early_stopping_epoch": faithful,
        # the kernel that owns 99 % of the GPU time is not HBM- but latency/matrix-bound: 327 680 strictly sequential
        # optimiser steps, each at least 368 v_mfma_f32_16x16x4_f32 (32 cycles each) per wave on one CU per network
        "update_kernel": ({"kernel": "ppo_update_kernel<64, persistent>", "bound": "fp32 MFMA issue of one CU per network + per-step latency chain",
                           "us_per_minibatch_step": round(upd / a.steps / (n_mb * a.learning_iters) * 1e6, 3),
                           "mfma_floor_us": round(368 * 32 / 2.4e9 * 1e6, 3),
                           "frac": round((368 * 32 / 2.4e9) / (upd / a.steps / (n_mb * a.learning_iters)), 4),
                           "note": "floor = MFMA issue cycles at 2.4 GHz; DESIGN.md 3.3 has the instruction mix"}
                          if a.algo == "ppo_lag" else None),
        "cpu_baseline": cpu,
        "phases": {"rollout_s_per_epoch": round(roll / a.steps, 4), "update_s_per_epoch": round(upd / a.steps, 4),
                   "update_us_per_minibatch_step": round(upd / a.steps / (n_mb * (a.learning_iters if a.algo == "ppo_lag" else 10)) * 1e6, 3),
                   "stop_iter": last["stop_iter"], "kl": last["kl"], "episodes_per_epoch": n_ep},
    }
    if cpu:
        line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
    print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
