"""MAPPO-L (BASELINE config 5 shape: 4 agents, obs 48, mamujoco overrides) on one MI355X: env-steps/s of
collect -> insert -> fused GAE/PopArt -> HAPPO-sequential training, with a phase breakdown.  Not the headline metric
(bench.py); a measurement of the f3 row."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8192)
    ap.add_argument("--episode-length", type=int, default=64)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--agents", type=int, default=4)
    a = ap.parse_args()
    print(json.dumps(run(a)))


def run(a, comm=None, dev=None):
    """One warm-up and a.episodes timed episodes of the MAPPO-L Runner on the config-5 shape; returns the result record
    (bench.py puts it into its JSON line as `config5_mappolag`).
    comm (safepo.parallel.Comm, world > 1): BASELINE config 5 is "num_envs=8192, 8 x MI355X" -- the a.threads rollout threads are
    SPLIT over the ranks (strong scaling of that row), every rank runs its shard of the Runner, every mean of the update is over
    the global batch and the three flat gradients / loss scalars / PopArt sums are all-reduced per full-batch step
    (safepo/multi_agent/mappolag.py); the time is the slowest rank's (barrier + synchronize on both sides)."""
    from safepo.multi_agent import mappolag
    from safepo.common.env import SynthMultiAgentEnv
    from safepo.parallel import Comm
    comm = comm or Comm()
    world = comm.world_size
    dev = torch.device("cuda:0") if dev is None else torch.device(dev)
    if a.threads % world:
        raise ValueError(f"{a.threads} rollout threads do not divide over {world} ranks")
    threads_local = a.threads // world
    cfg = dict(mappolag.default_cfg)
    cfg.update(mappolag.mamujoco_cfg)
    cfg.update(device=str(dev), n_rollout_threads=threads_local, episode_length=a.episode_length, hidden_size=a.hidden,
               log_dir=f"/tmp/ma_bench_run_{os.getpid()}", seed=0, env_name="SynthMultiAgent-v0", use_eval=False)
    env = SynthMultiAgentEnv(threads_local, num_agents=a.agents, obs_dim=48, act_dim=6, trunc_len=a.episode_length, device=dev,
                             seed=1000 * comm.rank)
    r = mappolag.Runner(env, None, cfg, comm=comm)
    r.logger.verbose = False
    r.warmup()
    ph = {"collect": 0.0, "compute": 0.0, "train": 0.0}

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            comm.barrier()

    def episode(timed):
        t0 = time.perf_counter()
        for step in range(a.episode_length):
            values, actions, lps, rnn, rnn_c, cps, rnn_k = r.collect(step)
            obs, share_obs, rewards, costs, dones, infos, _ = env.step(actions)
            r.insert((obs, share_obs, rewards, costs, dones, infos, values, actions, lps, rnn, rnn_c, cps, rnn_k, costs.mean()))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r.compute()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r.train()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if timed:
            ph["collect"] += t1 - t0; ph["compute"] += t2 - t1; ph["train"] += t3 - t2
    episode(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.episodes):
        episode(True)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:                                    # max over ranks of the timed region and of every phase
        t = torch.tensor([dt, ph["collect"], ph["compute"], ph["train"]], dtype=torch.float64, device=dev)
        comm.all_reduce_max_(t)
        dt, ph["collect"], ph["compute"], ph["train"] = (float(x) for x in t.tolist())
    steps = a.threads * a.episode_length * a.episodes
    rows = a.threads * a.episode_length
    # GEMM flops per epoch: per agent, 3 networks, learning_iters full-batch fwd+bwd (3x fwd) + collect/old/new evaluations
    H, D, S, A = a.hidden, 48, env.share_dim, 6
    per_row_actor = 2 * (D * H + 2 * H * H + H * A)
    per_row_critic = 2 * (S * H + 2 * H * H + H)
    train_flops = a.agents * cfg["learning_iters"] * rows * 3 * (per_row_actor + 2 * per_row_critic)
    return {"workload": f"mappolag synthetic {a.agents} agents obs 48 act 6, {a.threads} rollout threads"
                        + (f" split over {world} ranks ({threads_local} each)" if world > 1 else "")
                        + f" x {a.episode_length} steps, hidden {H}, "
                        f"learning_iters {cfg['learning_iters']}, num_mini_batch {cfg['num_mini_batch']}",
            "n_gpus": world, "scaling": "strong" if world > 1 else None,
            "parallelism": (f"dp{world} over rollout threads, {_backend_name()} all-reduce of the three flat gradients, loss scalars "
                            "and PopArt sums per full-batch step" if world > 1 else "single GPU"),
            "env_steps_per_s": round(steps / dt, 1), "episodes_timed": a.episodes, "s_per_epoch": round(dt / a.episodes, 4),
            "phases_s_per_epoch": {k: round(v / a.episodes, 4) for k, v in ph.items()},
            "train_gemm_tflops": round(train_flops / (ph["train"] / a.episodes) / 1e12, 2)}


def _backend_name() -> str:
    """The collective backend that actually ran ("nccl" is RCCL on ROCm; gloo in the one-GPU development mode)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return "no"
    b = dist.get_backend()
    return "RCCL" if b == "nccl" else f"{b} (host)"


if __name__ == "__main__":
    main()
