"""Shared driver of the first-order single-agent scripts (ppo_lag, ppo, pg, cppo_pid).

The reference keeps one near-identical copy of this loop per script (SURVEY.md section 2 rows 6 and 8:
ppo.py drops the Lagrange multiplier, `advantage = data["adv_r"]` (ppo.py:272); pg.py additionally drops
the ratio clip (pg.py:309); cppo_pid.py swaps Lagrange for PIDLagrangian (cppo_pid.py:39)).  Here the
variants are parameters of one loop over the same HIP kernels:
    multiplier = "adam" | "pid" | None       clip = 0.2 | None (no clipping)
The epoch structure follows safepo/single_agent/ppo_lag.py:159-386.
"""
from __future__ import annotations

import os
import random
import time

import numpy as np
import torch

from safepo.common.engine import PPOLagEngine, WidePPOLagEngine
from safepo.common.env import make_sa_mujoco_env
from safepo.common.lagrange import Lagrange, PIDLagrangian
from safepo.common.logger import EpochLogger
from safepo.common.model import ActorVCritic
from safepo.parallel import dp_epoch_stat, init_from_env, require_equal_shards, shard_envs
from safepo.utils.config import isaac_gym_map

NO_CLIP = 1e30      # clamp(ratio, 1-1e30, 1+1e30) is the identity: pg's unclipped surrogate on the same kernel


def _to_dev(x, dev):
    return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32, device=dev).contiguous()


FOCOPS_LAM, FOCOPS_NU = 1.50, 2.00     # focops.py:44-45
CUP_LAMBDA, CUP_NU = 0.95, 0.20        # cup.py:44-45


def run(args, cfg_env, default_cfg: dict, multiplier: str | None = "adam", clip: float | None = 0.2,
        variant: str = "ppo"):
    """variant: "ppo" (clipped surrogate family), "focops" (focops.py:279-367) or "cup" (cup.py:279-405)."""
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if args.device == "cpu":
        raise RuntimeError("this build runs the PPO-Lagrangian hot path on a ROCm GPU only (--device cuda); "
                           "there is no CPU fallback")
    comm = init_from_env()
    local_rank = int(os.environ.get("LOCAL_RANK", args.device_id))
    device = torch.device(f"cuda:{local_rank if comm.world_size > 1 else args.device_id}")
    torch.cuda.set_device(device)
    if args.task in isaac_gym_map:
        raise NotImplementedError("Isaac Gym tasks (isaac_gym_specific_cfg) are not part of this build")
    config = dict(default_cfg)
    config.update(getattr(args, "cfg_override", None) or {})
    config = {k: v for k, v in config.items() if v is not None}      # an override of None drops the key (e.g. batch_size,
    config["clip"] = NO_CLIP if clip is None else clip              # so that num_mini_batch decides as in isaac_gym_specific_cfg)

    require_equal_shards(args.num_envs, comm)
    _, n_local = shard_envs(args.num_envs, comm)
    env, obs_space, act_space = make_sa_mujoco_env(num_envs=n_local, env_id=args.task,
                                                   seed=args.seed + 1000 * comm.rank, device=device,
                                                   **(getattr(args, "env_kwargs", None) or {}))
    device_env = getattr(env, "is_device_env", False)
    eval_env = None
    if args.use_eval:
        eval_env, _, _ = make_sa_mujoco_env(num_envs=1, env_id=args.task, seed=None, device=device,
                                            **(getattr(args, "env_kwargs", None) or {}))

    steps_per_epoch = config.get("steps_per_epoch", args.steps_per_epoch)
    total_steps = config.get("total_steps", args.total_steps)
    local_steps_per_epoch = steps_per_epoch // args.num_envs
    epochs = total_steps // steps_per_epoch

    policy = ActorVCritic(obs_dim=obs_space.shape[0], act_dim=act_space.shape[0],
                          hidden_sizes=config["hidden_sizes"]).to(device)
    comm.broadcast_(policy.theta, 0)            # identical replicas
    # inside the persistent kernels' envelope (hidden_sizes [64, 64], obs_dim <= 128, act_dim <= 16 -- default_cfg on most tasks):
    # LDS-resident kernels; any other width / dims (HumanoidVelocity: 376 / 17): the wide-network kernels, all variants
    if policy.kernels_supported():
        engine = PPOLagEngine(policy, n_local, local_steps_per_epoch, config, device, comm=comm, lr=3e-4)
    else:
        engine = WidePPOLagEngine(policy, n_local, local_steps_per_epoch, config, device, comm=comm, lr=3e-4)
    if multiplier == "adam":
        upper = {"focops": FOCOPS_NU, "cup": CUP_NU}.get(variant)          # focops.py:136, cup.py:136
        lagrange = Lagrange(cost_limit=args.cost_limit, lagrangian_multiplier_init=args.lagrangian_multiplier_init,
                            lagrangian_multiplier_lr=args.lagrangian_multiplier_lr, lagrangian_upper_bound=upper)
    elif multiplier == "pid":
        lagrange = PIDLagrangian(cost_limit=args.cost_limit, lagrangian_multiplier_init=args.lagrangian_multiplier_init)
    else:
        lagrange = None

    dict_args = dict(vars(args))
    dict_args.update(config)
    if comm.world_size > 1:
        # which form of the per-minibatch gradient exchange this data-parallel run uses (it decides the order the ranks' gradients
        # are added in, hence the bits of a seeded run): recorded with the run
        tab = getattr(engine, "exchange_autotune", None)
        dict_args["gradient_exchange"] = (tab or {}).get("chosen") or ("in-kernel, default policy" if getattr(engine, "p2p", None) is not None
                                                                        else "kernel / all-reduce / kernel")
    is_root = comm.rank == 0
    log_dir = args.log_dir if is_root else os.path.join(args.log_dir, f"rank{comm.rank}")
    logger = EpochLogger(log_dir=log_dir, seed=str(args.seed), verbose=is_root)
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
        # ---- collect (ppo_lag.py:162-234): no host sync inside the loop for device envs
        obs = engine.rollout_epoch(env, obs, rms=rms)      # (one HIP-graph replay for capturable device envs)
        engine.drain_episode_events(logger)
        torch.cuda.synchronize(device)
        rollout_end_time = time.time()

        # ---- evaluation episodes (ppo_lag.py:237-269): deterministic policy on the single eval env.
        # (The reference steps `env` instead of `eval_env` inside this loop, which only works for
        #  --num-envs 1; the eval env is stepped here.)
        if args.use_eval:
            eval_rews, eval_costs, eval_lens = [], [], []
            for _ in range(1 if epoch < epochs - 1 else 10):
                eval_obs, _ = eval_env.reset()
                eval_obs = _to_dev(eval_obs, device).reshape(1, -1)
                eval_rew, eval_cost, eval_len, eval_done = 0.0, 0.0, 0.0, False
                while not eval_done:
                    act, _, _, _ = policy.step(eval_obs, deterministic=True)
                    a_in = act if getattr(eval_env, "is_device_env", False) else act.detach().cpu().numpy()
                    nobs, rew, cst, term, trunc, _ = eval_env.step(a_in)
                    eval_rew += float(np.asarray(rew.cpu() if torch.is_tensor(rew) else rew).reshape(-1)[0])
                    eval_cost += float(np.asarray(cst.cpu() if torch.is_tensor(cst) else cst).reshape(-1)[0])
                    eval_len += 1
                    t0 = np.asarray(term.cpu() if torch.is_tensor(term) else term).reshape(-1)[0]
                    t1 = np.asarray(trunc.cpu() if torch.is_tensor(trunc) else trunc).reshape(-1)[0]
                    eval_done = bool(t0) or bool(t1)
                    eval_obs = _to_dev(nobs, device).reshape(1, -1)
                eval_rews.append(eval_rew); eval_costs.append(eval_cost); eval_lens.append(eval_len)
            logger.store(**{"Metrics/EvalEpRet": np.mean(eval_rews), "Metrics/EvalEpCost": np.mean(eval_costs),
                            "Metrics/EvalEpLen": np.mean(eval_lens)})
        torch.cuda.synchronize(device)
        eval_end_time = time.time()

        # ---- Lagrange multiplier (ppo_lag.py:271-273); EpCost mean is all-reduced over shards
        ep_costs = dp_epoch_stat(comm, logger, "Metrics/EpCost", device)
        if lagrange is not None:
            lagrange.update_lagrange_multiplier(ep_costs)
        # lambda == 0 makes (adv_r - 0*adv_c)/(0+1) == adv_r exactly: ppo / pg (ppo.py:272)
        lam = lagrange.lagrangian_multiplier if lagrange is not None else 0.0

        # ---- policy update (ppo_lag.py:275-350)
        engine.lr_factor = 1.0 - epoch / epochs if epochs > 0 else 1.0      # LinearLR(1 -> 0, total_iters=epochs)
        if variant == "focops":
            out = engine.update_focops(lam, focops_lam=FOCOPS_LAM)
        elif variant == "cup":
            out = engine.update_cup(lam, cup_lambda=CUP_LAMBDA)
        else:
            out = engine.update(lam)
        torch.cuda.synchronize(device)
        update_end_time = time.time()
        next_lr = 3e-4 * (1.0 - min(epoch + 1, epochs) / epochs)
        logger.store(**{"Loss/Loss_reward_critic": out["loss_r"], "Loss/Loss_cost_critic": out["loss_c"],
                        "Loss/Loss_actor": out["loss_pi"]})
        timings.append((rollout_end_time - rollout_start_time, update_end_time - eval_end_time))
        if not logger.logged:
            logger.log_tabular("Metrics/EpRet")
            logger.log_tabular("Metrics/EpCost")
            logger.log_tabular("Metrics/EpLen")
            if args.use_eval:
                logger.log_tabular("Metrics/EvalEpRet")
                logger.log_tabular("Metrics/EvalEpCost")
                logger.log_tabular("Metrics/EvalEpLen")
            logger.log_tabular("Train/Epoch", epoch + 1)
            logger.log_tabular("Train/TotalSteps", (epoch + 1) * args.steps_per_epoch)
            logger.log_tabular("Train/StopIter", out["stop_iter"])
            if variant == "cup":
                logger.log_tabular("Train/SeconStageStopIter", out["second_stage_stop_iter"])   # (sic) cup.py:419
            logger.log_tabular("Train/KL", out["kl"])
            if lagrange is not None:
                logger.log_tabular("Train/LagragianMultiplier", lagrange.lagrangian_multiplier)
            logger.log_tabular("Train/LR", next_lr)
            logger.log_tabular("Loss/Loss_reward_critic")
            logger.log_tabular("Loss/Loss_cost_critic")
            logger.log_tabular("Loss/Loss_actor")
            logger.log_tabular("Time/Rollout", rollout_end_time - rollout_start_time)
            if args.use_eval:
                logger.log_tabular("Time/Eval", eval_end_time - rollout_end_time)
            logger.log_tabular("Time/Update", update_end_time - eval_end_time)
            logger.log_tabular("Time/Total", update_end_time - rollout_start_time)
            stats = engine.buffer.stats.cpu()
            d = engine.buffer.data
            logger.log_tabular("Value/RewardAdv", d["adv_r"].mean().item())
            logger.log_tabular("Value/CostAdv", d["adv_c"].mean().item())
            logger.dump_tabular()
            if is_root and ((epoch + 1) % 100 == 0 or epoch == 0):
                logger.torch_save(itr=epoch)
                logger.save_state(state_dict={"Normalizer": getattr(env, "obs_rms", None)}, itr=epoch)
        else:
            logger.epoch_dict["Loss/Loss_reward_critic"] = []
            logger.epoch_dict["Loss/Loss_cost_critic"] = []
            logger.epoch_dict["Loss/Loss_actor"] = []
            for k in ("Metrics/EvalEpRet", "Metrics/EvalEpCost", "Metrics/EvalEpLen"):
                if k in logger.epoch_dict:
                    logger.epoch_dict[k] = []
    logger.close()
    return {"timings": timings, "policy": policy, "engine": engine}


