"""Shared driver of the trust-region single-agent scripts natural_pg, trpo, rcpo, trpo_lag.

In the reference each is a ~520-line copy of the same file that differs in two places:
    natural_pg.py:356-381   step = x * sqrt(2*delta/xHx), applied directly
    trpo.py:384-428         the same direction + backtracking line search (surrogate improves, KL <= delta)
    rcpo.py:325-326         natural_pg on advantage = (adv_r - lambda*adv_c)/(lambda+1), Lagrange updated per epoch
    trpo_lag.py:326-327     trpo on the same mixed advantage
All of them reuse the CPO kernels (spo_cpo_surrogate_grad / spo_cpo_fvp / spo_cpo_linesearch_eval /
spo_critic_fit_iter) through safepo.single_agent.cpo.CPOEngine.
"""
from __future__ import annotations

import os
import random
import time

import numpy as np
import torch

from safepo.common.env import make_sa_mujoco_env
from safepo.common.lagrange import Lagrange
from safepo.common.logger import EpochLogger
from safepo.common.model import ActorVCritic
from safepo.parallel import dp_epoch_stat, init_from_env, require_equal_shards, shard_envs
from safepo.single_agent.cpo import _to_dev, make_engine
from safepo.utils.config import isaac_gym_map


def run(args, cfg_env, default_cfg: dict, line_search: bool, use_lagrange: bool):
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if args.device == "cpu":
        raise RuntimeError("this build runs the trust-region hot path on a ROCm GPU only (--device cuda); no CPU fallback")
    comm = init_from_env()
    local_rank = int(os.environ.get("LOCAL_RANK", args.device_id))
    device = torch.device(f"cuda:{local_rank if comm.world_size > 1 else args.device_id}")
    torch.cuda.set_device(device)
    if args.task in isaac_gym_map:
        raise NotImplementedError("Isaac Gym tasks (isaac_gym_specific_cfg) are not part of this build")
    config = dict(default_cfg)
    config.update(getattr(args, "cfg_override", None) or {})
    require_equal_shards(args.num_envs, comm)
    _, n_local = shard_envs(args.num_envs, comm)          # one process per GPU: a contiguous shard of the envs each
    env, obs_space, act_space = make_sa_mujoco_env(num_envs=n_local, env_id=args.task, seed=args.seed + 1000 * comm.rank,
                                                   device=device, **(getattr(args, "env_kwargs", None) or {}))
    device_env = getattr(env, "is_device_env", False)
    steps_per_epoch = config.get("steps_per_epoch", args.steps_per_epoch)
    total_steps = config.get("total_steps", args.total_steps)
    local_steps_per_epoch = steps_per_epoch // args.num_envs
    epochs = total_steps // steps_per_epoch
    policy = ActorVCritic(obs_dim=obs_space.shape[0], act_dim=act_space.shape[0],
                          hidden_sizes=config["hidden_sizes"]).to(device)
    comm.broadcast_(policy.theta, 0)                       # identical replicas
    engine = make_engine(policy, n_local, local_steps_per_epoch, config, device, comm=comm)
    lagrange = Lagrange(cost_limit=args.cost_limit, lagrangian_multiplier_init=args.lagrangian_multiplier_init,
                        lagrangian_multiplier_lr=args.lagrangian_multiplier_lr) if use_lagrange else None
    dict_args = dict(vars(args))
    dict_args.update(config)
    is_root = comm.rank == 0
    logger = EpochLogger(log_dir=args.log_dir if is_root else os.path.join(args.log_dir, f"rank{comm.rank}"),
                         seed=str(args.seed), verbose=is_root)
    logger.save_config(dict_args)
    logger.setup_torch_saver(policy.actor)
    logger.log("Start with training.")
    # device envs that normalise observations hand the normaliser over: statistics merge + normalisation then run inside
    # the collect kernel (spo_policy_step_norm), one pass over the raw observations
    rms = env.fuse_normalize(True) if hasattr(env, "fuse_normalize") else None
    obs, _ = env.reset()
    obs = _to_dev(obs, device)
    outs = []
    for epoch in range(epochs):
        rollout_start_time = time.time()
        obs = engine.rollout_epoch(env, obs, rms=rms)      # (one HIP-graph replay for capturable device envs)
        engine.drain_episode_events(logger)
        torch.cuda.synchronize(device)
        rollout_end_time = time.time()
        eval_end_time = rollout_end_time

        ep_costs = dp_epoch_stat(comm, logger, "Metrics/EpCost", device)
        lam = None
        if lagrange is not None:
            lagrange.update_lagrange_multiplier(ep_costs)
            lam = lagrange.lagrangian_multiplier
        engine.buffer.compute_gae(lam, comm)
        M = engine.M
        advantage = engine.buffer.adv_mix.reshape(M) if lam is not None else engine.buffer.data["adv_r"].reshape(M)
        out = engine.trust_region_update(advantage, line_search, logger)
        misc = {"Misc/Alpha": out["alpha"], "Misc/FinalStepNorm": out["final_step_norm"], "Misc/xHx": out["xHx"],
                "Misc/gradient_norm": out["gradient_norm"], "Misc/H_inv_g": out["H_inv_g"],
                "Loss/Loss_actor": out["loss_actor"], "Train/KL": out["kl"]}
        if line_search:
            misc["Misc/AcceptanceStep"] = out["acceptance_step"]
        logger.store(**misc)
        fit = engine.critic_fit()
        engine.buffer.reset()
        logger.store(**{"Loss/Loss_reward_critic": fit["loss_r"], "Loss/Loss_cost_critic": fit["loss_c"]})
        torch.cuda.synchronize(device)
        update_end_time = time.time()
        outs.append(out)
        if not logger.logged:
            logger.log_tabular("Metrics/EpRet")
            logger.log_tabular("Metrics/EpCost")
            logger.log_tabular("Metrics/EpLen")
            logger.log_tabular("Train/Epoch", epoch + 1)
            logger.log_tabular("Train/TotalSteps", (epoch + 1) * args.steps_per_epoch)
            logger.log_tabular("Train/KL")
            if lagrange is not None:
                logger.log_tabular("Train/LagragianMultiplier", lagrange.lagrangian_multiplier)
            logger.log_tabular("Loss/Loss_reward_critic")
            logger.log_tabular("Loss/Loss_cost_critic")
            logger.log_tabular("Loss/Loss_actor")
            logger.log_tabular("Time/Rollout", rollout_end_time - rollout_start_time)
            logger.log_tabular("Time/Update", update_end_time - eval_end_time)
            logger.log_tabular("Time/Total", update_end_time - rollout_start_time)
            d = engine.buffer.data
            logger.log_tabular("Value/RewardAdv", d["adv_r"].mean().item())
            logger.log_tabular("Value/CostAdv", d["adv_c"].mean().item())
            for k in ("Misc/Alpha", "Misc/FinalStepNorm", "Misc/xHx", "Misc/gradient_norm", "Misc/H_inv_g"):
                logger.log_tabular(k)
            if line_search:
                logger.log_tabular("Misc/AcceptanceStep")
            logger.dump_tabular()
            if (epoch + 1) % 100 == 0 or epoch == 0:
                logger.torch_save(itr=epoch)
                logger.save_state(state_dict={"Normalizer": getattr(env, "obs_rms", None)}, itr=epoch)
        else:
            for k in list(logger.epoch_dict):
                logger.epoch_dict[k] = []
    logger.close()
    return {"policy": policy, "engine": engine, "outs": outs}
