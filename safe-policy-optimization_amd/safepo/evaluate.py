"""Evaluate saved runs: reference safepo/evaluate.py (eval_single_agent, eval_multi_agent, single_runs_eval,
benchmark_eval over a <benchmark-dir>/<task>/<algo>/<seed-run>/ tree).

Reads what the training scripts of this package write -- config.json, torch_save/model{itr}.pt (actor state_dict),
state{itr}.pkl ({"Normalizer": obs_rms}), models_seed{seed}/actor_agent{i}.pt -- which are the reference's formats, so
runs of either side can be evaluated.  The policy steps run through spo_policy_step / spo_ma_forward on the GPU.
"""
from __future__ import annotations

import argparse
import json
import os
from collections import deque

import numpy as np
import torch

from safepo.utils.config import multi_agent_goal_tasks, multi_agent_velocity_map

MULTI_AGENT_ALGOS = ("macpo", "mappo", "mappolag", "happo")


def _latest(directory: str, suffix: str):
    """Newest checkpoint by epoch number (the reference sorts the names as strings, which puts model9 after model10)."""
    names = [n for n in os.listdir(directory) if n.endswith(suffix)]
    if not names:
        return None
    num = lambda n: int("".join(ch for ch in n if ch.isdigit()) or -1)
    return os.path.join(directory, max(names, key=num))


def eval_single_agent(eval_dir: str, eval_episodes: int, device: str = "cuda:0"):
    """evaluate.py:32-90: deterministic episodes of env 0 with the last saved actor and observation normaliser."""
    import joblib
    from safepo.common.env import make_sa_mujoco_env
    from safepo.common.model import ActorVCritic
    config = json.load(open(os.path.join(eval_dir, "config.json")))
    env_id = config["task"] if "task" in config else config["env_name"]
    model_path = _latest(os.path.join(eval_dir, "torch_save"), ".pt")
    norm_path = _latest(eval_dir, ".pkl")
    kw = {"device": device} if env_id.startswith("Synth") and not env_id.startswith("SynthHost") else {}
    eval_env, obs_space, act_space = make_sa_mujoco_env(num_envs=config["num_envs"], env_id=env_id, seed=None, **kw)
    model = ActorVCritic(obs_dim=obs_space.shape[0], act_dim=act_space.shape[0],
                         hidden_sizes=config.get("hidden_sizes", [64, 64])).to(device)
    model.actor.load_state_dict(torch.load(model_path, map_location=device))
    if norm_path is not None:
        norm = joblib.load(open(norm_path, "rb")).get("Normalizer")
        if norm is not None:
            if hasattr(eval_env, "load_obs_rms"):
                eval_env.load_obs_rms(norm, freeze=True)      # device normaliser: restore the statistics and stop updating them
            else:
                eval_env.obs_rms = norm
    rews, costs, lens = deque(maxlen=50), deque(maxlen=50), deque(maxlen=50)
    host_env = not getattr(eval_env, "is_device_env", False)          # host envs take numpy actions
    first = lambda v: float(v[0]) if not torch.is_tensor(v) else float(v[0].item())
    for _ in range(eval_episodes):
        done = False
        obs, _ = eval_env.reset()
        ep_r = ep_c = ep_l = 0.0
        while not done:
            obs = torch.as_tensor(obs, dtype=torch.float32, device=device)
            with torch.no_grad():
                act, _, _, _ = model.step(obs, deterministic=True)
            obs, reward, cost, terminated, truncated, _ = eval_env.step(act.detach().cpu().numpy() if host_env else act)
            ep_r += first(reward); ep_c += first(cost); ep_l += 1
            done = bool(first(terminated)) or bool(first(truncated))
        rews.append(ep_r); costs.append(ep_c); lens.append(ep_l)
    return sum(rews) / len(rews), sum(costs) / len(costs)


def eval_multi_agent(eval_dir: str, eval_episodes: int):
    """evaluate.py:93-139: rebuild the algorithm's Runner on the saved per-agent actors and run its eval()."""
    import importlib
    from safepo.common.env import make_ma_synth_env
    config = json.load(open(os.path.join(eval_dir, "config.json")))
    algo = config["algorithm_name"]
    if algo not in MULTI_AGENT_ALGOS:
        raise NotImplementedError(algo)
    env_name = config["env_name"]
    if not str(env_name).startswith("Synth"):
        raise NotImplementedError("this build has no simulator (safety_gymnasium is not installed here); evaluate Synth* runs, "
                                  "or build the vector env yourself and call Runner(...).eval()")
    cfg = dict(config)
    cfg["n_rollout_threads"] = cfg["n_eval_rollout_threads"]
    cfg["log_dir"] = os.path.join(eval_dir, "eval")
    eval_env = make_ma_synth_env(cfg, seed=int(np.random.randint(0, 1000)))
    Runner = importlib.import_module(f"safepo.multi_agent.{algo}").Runner
    runner = Runner(vec_env=eval_env, vec_eval_env=eval_env, config=cfg,
                    model_dir=os.path.join(eval_dir, f"models_seed{config['seed']}"))
    return runner.eval(eval_episodes)


def single_runs_eval(eval_dir: str, eval_episodes: int):
    config = json.load(open(os.path.join(eval_dir, "config.json")))
    env = config["task"] if "task" in config else config["env_name"]
    multi = env in multi_agent_velocity_map or env in multi_agent_goal_tasks or config.get("algorithm_name") in MULTI_AGENT_ALGOS
    return eval_multi_agent(eval_dir, eval_episodes) if multi else eval_single_agent(eval_dir, eval_episodes)


def benchmark_eval(argv=None):
    """evaluate.py:152-187: every <benchmark-dir>/<env>/<algo>/<seed-run>; one line per (env, algo) in eval_result.txt."""
    p = argparse.ArgumentParser()
    p.add_argument("--benchmark-dir", type=str, default="", help="the directory of the evaluation")
    p.add_argument("--eval-episodes", type=int, default=3, help="the number of episodes to evaluate")
    p.add_argument("--save-dir", type=str, default=None, help="the directory to save the evaluation result")
    args = p.parse_args(argv)
    save_dir = args.save_dir if args.save_dir is not None else args.benchmark_dir.replace("runs", "results")
    os.makedirs(save_dir, exist_ok=True)
    results = {}
    for env in sorted(os.listdir(args.benchmark_dir)):
        for algo in sorted(os.listdir(os.path.join(args.benchmark_dir, env))):
            algo_path = os.path.join(args.benchmark_dir, env, algo)
            print(f"Start evaluating {algo} in {env}")
            rewards, costs = [], []
            for seed in sorted(os.listdir(algo_path)):
                r, c = single_runs_eval(os.path.join(algo_path, seed), args.eval_episodes)
                rewards.append(r); costs.append(c)
            rm, rs, cm, cs = (round(float(v), 2) for v in (np.mean(rewards), np.std(rewards), np.mean(costs), np.std(costs)))
            line = (f"After {args.eval_episodes} episodes evaluation, the {algo} in {env} evaluation reward: {rm}±{rs}, "
                    f"cost: {cm}±{cs}")
            print(line + f", the result is saved in {save_dir}/eval_result.txt")
            with open(os.path.join(save_dir, "eval_result.txt"), "a") as f:
                f.write(line + " \n")
            results[(env, algo)] = (rm, rs, cm, cs)
    return results


if __name__ == "__main__":
    benchmark_eval()
