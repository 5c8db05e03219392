"""Command line of the single-agent scripts: same flags, defaults and types as the reference
`single_agent_args()` (safepo/utils/config.py:144-191).  Isaac Gym tasks are recognised by name
(isaac_gym_map) but need the isaacgym package, exactly like the reference."""
from __future__ import annotations

import argparse

isaac_gym_map = {
    "ShadowHandOver_Safe_finger": "shadow_hand_over_safe_finger",
    "ShadowHandOver_Safe_joint": "shadow_hand_over_safe_joint",
    "ShadowHandCatchOver2Underarm_Safe_finger": "shadow_hand_catch_over_2_underarm_safe_finger",
    "ShadowHandCatchOver2Underarm_Safe_joint": "shadow_hand_catch_over_2_underarm_safe_joint",
    "FreightFrankaCloseDrawer": "freight_franka_close_drawer",
    "FreightFrankaPickAndPlace": "freight_franka_pick_and_place",
}


def strtobool(val: str) -> int:
    """distutils.util.strtobool (removed in Python 3.12) -- same accepted spellings."""
    v = str(val).lower()
    if v in ("y", "yes", "t", "true", "on", "1"):
        return 1
    if v in ("n", "no", "f", "false", "off", "0"):
        return 0
    raise ValueError(f"invalid truth value {val!r}")


def _bool(x):
    return bool(strtobool(x))


SINGLE_AGENT_FLAGS = [
    ("--seed", int, 0, "Random seed"),
    ("--use-eval", _bool, False, "Use evaluation environment for testing"),
    ("--task", str, "SafetyPointGoal1-v0", "The task to run"),
    ("--num-envs", int, 10, "The number of parallel game environments"),
    ("--experiment", str, "single_agent_exp", "Experiment name"),
    ("--log-dir", str, "../runs", "directory to save agent logs"),
    ("--device", str, "cuda", "The device to run the model on (this build: a ROCm GPU; reference default cpu)"),
    ("--device-id", int, 0, "The device id to run the model on"),
    ("--write-terminal", _bool, True, "Toggles terminal logging"),
    ("--headless", _bool, False, "Toggles headless mode"),
    ("--total-steps", int, 10000000, "Total timesteps of the experiments"),
    ("--steps-per-epoch", int, 20000, "The number of steps to run in each environment per policy rollout"),
    ("--randomize", bool, False, "Wheather to randomize the environments' initial states"),
    ("--cost-limit", float, 25.0, "cost_lim"),
    ("--lagrangian-multiplier-init", float, 0.001, "initial value of lagrangian multiplier"),
    ("--lagrangian-multiplier-lr", float, 0.035, "learning rate of lagrangian multiplier"),
]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="RL Policy")
    for name, typ, default, help_ in SINGLE_AGENT_FLAGS:
        parser.add_argument(name, type=typ, default=default, help=help_)
    return parser


def single_agent_args(argv=None):
    """-> (args, cfg_env).  cfg_env is {} for non-Isaac tasks (reference config.py:172-191)."""
    args = build_parser().parse_args(argv)
    if args.task in isaac_gym_map:
        raise Exception("Please install isaacgym to run Isaac Gym tasks!")
    return args, {}


multi_agent_velocity_map = {
    "Safety2x4AntVelocity-v0": {"agent_conf": "2x4", "scenario": "Ant"},
    "Safety4x2AntVelocity-v0": {"agent_conf": "4x2", "scenario": "Ant"},
    "Safety2x3HalfCheetahVelocity-v0": {"agent_conf": "2x3", "scenario": "HalfCheetah"},
    "Safety6x1HalfCheetahVelocity-v0": {"agent_conf": "6x1", "scenario": "HalfCheetah"},
    "Safety3x1HopperVelocity-v0": {"agent_conf": "3x1", "scenario": "Hopper"},
    "Safety2x3Walker2dVelocity-v0": {"agent_conf": "2x3", "scenario": "Walker2d"},
    "Safety2x1SwimmerVelocity-v0": {"agent_conf": "2x1", "scenario": "Swimmer"},
    "Safety9|8HumanoidVelocity-v0": {"agent_conf": "9|8", "scenario": "Humanoid"},
}


multi_agent_goal_tasks = [f"Safety{robot}MultiGoal{level}-v0" for robot in ("Point", "Ant") for level in (0, 1, 2)]


def multi_agent_args(algo: str, argv=None):
    """CLI of the multi-agent scripts (reference safepo/utils/config.py:194-278): same flags; the training config
    starts from the algorithm's defaults (the reference's marl_cfg/<algo>/config.yaml, here a dict in the algorithm
    module), with the `mamujoco` overrides for MuJoCo-style tasks -- the Synth* tasks count as such."""
    import importlib
    import time
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--use-eval", type=_bool, default=False)
    p.add_argument("--task", type=str, default="SynthMultiAgent-v0")
    p.add_argument("--agent-conf", type=str, default="2x4")
    p.add_argument("--scenario", type=str, default="Ant")
    p.add_argument("--experiment", type=str, default="Base")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--model-dir", type=str, default="")
    p.add_argument("--cost-limit", type=float, default=25.0)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--device-id", type=int, default=0)
    p.add_argument("--write-terminal", type=_bool, default=True)
    p.add_argument("--headless", type=_bool, default=False)
    p.add_argument("--total-steps", type=int, default=None)
    p.add_argument("--num-envs", type=int, default=None)
    p.add_argument("--randomize", type=bool, default=False)
    args = p.parse_args(argv)
    if args.task in isaac_gym_map:
        raise NotImplementedError("Isaac Gym tasks need the isaacgym package (not part of this build)")
    mod = importlib.import_module(f"safepo.multi_agent.{algo}")
    cfg_train = dict(mod.default_cfg)
    if args.task in multi_agent_velocity_map or args.task in multi_agent_goal_tasks or args.task.startswith("Synth"):
        cfg_train.update(mod.mamujoco_cfg)
        if args.task in multi_agent_velocity_map:
            args.agent_conf = multi_agent_velocity_map[args.task]["agent_conf"]
            args.scenario = multi_agent_velocity_map[args.task]["scenario"]
    cfg_train["use_eval"] = args.use_eval
    cfg_train["cost_limit"] = args.cost_limit
    cfg_train["algorithm_name"] = algo
    cfg_train["device"] = args.device + ":" + str(args.device_id)
    cfg_train["env_name"] = args.task
    cfg_train["seed"] = args.seed
    if args.total_steps:
        cfg_train["num_env_steps"] = args.total_steps
    if args.num_envs:
        cfg_train["n_rollout_threads"] = args.num_envs
        cfg_train["n_eval_rollout_threads"] = args.num_envs
    relpath = "-".join(["-".join(["seed", str(args.seed).zfill(3)]), time.strftime("%Y-%m-%d-%H-%M-%S")])
    cfg_train["log_dir"] = "../runs/" + args.experiment + "/" + args.task + "/" + algo + "/" + relpath
    return args, {}, cfg_train


def run_as_script(main, script_file):
    """The `if __name__ == "__main__":` block that every single-agent script of the reference repeats verbatim (e.g.
    ppo_lag.py:392-409): parse the reference's flags, build log_dir = <log_dir>/<experiment>/<task>/<algo>/seed-XXX-<time>, and run
    main() with stdout / stderr redirected into seed<N>_terminal.log / seed<N>_error.log unless --write-terminal is set."""
    import os
    import sys
    import time
    args, cfg_env = single_agent_args()
    relpath = time.strftime("%Y-%m-%d-%H-%M-%S")
    subfolder = "-".join(["seed", str(args.seed).zfill(3)])
    relpath = "-".join([subfolder, relpath])
    algo = os.path.basename(script_file).split(".")[0]
    args.log_dir = os.path.join(args.log_dir, args.experiment, args.task, algo, relpath)
    if not args.write_terminal:
        os.makedirs(args.log_dir, exist_ok=True)
        with open(os.path.join(args.log_dir, f"seed{args.seed}_terminal.log"), "w", encoding="utf-8") as f_out, \
                open(os.path.join(args.log_dir, f"seed{args.seed}_error.log"), "w", encoding="utf-8") as f_err:
            sys.stdout, sys.stderr = f_out, f_err
            return main(args, cfg_env)
    return main(args, cfg_env)
