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
