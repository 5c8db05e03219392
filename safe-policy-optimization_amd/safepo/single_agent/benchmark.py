"""Sweep launcher of the single-agent scripts: reference safepo/single_agent/benchmark.py (same flags; one
`python <algo>.py --task ... --seed ... --write-terminal False --experiment ... --total-steps ... --num-envs ...
--steps-per-epoch ...` subprocess per (seed, task, algo), `--workers` at a time).

MI355X specifics: every run is a whole-GPU job (persistent kernels), so runs are dealt round-robin over the visible GPUs
through `--device-id`, and `--workers` defaults to one per GPU.  `--workers 0` only prints the commands.
"""
from __future__ import annotations

import argparse
import os
import shlex
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ALGOS = ["pcpo", "ppo_lag", "cup", "focops", "rcpo", "trpo_lag", "cpo", "cppo_pid"]
NAVI_TASKS = [f"Safety{robot}{task}{level}-v0" for level in (1, 2) for robot in ("Ant", "Car", "Doggo", "Point", "Racecar")
              for task in ("Button", "Circle", "Goal", "Push")]
VEL_TASKS = [f"Safety{robot}Velocity-v1" for robot in ("Ant", "HalfCheetah", "Hopper", "Walker2d", "Swimmer", "Humanoid")]


def visible_gpus() -> int:
    try:
        import torch
        return max(torch.cuda.device_count(), 1)
    except Exception:  # noqa: BLE001 -- the launcher itself needs no GPU
        return 1


def default_tasks():
    try:
        import safety_gymnasium  # noqa: F401
        return NAVI_TASKS + VEL_TASKS
    except ImportError:
        return ["SynthSafe-v0"]           # no simulator in this image: the synthetic device env


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--tasks", nargs="+", default=default_tasks(), help="the ids of the environment to benchmark")
    p.add_argument("--algo", nargs="+", default=ALGOS, help="the ids of the algorithm to benchmark")
    p.add_argument("--num-seeds", type=int, default=3, help="the number of random seeds")
    p.add_argument("--start-seed", type=int, default=0, help="the number of the starting seed")
    p.add_argument("--workers", type=int, default=None, help="concurrent runs (default: one per visible GPU)")
    p.add_argument("--experiment", type=str, default="benchmark", help="name of the experiment")
    p.add_argument("--total-steps", type=int, default=10000000, help="total number of steps")
    p.add_argument("--num-envs", type=int, default=10, help="number of environments to run in parallel")
    p.add_argument("--steps-per-epoch", type=int, default=20000, help="number of steps per epoch")
    return p.parse_args(argv)


def build_commands(args, script_dir: str = HERE, n_gpus: int | None = None):
    n_gpus = n_gpus or visible_gpus()
    commands = []
    for seed in range(args.num_seeds):
        for task in args.tasks:
            total, per_epoch, envs = args.total_steps, args.steps_per_epoch, args.num_envs
            if "Doggo" in task:                     # benchmark.py:99-102: the long-horizon robot gets 10x the budget
                total, per_epoch, envs = 100000000, 200000, 20
            for algo in args.algo:
                commands.append(" ".join([
                    shlex.quote(sys.executable), shlex.quote(os.path.join(script_dir, f"{algo}.py")), "--task", shlex.quote(task),
                    "--seed", str(args.start_seed + 1000 * seed), "--write-terminal", "False", "--experiment",
                    shlex.quote(args.experiment), "--total-steps", str(total), "--num-envs", str(envs), "--steps-per-epoch",
                    str(per_epoch), "--device-id", str(len(commands) % n_gpus)]))
    return commands


def run_experiment(command: str) -> int:
    print(f"running {command}", flush=True)
    rc = subprocess.Popen(shlex.split(command)).wait()
    assert rc == 0, f"exit code {rc}: {command}"
    return rc


def main(argv=None):
    args = parse_args(argv)
    commands = build_commands(args)
    print("======= commands to run:")
    for c in commands:
        print(c)
    workers = visible_gpus() if args.workers is None else args.workers
    if workers <= 0:
        print("not running the experiments because --workers is set to 0; just printing the commands to run")
        return commands
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="safepo-benchmark-worker-") as ex:
        futures = [ex.submit(run_experiment, c) for c in commands]
    for fu in futures:
        fu.result()
    return commands


if __name__ == "__main__":
    main()
