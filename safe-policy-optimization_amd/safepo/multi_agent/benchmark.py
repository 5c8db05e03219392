"""Sweep launcher of the multi-agent scripts: reference safepo/multi_agent/benchmark.py (same flags; one
`python <algo>.py --task ... --seed ... --write-terminal False --experiment ... --headless True --total-steps ...
--num-envs ...` subprocess per (seed, task, algo)).  Runs are dealt round-robin over the visible GPUs (`--device-id`),
`--workers` defaults to one per GPU, `--workers 0` only prints the commands."""
from __future__ import annotations

import argparse
import os
import shlex
import sys
from concurrent.futures import ThreadPoolExecutor

from safepo.single_agent.benchmark import run_experiment, visible_gpus
from safepo.utils.config import multi_agent_velocity_map

HERE = os.path.dirname(os.path.abspath(__file__))
ALGOS = ["macpo", "mappo", "mappolag", "happo"]


def default_tasks():
    try:
        import safety_gymnasium  # noqa: F401
        return list(multi_agent_velocity_map.keys())
    except ImportError:
        return ["SynthMultiAgent-v0"]     # no simulator in this image: the synthetic device env


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
    return p.parse_args(argv)


def build_commands(args, script_dir: str = HERE, n_gpus: int | None = None):
    n_gpus = n_gpus or visible_gpus()
    commands = []
    for seed in range(args.num_seeds):
        for task in args.tasks:
            for algo in args.algo:
                commands.append(" ".join([
                    shlex.quote(sys.executable), shlex.quote(os.path.join(script_dir, f"{algo}.py")), "--task", shlex.quote(task),
                    "--seed", str(args.start_seed + 1000 * seed), "--write-terminal", "False", "--experiment",
                    shlex.quote(args.experiment), "--headless", "True", "--total-steps", str(args.total_steps), "--num-envs",
                    str(args.num_envs), "--device-id", str(len(commands) % n_gpus)]))
    return commands


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
