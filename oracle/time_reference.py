"""TEST INFRASTRUCTURE -- time the UNMODIFIED reference ppo_lag.main() and the oracle port on the same bounded
sample in the build container (the reference cannot travel to the GPU box; bench.py's cpu_baseline uses the port).

    python oracle/time_reference.py [num_envs] [T]

Prints env-steps/s for both and their ratio, to show the port is a representative stand-in."""
from __future__ import annotations

import os
import sys
import time
from collections import deque

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402
from oracle.synth_env import Space, SynthEnv  # noqa: E402


def time_reference(n, T):
    P = ref_shim.load_reference("ppo_lag")
    ref_shim.set_env_factory(P, lambda num_envs, env_id, seed: (SynthEnv(num_envs, 60, 8, seed=0, trunc_len=64), Space(60), Space(8)))
    real_logger = P.EpochLogger
    rows = {}

    class Log(real_logger):
        def __init__(self, *a, **k):
            k["use_tensorboard"] = False
            k["verbose"] = False
            super().__init__(*a, **k)

        def dump_tabular(self):
            rows.update(self.log_current_row)
            super().dump_tabular()
    P.EpochLogger = Log
    cfg_saved = dict(P.default_cfg)
    P.default_cfg["target_kl"] = float("inf")
    args = ref_shim.make_args(num_envs=n, steps_per_epoch=n * T, total_steps=n * T, log_dir="/tmp/oracle_runs/time/task/run")
    t0 = time.time()
    try:
        P.main(args, {})
    finally:
        P.EpochLogger = real_logger
        P.default_cfg.clear(); P.default_cfg.update(cfg_saved)
    wall = time.time() - t0
    return n * T / (rows["Time/Rollout"] + rows["Time/Update"]), rows["Time/Rollout"], rows["Time/Update"], wall


def time_port(n, T):
    from oracle import restatement as R
    torch.set_num_threads(4)
    torch.manual_seed(0)
    cfg = {"gamma": 0.99, "target_kl": float("inf"), "batch_size": 64, "learning_iters": 40}
    env = SynthEnv(n, 60, 8, seed=0, p_term=0.0, trunc_len=64)
    pol = R.OraclePolicy(60, 8)
    upd = R.PPOLagUpdater(pol, epochs=1)
    lag = R.OracleLagrange(25.0, 0.001, 0.035)
    obs, _ = env.reset()
    timers = {}
    R.ppo_lag_epoch_port(env, pol, upd, lag, torch.as_tensor(obs), n, T, R.StatsLog(),
                         (deque(maxlen=50), deque(maxlen=50), deque(maxlen=50)),
                         (np.zeros(n), np.zeros(n), np.zeros(n)), cfg, timers)
    return n * T / (timers["rollout"] + timers["update"]), timers["rollout"], timers["update"]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    r = time_reference(n, T)
    p = time_port(n, T)
    print(f"reference main(): {r[0]:.1f} env-steps/s (rollout {r[1]:.2f}s update {r[2]:.2f}s)")
    print(f"oracle port     : {p[0]:.1f} env-steps/s (rollout {p[1]:.2f}s update {p[2]:.2f}s)")
    print(f"port/reference  : {p[0] / r[0]:.3f}")
