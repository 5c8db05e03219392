"""TEST INFRASTRUCTURE -- time the UNMODIFIED reference ppo_lag.main() / cpo.main() and the oracle port on the same
sample in the build container (the reference cannot travel to the GPU box; bench.py's cpu_baseline times the port there
and quotes the numbers recorded by this script, with their provenance, beside it).

    python oracle/time_reference.py [num_envs] [T] [algo=ppo_lag|cpo] [--no-port] [--json out.json]

Prints env-steps/s for both and their ratio, to show the port is a representative stand-in.  With --json the record
(box, torch version, threads, Time/Rollout, Time/Update as the reference's own logger reports them; SURVEY.md 8(d)
"Timing the reference CPU path") is appended to a JSON list."""
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


def time_reference(n, T, algo="ppo_lag"):
    P = ref_shim.load_reference(algo)
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
    if algo == "ppo_lag":
        P.default_cfg["target_kl"] = float("inf")          # all 40 learning iterations run
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


def time_port_cpo(n, T):
    """The oracle's CPO epoch (R.cpo_epoch_port) on the same sample."""
    from oracle import restatement as R
    torch.set_num_threads(4)
    torch.manual_seed(0)
    cfg = {"gamma": 0.99, "target_kl": 0.01, "batch_size": 128, "learning_iters": 10, "cg_iters": 15}
    env = SynthEnv(n, 60, 8, seed=0, p_term=0.0, trunc_len=64)
    pol = R.OraclePolicy(60, 8)
    obs, _ = env.reset()
    timers = {}
    R.cpo_epoch_port(env, pol, R.CriticFitter(pol), torch.as_tensor(obs), n, T, R.StatsLog(),
                     (deque(maxlen=50), deque(maxlen=50), deque(maxlen=50)),
                     (np.zeros(n), np.zeros(n), np.zeros(n)), cfg, timers=timers)
    return n * T / (timers["rollout"] + timers["update"]), timers["rollout"], timers["update"]


if __name__ == "__main__":
    import json
    import platform
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(argv[0]) if len(argv) > 0 else 32
    T = int(argv[1]) if len(argv) > 1 else 128
    algo = argv[2] if len(argv) > 2 else "ppo_lag"
    r = time_reference(n, T, algo)
    print(f"reference {algo}.main(): {r[0]:.1f} env-steps/s (rollout {r[1]:.2f}s update {r[2]:.2f}s wall {r[3]:.1f}s)", flush=True)
    rec = {"algo": algo, "num_envs": n, "num_steps": T, "env_steps": n * T, "kind": "reference",
           "what": f"unmodified /root/reference safepo.single_agent.{algo}.main via oracle/ref_shim.py on oracle.synth_env.SynthEnv "
                   "(obs 60, act 8, truncation every 64 steps), one epoch, device=cpu, torch.set_num_threads(4) (the reference's own)"
                   + (", target_kl=inf (all 40 learning iterations)" if algo == "ppo_lag" else ""),
           "env_steps_per_s": round(r[0], 1), "time_rollout_s": round(r[1], 2), "time_update_s": round(r[2], 2),
           "wall_s": round(r[3], 1), "threads": 4,
           "box": {"cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
                   "nproc": os.cpu_count(), "where": "build container (no GPU)", "torch": torch.__version__,
                   "python": platform.python_version()},
           "date": time.strftime("%Y-%m-%d %H:%M:%S")}
    if "--no-port" not in sys.argv and algo in ("ppo_lag", "cpo"):
        p = time_port(n, T) if algo == "ppo_lag" else time_port_cpo(n, T)
        print(f"oracle port     : {p[0]:.1f} env-steps/s (rollout {p[1]:.2f}s update {p[2]:.2f}s)")
        print(f"port/reference  : {p[0] / r[0]:.3f}")
        rec["port_env_steps_per_s"] = round(p[0], 1)
        rec["port_over_reference"] = round(p[0] / r[0], 3)
    if "--json" in sys.argv:
        out = sys.argv[sys.argv.index("--json") + 1]
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        recs = json.load(open(out)) if os.path.exists(out) else []
        recs.append(rec)
        json.dump(recs, open(out, "w"), indent=1)
