"""Per-launch time of the collect step's kernels (4096 envs, obs 60, act 8 by default), each replayed REPS times from one HIP
graph between two events -- what a launch costs inside engine.rollout_epoch's graph (dispatch gap included).
Usage: python tools/collect_kernels_bench.py [num_envs] ; knobs: SPO_STEP_PAR, SPO_OBS_STATS_REG."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "safe-policy-optimization_amd"))

from safepo import _abi
from safepo.common.engine import PPOLagEngine
from safepo.common.env import SynthDeviceEnv
from safepo.common.model import ActorVCritic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T, D, A, REPS = 128, 60, 8, 100
dev = torch.device("cuda:0")
torch.manual_seed(0)
pol = ActorVCritic(D, A).to(dev)
cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": float("inf"), "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
eng = PPOLagEngine(pol, N, T, cfg, dev)
env = SynthDeviceEnv(N, D, A, seed=1, p_term=0.0, p_cost=0.1, trunc_len=64, device=dev, normalize_obs=True)
rms = env.fuse_normalize(True)
obs, _ = env.reset()
eps = torch.randn((N, A), device=dev)
lib = eng.lib
out = {}


def timed(name, fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    out[name] = round(best, 2)


def step_norm():
    eng.buffer.ptr = 0
    rms.pending = True
    eng.collect_step(0, obs, eps=eps, rms=rms)


def step_plain():
    eng.buffer.ptr = 0
    rms.pending = False
    eng.collect_step(0, obs, eps=eps, rms=rms)


def env_step():
    env._advance()


def post():
    eng.buffer.ptr = 0
    eng._events_pending = False         # (timing loop: step 0 over and over, the episode log is not what is measured)
    eng.post_step(0, env.obs, env.reward, env.cost, env.terminated, env.truncated, env.final_obs, rms=None)


def values():
    eng._values_into(env.final_obs, eng.vfinal_r, eng.vfinal_c)


def noop():
    _abi.check(lib.spo_obs_normalize(_abi.ptr(obs[:1]), _abi.ptr(rms.state), 1, D, 0, _abi.stream_ptr()), "n")


timed("policy_step_norm (stats + step)", step_norm)
timed("policy_step (no normaliser)", step_plain)
timed("synth env step", env_step)
timed("values(final_obs) + boundary", post)
timed("values alone", values)
timed("tiny kernel (1 row normalise)", noop)
print(json.dumps({"num_envs": N, "us_per_launch_in_graph": out}))
