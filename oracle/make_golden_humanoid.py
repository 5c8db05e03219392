"""TEST INFRASTRUCTURE -- tests/golden/*_trace_humanoid.npz: complete runs of the UNMODIFIED reference's main() (ppo_lag, focops,
cup, cpo) at HumanoidVelocity's dims -- ActorVCritic(376, 17), the shape of the reference's default sweep
(safepo/single_agent/benchmark.py:5-22) that the round-5 feature-split kernels serve -- on the seeded host SynthEnv, recorded
exactly like the 60 / 8 traces of oracle/make_golden.py (same recorder, same keys); plus ppo_lag / cpo at Car-class dims (72 / 2).

    python oracle/make_golden_humanoid.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import make_golden as G  # noqa: E402

if __name__ == "__main__":
    os.makedirs(G.OUT, exist_ok=True)
    torch.set_num_threads(1)
    env_kw = dict(obs_dim=376, act_dim=17, p_term=0.03, p_cost=0.3, trunc_len=20)
    G.golden_trace("ppo_lag", "ppo_lag_trace_humanoid.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw,
                   cfg_over={"learning_iters": 4, "target_kl": 0.004},
                   args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
    G.golden_trace("focops", "focops_trace_humanoid.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw,
                   cfg_over={"learning_iters": 4, "target_kl": 0.0006},
                   args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
    G.golden_trace("cup", "cup_trace_humanoid.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw,
                   cfg_over={"learning_iters": 4, "target_kl": 0.002},
                   args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
    # Car-class dims of the same sweep (72 observations, 2 actions: the 128-wide instantiation of the LDS-resident kernels;
    # CPO's actor on the wide kernels with the critic fit on the persistent two-critic kernel)
    car_kw = dict(obs_dim=72, act_dim=2, p_term=0.03, p_cost=0.3, trunc_len=20)
    G.golden_trace("ppo_lag", "ppo_lag_trace_car.npz", num_envs=4, T=48, epochs=2, env_kw=car_kw,
                   cfg_over={"learning_iters": 4, "target_kl": 0.004},
                   args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
    G.golden_trace("cpo", "cpo_trace_car.npz", num_envs=4, T=48, epochs=2, env_kw=car_kw,
                   cfg_over={"learning_iters": 2, "batch_size": 64}, args_over={"cost_limit": 3.0})
    # two more of the sweep's second-order scripts (smaller runs: 2 envs): PCPO's projection step and TRPO-Lagrangian
    G.golden_trace("pcpo", "pcpo_trace_humanoid.npz", num_envs=2, T=48, epochs=2, env_kw=env_kw,
                   cfg_over={"learning_iters": 2, "batch_size": 64}, args_over={"cost_limit": 3.0})
    G.golden_trace("trpo_lag", "trpo_lag_trace_humanoid.npz", num_envs=2, T=48, epochs=2, env_kw=env_kw,
                   cfg_over={"learning_iters": 2, "batch_size": 64},
                   args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
    # the reference's default critic-fit minibatch of 128 rows (192 rows per epoch = one full + one 64-row minibatch): the
    # feature-split critic fit's two-chunk path
    G.golden_trace("cpo", "cpo_trace_humanoid_b128.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw,
                   cfg_over={"learning_iters": 3, "batch_size": 128}, args_over={"cost_limit": 3.0})
    G.golden_trace("cpo", "cpo_trace_humanoid.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw,
                   cfg_over={"learning_iters": 2, "batch_size": 64}, args_over={"cost_limit": 3.0})
