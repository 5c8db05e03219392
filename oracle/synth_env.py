"""TEST INFRASTRUCTURE (oracle side) -- host-numpy synthetic vector env.

This is NOT part of the product path. It is the seeded transition source that
is fed, identically, to (a) the unmodified reference `main()` (through
oracle/ref_shim.py, only in the build container where /root/reference exists),
(b) the oracle restatement (oracle/restatement.py) and (c) the HIP product path
in parity tests.  SURVEY.md section 8(d) "Synthetic inputs (concrete)".

Env contract consumed by the reference main loop
(/root/reference/safepo/single_agent/ppo_lag.py:151,166,175-186,383):
    reset() -> (obs[N,obs_dim] f32, info)
    step(action[N,act_dim]) -> (obs, reward[N], cost[N], terminated[N],
                               truncated[N], info)
    info["final_observation"]: 1-D object array, per-env ndarray or None
    attribute obs_rms (only pickled into state{itr}.pkl)
"""
from __future__ import annotations

import numpy as np


class Space:
    """Minimal stand-in for gymnasium.spaces.Box: only `.shape` is consumed
    (/root/reference/safepo/common/buffer.py:55-60, ppo_lag.py:100-101)."""

    def __init__(self, dim: int):
        self.shape = (int(dim),)


class SynthEnv:
    """obs' ~ N(0,1), reward ~ N(0,1), cost ~ Bernoulli(p_cost),
    terminated ~ Bernoulli(p_term), truncated = (episode length >= trunc_len).

    All draws come from one numpy Generator so two instances built with the
    same seed yield bit-identical transition streams regardless of the actions
    they are fed (actions are ignored on purpose: parity is on *identical
    transitions*).
    """

    def __init__(self, num_envs: int, obs_dim: int = 60, act_dim: int = 8,
                 seed: int = 0, p_term: float = 0.0, p_cost: float = 0.1,
                 trunc_len: int = 64):
        self.num_envs = int(num_envs)
        self.obs_dim = int(obs_dim)
        self.act_dim = int(act_dim)
        self.p_term = float(p_term)
        self.p_cost = float(p_cost)
        self.trunc_len = int(trunc_len)
        self.rng = np.random.default_rng(seed)
        self.t_env = np.zeros(self.num_envs, dtype=np.int64)
        self.obs_rms = {"mean": np.zeros(obs_dim), "var": np.ones(obs_dim), "count": 1e-4}
        self.single_observation_space = Space(obs_dim)
        self.single_action_space = Space(act_dim)
        self.n_steps = 0

    def reset(self, seed=None):
        self.t_env[:] = 0
        obs = self.rng.standard_normal((self.num_envs, self.obs_dim)).astype(np.float32)
        return obs, {}

    def step(self, action):
        n = self.num_envs
        self.n_steps += 1
        self.t_env += 1
        obs = self.rng.standard_normal((n, self.obs_dim)).astype(np.float32)
        reward = self.rng.standard_normal(n).astype(np.float32)
        cost = (self.rng.random(n) < self.p_cost).astype(np.float32)
        terminated = self.rng.random(n) < self.p_term
        truncated = (self.t_env >= self.trunc_len) & ~terminated
        finished = terminated | truncated
        info = {}
        if finished.any():
            final = np.empty(n, dtype=object)
            reset_obs = self.rng.standard_normal((n, self.obs_dim)).astype(np.float32)
            for i in range(n):
                if finished[i]:
                    final[i] = obs[i].copy()
                    obs[i] = reset_obs[i]
                else:
                    final[i] = None
            info["final_observation"] = final
            self.t_env[finished] = 0
        return obs, reward, cost, terminated, truncated, info
