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


class SynthMAEnv:
    """CPU multi-agent vector env with the interface the reference MAPPO-L Runner consumes (TEST INFRASTRUCTURE:
    drives the unmodified reference Runner for tests/golden/ma_runner_trace.npz).  All agents of a thread finish
    together every `trunc_len` steps."""

    def __init__(self, num_envs, num_agents=3, obs_dim=10, act_dim=2, share_dim=14, seed=0, p_cost=0.3, trunc_len=6):
        import torch
        self.torch = torch
        self.num_envs, self.num_agents, self.obs_dim, self.act_dim, self.share_dim = num_envs, num_agents, obs_dim, act_dim, share_dim
        self.p_cost, self.trunc_len, self.t = p_cost, trunc_len, 0
        self.gen = torch.Generator().manual_seed(seed)
        self.W = torch.randn(num_agents, obs_dim, act_dim, generator=torch.Generator().manual_seed(99)) / obs_dim ** 0.5
        self.observation_space = [Space(obs_dim) for _ in range(num_agents)]
        self.share_observation_space = [Space(share_dim) for _ in range(num_agents)]
        self.action_space = [Space(act_dim) for _ in range(num_agents)]

    def _draw(self):
        t = self.torch
        obs = t.randn(self.num_envs, self.num_agents, self.obs_dim, generator=self.gen)
        share = obs.reshape(self.num_envs, -1)[:, :self.share_dim].unsqueeze(1).expand(-1, self.num_agents, -1).contiguous()
        return obs, share

    def reset(self):
        self.t = 0
        self._obs, share = self._draw()
        return self._obs, share, None

    def step(self, actions):
        t = self.torch
        act = t.stack([a.reshape(self.num_envs, self.act_dim) for a in actions], dim=1)
        target = t.tanh(t.einsum("nad,adk->nak", self._obs, self.W))
        team = -((act - target) ** 2).mean(dim=(1, 2))
        rewards = team.view(-1, 1, 1).expand(-1, self.num_agents, 1).contiguous()
        costs = (t.rand(self.num_envs, 1, 1, generator=self.gen) < self.p_cost).float().expand(-1, self.num_agents, 1).contiguous()
        self.t += 1
        dones = t.full((self.num_envs, self.num_agents), self.t % self.trunc_len == 0)
        self._obs, share = self._draw()
        return self._obs, share, rewards, costs, dones, [{} for _ in range(self.num_envs)], None
