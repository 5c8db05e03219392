"""Environment factories.

`make_sa_mujoco_env(num_envs, env_id, seed)` keeps the reference contract
(safepo/common/env.py:35-80): returns (env, obs_space, act_space) where
    env.reset() -> (obs[N, obs_dim], info)
    env.step(action[N, act_dim]) -> (obs, reward[N], cost[N], terminated[N], truncated[N], info)
    info["final_observation"] present when an episode ended; env.obs_rms is checkpointed.
Safety-Gymnasium tasks are created exactly like the reference when that package is installed
(simulators are out of scope of this build and are not vendored).  Task ids starting with "Synth"
select the device-resident synthetic env of SURVEY.md 8(d) (Isaac-Gym style: tensors stay in HBM).
"""
from __future__ import annotations

import numpy as np
import torch

from safepo import _abi


class Box:
    """Minimal observation/action space: only `.shape` is consumed by the training loop."""

    def __init__(self, dim: int, low: float = -np.inf, high: float = np.inf):
        self.shape = (int(dim),)
        self.low, self.high = low, high


class DeviceObsNormalizer:
    """Device-side SafeNormalizeObservation (reference safepo/common/wrappers.py:42-49): running mean/var/count
    in fp64 (gymnasium RunningMeanStd: mean 0, var 1, count 1e-4), updated with every batch of observations and
    applied in place by spo_obs_normalize.  `.obs_rms` gives the host object that the training loop checkpoints
    into state{itr}.pkl (ppo_lag.py:383)."""

    class _Rms:
        def __init__(self, mean, var, count):
            self.mean, self.var, self.count = mean, var, count

    def __init__(self, obs_dim: int, device):
        self.D = int(obs_dim)
        self.state = torch.zeros(2 * self.D + 1, dtype=torch.float64, device=device)
        self.state[self.D:2 * self.D] = 1.0
        self.state[2 * self.D] = 1e-4
        self.lib = _abi.load()
        self.update_enabled = True       # False: frozen statistics (evaluation with checkpointed statistics)
        self.pending = False             # fused mode: the env's current observation is still raw (see SynthDeviceEnv)

    def load(self, rms, freeze: bool = True) -> None:
        """Restore checkpointed statistics (state{itr}.pkl "Normalizer": an object or dict with mean / var / count) and,
        by default, stop updating them -- what evaluate.py needs (reference evaluate.py:53-57 assigns eval_env.obs_rms)."""
        get = (lambda k: rms[k]) if isinstance(rms, dict) else (lambda k: getattr(rms, k))
        D = self.D
        self.state[:D] = torch.as_tensor(np.asarray(get("mean"), np.float64).reshape(-1), device=self.state.device)
        self.state[D:2 * D] = torch.as_tensor(np.asarray(get("var"), np.float64).reshape(-1), device=self.state.device)
        self.state[2 * D] = float(get("count"))
        self.update_enabled = not freeze

    def normalize_(self, obs: torch.Tensor, update: bool = True) -> torch.Tensor:
        obs = _abi.require_gpu_tensor(obs, "obs", torch.float32)
        _abi.check(self.lib.spo_obs_normalize(_abi.ptr(obs), _abi.ptr(self.state), obs.shape[0], self.D,
                                              int(update and self.update_enabled), _abi.stream_ptr()), "spo_obs_normalize")
        return obs

    @property
    def obs_rms(self):
        s = self.state.cpu().numpy()
        return self._Rms(s[:self.D].copy(), s[self.D:2 * self.D].copy(), float(s[2 * self.D]))


class SynthDeviceEnv:
    """obs' ~ N(0,1), reward ~ N(0,1), cost ~ Bernoulli(p_cost), terminated ~ Bernoulli(p_term),
    truncated = (episode length >= trunc_len); counter-based RNG on the GPU (spo_synth_env_step).
    Returns device tensors, including a dense `final_observation` [N, obs_dim]."""

    is_device_env = True
    graph_safe = True            # step() = a fixed launch sequence on fixed tensors (engine.rollout_epoch may capture it)

    def __init__(self, num_envs: int, obs_dim: int = 60, act_dim: int = 8, seed: int = 0, p_term: float = 0.0,
                 p_cost: float = 0.1, trunc_len: int = 64, device="cuda:0", normalize_obs: bool = False,
                 obs_scale: float = 1.0, obs_shift: float = 0.0):
        self.num_envs, self.obs_dim, self.act_dim = int(num_envs), int(obs_dim), int(act_dim)
        self.obs_scale, self.obs_shift = float(obs_scale), float(obs_shift)
        self.normalizer = DeviceObsNormalizer(obs_dim, device) if normalize_obs else None
        self.seed, self.p_term, self.p_cost, self.trunc_len = int(seed or 0), float(p_term), float(p_cost), int(trunc_len)
        self.dev = torch.device(device)
        self.lib = _abi.load()
        f32 = dict(dtype=torch.float32, device=self.dev)
        N, D = self.num_envs, self.obs_dim
        self.obs = torch.zeros((N, D), **f32)
        self.final_obs = torch.zeros((N, D), **f32)
        self.reward, self.cost = torch.zeros(N, **f32), torch.zeros(N, **f32)
        self.terminated, self.truncated = torch.zeros(N, **f32), torch.zeros(N, **f32)
        self.t_env = torch.zeros(N, dtype=torch.int32, device=self.dev)
        self.step_count = 0
        # counter-based RNG step = *step_base (device) + (step_count - base_host): identical to step_count, but a captured launch
        # (engine.rollout_epoch replays one HIP graph per step) carries only the offset inside the epoch
        self.step_base = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self._base_host = 0
        self.fused_normalizer = None
        self._identity_rms = {"mean": np.zeros(D), "var": np.ones(D), "count": 1e-4}
        self.single_observation_space, self.single_action_space = Box(D), Box(act_dim, -1.0, 1.0)

    def begin_epoch_base(self) -> None:
        """Move the device-resident part of the step counter to the current step (call outside graph capture / replay)."""
        self._base_host = self.step_count
        self.step_base.fill_(self._base_host)

    def _advance(self):
        self.step_count += 1
        _abi.check(self.lib.spo_synth_env_step_rel(
            _abi.ptr(self.obs), _abi.ptr(self.final_obs), _abi.ptr(self.reward), _abi.ptr(self.cost),
            _abi.ptr(self.terminated), _abi.ptr(self.truncated), _abi.ptr(self.t_env), self.num_envs, self.obs_dim,
            self.seed, self.step_count - self._base_host, _abi.ptr(self.step_base), self.p_term, self.p_cost, self.trunc_len,
            int(self.normalizer is not None), self.obs_scale, self.obs_shift, _abi.stream_ptr()), "spo_synth_env_step_rel")
        if self.normalizer is not None:
            # (raw observations x*scale + shift: inside the kernel), then the running normaliser (as the wrapper does in step())
            if self.fused_normalizer is not None:
                # fused mode: the observation stays RAW here; the training loop hands `fused_normalizer` to
                # engine.collect_step / post_step, whose kernels merge the statistics and normalise on load
                # (spo_policy_step_norm) -- once per observation, tracked by `pending`
                self.normalizer.pending = True
            else:
                self.normalizer.normalize_(self.obs, update=True)

    @property
    def obs_rms(self):
        """What the training loop checkpoints into state{itr}.pkl (ppo_lag.py:383): the live statistics."""
        return self.normalizer.obs_rms if self.normalizer is not None else self._identity_rms

    @obs_rms.setter
    def obs_rms(self, rms):
        self.load_obs_rms(rms)

    def load_obs_rms(self, rms, freeze: bool = True) -> None:
        """Checkpointed statistics for evaluation (reference evaluate.py:53-57): restored into the device normaliser and
        frozen.  Without a normaliser (normalize_obs=False) the statistics are ignored, as an identity wrapper would."""
        if self.normalizer is not None and rms is not None:
            self.normalizer.load(rms, freeze=freeze)

    def fuse_normalize(self, on: bool = True):
        """Leave observations raw and let the engine's fused kernel normalise them (returns the normaliser to pass as
        `rms=` to collect_step / post_step, or None when this env does not normalise)."""
        self.fused_normalizer = self.normalizer if (on and self.normalizer is not None) else None
        return self.fused_normalizer

    def reset(self, seed=None):
        if seed is not None:
            self.seed = int(seed)
        self.t_env.zero_()
        p_term, self.p_term = self.p_term, 0.0
        self._advance()
        self.p_term = p_term
        self.t_env.zero_()
        return self.obs, {}

    def step(self, action):
        self._advance()
        info = {"final_observation": self.final_obs}
        return self.obs, self.reward, self.cost, self.terminated, self.truncated, info


class SynthHostEnv:
    """Host (numpy) counterpart with the gymnasium vector-env return contract the reference loop consumes
    (numpy arrays, bool terminated/truncated, `final_observation` object array with None holes,
    ppo_lag.py:166-186).  Exercises the host-env path of main(): observations cross PCIe every step."""

    is_device_env = False

    def __init__(self, num_envs: int, obs_dim: int = 60, act_dim: int = 8, seed: int = 0, p_term: float = 0.0,
                 p_cost: float = 0.1, trunc_len: int = 64, **_):
        self.num_envs, self.obs_dim, self.act_dim = int(num_envs), int(obs_dim), int(act_dim)
        self.p_term, self.p_cost, self.trunc_len = float(p_term), float(p_cost), int(trunc_len)
        self.rng = np.random.default_rng(seed or 0)
        self.t_env = np.zeros(self.num_envs, dtype=np.int64)
        self.obs_rms = {"mean": np.zeros(obs_dim), "var": np.ones(obs_dim), "count": 1e-4}
        self.single_observation_space, self.single_action_space = Box(obs_dim), Box(act_dim, -1.0, 1.0)

    def reset(self, seed=None):
        self.t_env[:] = 0
        return self.rng.standard_normal((self.num_envs, self.obs_dim)).astype(np.float32), {}

    def step(self, action):
        n = self.num_envs
        assert np.asarray(action).shape == (n, self.act_dim)
        self.t_env += 1
        obs = self.rng.standard_normal((n, self.obs_dim)).astype(np.float32)
        reward = self.rng.standard_normal(n).astype(np.float32)
        cost = (self.rng.random(n) < self.p_cost).astype(np.float32)
        terminated = self.rng.random(n) < self.p_term
        truncated = (self.t_env >= self.trunc_len) & ~terminated
        done = terminated | truncated
        info = {}
        if done.any():
            final = np.empty(n, dtype=object)
            for i in range(n):
                final[i] = obs[i].copy() if done[i] else None
            obs[done] = self.rng.standard_normal((int(done.sum()), self.obs_dim)).astype(np.float32)
            info["final_observation"] = final
            self.t_env[done] = 0
        return obs, reward, cost, terminated, truncated, info


def make_sa_mujoco_env(num_envs: int, env_id: str, seed: int | None = None, device="cuda:0", **synth_kw):
    """(env, obs_space, act_space) -- reference signature plus `device` for synthetic tasks."""
    if env_id.startswith("SynthHost"):
        env = SynthHostEnv(num_envs, seed=seed or 0, **synth_kw)
        return env, env.single_observation_space, env.single_action_space
    if env_id.startswith("Synth"):
        env = SynthDeviceEnv(num_envs, seed=seed or 0, device=device, **synth_kw)
        return env, env.single_observation_space, env.single_action_space
    try:
        import safety_gymnasium
        from safety_gymnasium.vector.async_vector_env import SafetyAsyncVectorEnv
        from safety_gymnasium.wrappers import SafeAutoResetWrapper, SafeRescaleAction, SafeUnsqueeze
        from gymnasium.wrappers.normalize import NormalizeObservation
    except ImportError as e:  # simulators are not part of this image
        raise ImportError(
            f"task {env_id!r} needs safety_gymnasium/gymnasium (not installed here). "
            "Use a 'Synth*' task id for the synthetic device env.") from e

    class SafeNormalizeObservation(NormalizeObservation):
        def step(self, action):
            obs, rews, costs, terminateds, truncateds, infos = self.env.step(action)
            obs = self.normalize(obs) if self.is_vector_env else self.normalize(np.array([obs]))[0]
            return obs, rews, costs, terminateds, truncateds, infos

    if num_envs > 1:
        def create_env():
            return SafeRescaleAction(safety_gymnasium.make(env_id), -1.0, 1.0)
        env = SafeNormalizeObservation(SafetyAsyncVectorEnv([create_env for _ in range(num_envs)]))
        env.reset(seed=seed)
        return env, env.single_observation_space, env.single_action_space
    env = safety_gymnasium.make(env_id)
    env.reset(seed=seed)
    obs_space, act_space = env.observation_space, env.action_space
    env = SafeUnsqueeze(SafeNormalizeObservation(SafeRescaleAction(SafeAutoResetWrapper(env), -1.0, 1.0)))
    return env, obs_space, act_space


class SynthMultiAgentEnv:
    """Device-resident synthetic multi-agent vector env with the interface the MAPPO-L Runner consumes
    (reference safepo/common/env.py make_ma_mujoco_env -> ShareVecEnv: reset() -> (obs, share_obs, available_actions),
    step(list of per-agent actions) -> (obs, share_obs, rewards, costs, dones, infos, available_actions) with
    obs [N, agents, obs_dim], share_obs [N, agents, share_dim], rewards / costs [N, agents, 1], dones [N, agents] bool).
    Dynamics (BASELINE config 5 shape: SafetyMujocoMulti-style, 4 agents, obs 48): observations are i.i.d. normals, the
    team reward prefers actions near a fixed linear map of each agent's observation, cost is Bernoulli; all agents of a
    thread finish together every `trunc_len` steps (auto-reset)."""
    is_device_env = True

    def __init__(self, num_envs: int, num_agents: int = 4, obs_dim: int = 48, act_dim: int = 6, share_dim: int | None = None,
                 seed: int = 0, p_cost: float = 0.2, trunc_len: int = 64, device="cuda:0"):
        self.num_envs, self.num_agents, self.obs_dim, self.act_dim = num_envs, num_agents, obs_dim, act_dim
        self.share_dim = share_dim if share_dim is not None else obs_dim * num_agents // 2
        self.dev = torch.device(device)
        self.p_cost, self.trunc_len, self.t = p_cost, trunc_len, 0
        self.gen = torch.Generator(device=self.dev).manual_seed(int(seed))
        g = torch.Generator().manual_seed(12345)
        self.W = (torch.randn(num_agents, obs_dim, act_dim, generator=g) / obs_dim ** 0.5).to(self.dev)
        self.observation_space = [Box(obs_dim) for _ in range(num_agents)]
        self.share_observation_space = [Box(self.share_dim) for _ in range(num_agents)]
        self.action_space = [Box(act_dim, -1.0, 1.0) for _ in range(num_agents)]
        self._obs = None
        self._dones = {}
        self._infos = [{} for _ in range(num_envs)]          # (nothing to report; built once, not 8 192 dicts per step)

    def _draw(self):
        obs = torch.randn(self.num_envs, self.num_agents, self.obs_dim, device=self.dev, generator=self.gen)
        flat = obs.reshape(self.num_envs, -1)
        share = flat[:, :self.share_dim].unsqueeze(1).expand(-1, self.num_agents, -1).contiguous()
        return obs, share

    def reset(self):
        self.t = 0
        self._obs, share = self._draw()
        return self._obs, share, None

    def _dones_of(self, done: bool):
        d = self._dones.get(done)
        if d is None:
            d = self._dones[done] = torch.full((self.num_envs, self.num_agents), bool(done), device=self.dev)
        return d

    def step(self, actions):
        act = torch.stack([a.reshape(self.num_envs, self.act_dim) for a in actions], dim=1)        # [N, agents, A]
        target = torch.tanh(torch.einsum("nad,adk->nak", self._obs, self.W))
        team = -((act - target) ** 2).mean(dim=(1, 2))                                            # shared team reward
        rewards = team.view(-1, 1, 1).expand(-1, self.num_agents, 1).contiguous()
        costs = (torch.rand(self.num_envs, 1, 1, device=self.dev, generator=self.gen) < self.p_cost).float() \
            .expand(-1, self.num_agents, 1).contiguous()
        self.t += 1
        done = self.t % self.trunc_len == 0
        dones = self._dones_of(bool(done))
        self._obs, share = self._draw()
        return self._obs, share, rewards, costs, dones, self._infos, None


def make_ma_synth_env(cfg_train: dict, seed: int = 0, **kw):
    """Multi-agent counterpart of make_sa_mujoco_env for Synth* tasks (BASELINE config 5 shape by default)."""
    return SynthMultiAgentEnv(cfg_train["n_rollout_threads"], seed=seed, device=cfg_train["device"],
                              **{**(cfg_train.get("env_kwargs") or {}), **kw})
