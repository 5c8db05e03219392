"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz by EXECUTING the unmodified
reference (/root/reference) in the build container.

    python oracle/make_golden.py            # writes tests/golden/{gae,model,ppo_lag_trace,cpo_trace}.npz

The reference is Python and cannot travel to the GPU box, so its outputs are committed
as small fixtures together with this script (task brief section 3).  Every array stored
here is produced by reference code: VectorizedOnPolicyBuffer.store/finish_path/get,
ActorVCritic.step, and complete runs of safepo.single_agent.{ppo_lag,cpo}.main on the
seeded host SynthEnv, observed through wrappers that record but do not alter values.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402
from oracle.synth_env import Space, SynthEnv  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _self_contained(fn):
    """Every generator fixes its own thread count and seeds, so the fixtures do not depend on which generators ran before it in
    the process: the reference's main() calls torch.set_num_threads(4), which used to leak into whatever came next (VERDICT r03:
    golden_ma_mappolag() after golden_trace("cpo") gave 121 / 268 different arrays -- fp32 sums re-associate with the thread
    count).  The committed files are the single-thread outputs (the traces of main() run at the reference's own 4 threads)."""
    import functools
    import random

    @functools.wraps(fn)
    def wrapped(*a, **k):
        torch.set_num_threads(1)
        torch.manual_seed(0); np.random.seed(0); random.seed(0)
        try:
            return fn(*a, **k)
        finally:
            torch.set_num_threads(1)
    return wrapped


def _np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


# ------------------------------------------------------------------ GAE
@_self_contained
def golden_gae():
    P = ref_shim.load_reference("ppo_lag")
    Buf = P.VectorizedOnPolicyBuffer
    out = {}
    for tag, (N, T, p_seg, seed) in {"a": (8, 48, 1 / 8, 1), "b": (16, 128, 1 / 32, 2),
                                      "c": (3, 7, 0.4, 3), "d": (5, 1, 0.0, 4)}.items():
        rng = np.random.default_rng(seed)
        reward = rng.standard_normal((N, T)).astype(np.float32)
        cost = (rng.random((N, T)) < 0.3).astype(np.float32)
        v_r = rng.standard_normal((N, T)).astype(np.float32)
        v_c = rng.standard_normal((N, T)).astype(np.float32)
        seg = rng.random((N, T)) < p_seg
        seg[:, T - 1] = True
        term = seg & (rng.random((N, T)) < 0.5)
        boot_r = np.where(seg & ~term, rng.standard_normal((N, T)), 0).astype(np.float32)
        boot_c = np.where(seg & ~term, rng.standard_normal((N, T)), 0).astype(np.float32)
        obs = rng.standard_normal((N, T, 6)).astype(np.float32)
        act = rng.standard_normal((N, T, 2)).astype(np.float32)
        logp = rng.standard_normal((N, T)).astype(np.float32)
        buf = Buf(obs_space=Space(6), act_space=Space(2), size=T, num_envs=N, gamma=0.99)
        for t in range(T):
            buf.store(obs=torch.from_numpy(obs[:, t]), act=torch.from_numpy(act[:, t]),
                      reward=torch.from_numpy(reward[:, t]), cost=torch.from_numpy(cost[:, t]),
                      value_r=torch.from_numpy(v_r[:, t]), value_c=torch.from_numpy(v_c[:, t]),
                      log_prob=torch.from_numpy(logp[:, t]))
            for n in range(N):
                if seg[n, t]:
                    buf.finish_path(last_value_r=torch.tensor([boot_r[n, t]]),
                                    last_value_c=torch.tensor([boot_c[n, t]]), idx=n)
        raw = {k: np.stack([b[k].numpy().copy() for b in buf.buffers]) for k in
               ("adv_r", "adv_c", "target_value_r", "target_value_c")}
        data = _np(buf.get())
        for k, v in dict(reward=reward, cost=cost, value_r=v_r, value_c=v_c, seg_end=seg.astype(np.uint8),
                         boot_r=boot_r, boot_c=boot_c, obs=obs, act=act, log_prob=logp).items():
            out[f"{tag}_in_{k}"] = v
        for k, v in raw.items():
            out[f"{tag}_raw_{k}"] = v
        for k, v in data.items():
            out[f"{tag}_get_{k}"] = v
    np.savez_compressed(os.path.join(OUT, "gae.npz"), **out)
    print("gae.npz", len(out), "arrays")


# ------------------------------------------------------------------ model
@_self_contained
def golden_model():
    P = ref_shim.load_reference("ppo_lag")
    torch.manual_seed(7)
    pol = P.ActorVCritic(obs_dim=60, act_dim=8, hidden_sizes=[64, 64])
    out = {f"sd_{k}": v.numpy().copy() for k, v in pol.state_dict().items()}
    obs = torch.randn(33, 60)
    torch.manual_seed(11)
    with torch.no_grad():
        act, logp, v_r, v_c = pol.step(obs, deterministic=False)
        dist = pol.actor(obs)
        eps = (act - dist.mean) / dist.stddev
        act_d, logp_d, _, _ = pol.step(obs, deterministic=True)
        a1, l1, r1, c1 = pol.step(obs[5], deterministic=True)       # single row (bootstrap call shape)
    out.update(_np(dict(obs=obs, act=act, logp=logp, v_r=v_r, v_c=v_c, mean=dist.mean, eps=eps,
                        act_det=act_d, logp_det=logp_d, row5_v_r=r1, row5_v_c=c1)))
    np.savez_compressed(os.path.join(OUT, "model.npz"), **out)
    print("model.npz", len(out), "arrays")


# ------------------------------------------------------------------ full-main traces
class Recorder:
    def __init__(self):
        self.a = {}

    def put(self, key, val):
        self.a[key] = val.detach().cpu().numpy().copy() if torch.is_tensor(val) else np.asarray(val).copy()


def _instrument(P, rec: Recorder, env_kw: dict, algo: str):
    """Wrap (not modify) the names main() looks up in its module globals."""
    state = {"epoch": 0, "policy": None, "perm_i": 0}
    RealPolicy, RealBuf, RealLogger = P.ActorVCritic, P.VectorizedOnPolicyBuffer, P.EpochLogger
    RealDS, RealDL = P.TensorDataset, P.DataLoader

    def policy_factory(*a, **k):
        pol = RealPolicy(*a, **k)
        state["policy"] = pol
        for kk, v in pol.state_dict().items():
            rec.put(f"init_sd_{kk}", v)
        return pol

    class Buf(RealBuf):
        def get(self):
            e = state["epoch"]
            for k in ("obs", "act", "reward", "cost", "value_r", "value_c", "log_prob", "adv_r", "adv_c",
                      "target_value_r", "target_value_c"):
                rec.put(f"e{e}_raw_{k}", torch.stack([b[k] for b in self.buffers]))
            rec.put(f"e{e}_seg_end", np.stack(self._seg_log))
            rec.put(f"e{e}_boot_r", np.stack(self._boot_r_log))
            rec.put(f"e{e}_boot_c", np.stack(self._boot_c_log))
            for kk, v in state["policy"].state_dict().items():
                rec.put(f"e{e}_sd_before_{kk}", v)
            data = super().get()
            for k in ("adv_r", "adv_c"):
                rec.put(f"e{e}_get_{k}", data[k])
            self._reset_logs()
            return data

        def _reset_logs(self):
            size = self.buffers[0]["reward"].shape[0]
            self._seg_log = [np.zeros(size, np.uint8) for _ in range(self.num_envs)]
            self._boot_r_log = [np.zeros(size, np.float32) for _ in range(self.num_envs)]
            self._boot_c_log = [np.zeros(size, np.float32) for _ in range(self.num_envs)]

        def finish_path(self, last_value_r=None, last_value_c=None, idx=0):
            if not hasattr(self, "_seg_log"):
                self._reset_logs()
            t = self.ptr_list[idx] - 1
            self._seg_log[idx][t] = 1
            self._boot_r_log[idx][t] = float(last_value_r.reshape(-1)[0])
            self._boot_c_log[idx][t] = float(last_value_c.reshape(-1)[0])
            return super().finish_path(last_value_r, last_value_c, idx)

    class Log(RealLogger):
        def __init__(self, *a, **k):
            k["use_tensorboard"] = False
            super().__init__(*a, **k)
            self._mb = []

        def store(self, add_value=False, **kw):
            if "Loss/Loss_reward_critic" in kw:
                self._mb.append([kw.get("Loss/Loss_reward_critic", np.nan), kw.get("Loss/Loss_cost_critic", np.nan),
                                 kw.get("Loss/Loss_actor", np.nan)])
            if "Misc/Alpha" in kw:                       # CPO: right after the actor update
                e = state["epoch"]
                for k, v in kw.items():
                    rec.put(f"e{e}_{k.replace('/', '_')}", np.float64(v))
                for kk, v in state["policy"].actor.state_dict().items():
                    rec.put(f"e{e}_actor_after_{kk}", v)
            return super().store(add_value=add_value, **kw)

        def get_stats(self, key):
            v = super().get_stats(key)
            rec.put(f"e{state['epoch']}_get_stats_{key.replace('/', '_')}", np.float64(v))
            return v

        def dump_tabular(self):
            e = state["epoch"]
            for k, v in self.log_current_row.items():
                rec.put(f"e{e}_row_{k.replace('/', '_')}", np.float64(v))
            self._flush_mb()
            super().dump_tabular()

        def _flush_mb(self):
            if self._mb:
                rec.put(f"e{state['epoch']}_mb_losses", np.asarray(self._mb, np.float64))
                self._mb = []

        def close(self):
            self._flush_mb()
            super().close()

    def ds_factory(*tensors):
        return RealDS(*tensors, torch.arange(tensors[0].shape[0]))

    class DL:
        def __init__(self, dataset, batch_size, shuffle):
            self.inner = RealDL(dataset=dataset, batch_size=batch_size, shuffle=shuffle)
            e = state["epoch"]
            rec.put(f"e{e}_batch_size", np.int64(batch_size))
            n_loaders = state.setdefault("loaders", {})
            if n_loaders.get(e, 0) > 0:                # CUP builds a second loader per epoch (cup.py:362)
                for kk, v in state["policy"].state_dict().items():
                    rec.put(f"e{e}_sd_stage{n_loaders[e]}_{kk}", v)
            n_loaders[e] = n_loaders.get(e, 0) + 1

        def __iter__(self):
            idxs = []
            for batch in self.inner:
                idxs.append(batch[-1])
                yield batch[:-1]
            e = state["epoch"]
            passes = state.setdefault("passes", {})    # numbered per epoch across all loaders
            rec.put(f"e{e}_perm{passes.get(e, 0)}", torch.cat(idxs))
            passes[e] = passes.get(e, 0) + 1

    P.ActorVCritic = policy_factory
    P.VectorizedOnPolicyBuffer = Buf
    P.EpochLogger = Log
    P.TensorDataset = ds_factory
    P.DataLoader = DL
    ref_shim.set_env_factory(P, lambda n, env_id, seed: (
        SynthEnv(n, seed=(0 if seed is None else seed), **env_kw), Space(env_kw["obs_dim"]), Space(env_kw["act_dim"])))
    return state, (RealPolicy, RealBuf, RealLogger, RealDS, RealDL)


def _restore(P, saved):
    P.ActorVCritic, P.VectorizedOnPolicyBuffer, P.EpochLogger, P.TensorDataset, P.DataLoader = saved


@_self_contained
def golden_trace(algo: str, fname: str, num_envs: int, T: int, epochs: int, env_kw: dict, cfg_over: dict,
                 fvp_calls: int = 0, args_over: dict | None = None):
    P = ref_shim.load_reference(algo)
    rec = Recorder()
    state, saved = _instrument(P, rec, env_kw, algo)
    cfg_saved = dict(P.default_cfg)
    P.default_cfg.update(cfg_over)
    real_fvp = getattr(P, "fvp", None)
    if real_fvp is not None and fvp_calls:
        calls = {"n": 0}

        def fvp_rec(params, policy, fvp_obs):
            out = real_fvp(params, policy, fvp_obs)
            if calls["n"] < fvp_calls and float(params.abs().sum()) > 0:
                rec.put(f"fvp_in{calls['n']}", params)
                rec.put(f"fvp_out{calls['n']}", out)
                for kk, v in policy.actor.state_dict().items():
                    rec.put(f"fvp_sd{calls['n']}_{kk}", v)
                calls["n"] += 1
            return out
        P.fvp = fvp_rec
    # epoch counter: bump when LinearLR.step()/buffer.get cycle completes -> hook update_end via logger rows
    RealBufGet = P.VectorizedOnPolicyBuffer.get

    def counting_get(self):
        d = RealBufGet(self)
        return d
    steps = num_envs * T
    args = ref_shim.make_args(num_envs=num_envs, steps_per_epoch=steps, total_steps=steps * epochs, seed=0,
                              log_dir=f"/tmp/oracle_runs/{algo}/task/run", **(args_over or {}))
    for k in ("cost_limit", "lagrangian_multiplier_init", "lagrangian_multiplier_lr"):
        rec.put(f"meta_arg_{k}", np.float64(getattr(args, k)))
    # main() has no epoch hook; advance our counter from the rollout loop by wrapping env.step count:
    env_holder = {}
    fac = P.make_sa_mujoco_env

    def fac2(num_envs, env_id, seed=None):
        env, o, a = fac(num_envs, env_id, seed)
        if "train" not in env_holder:
            env_holder["train"] = env
            real_step = env.step

            def step(action):
                state["epoch"] = env.n_steps // T
                return real_step(action)
            env.step = step
        return env, o, a
    P.make_sa_mujoco_env = fac2
    try:
        P.main(args, {})
    finally:
        _restore(P, saved)
        P.default_cfg.clear()
        P.default_cfg.update(cfg_saved)
        if real_fvp is not None:
            P.fvp = real_fvp
    pol = state["policy"]
    for kk, v in pol.state_dict().items():
        rec.put(f"final_sd_{kk}", v)
    rec.put("meta_num_envs", np.int64(num_envs))
    rec.put("meta_T", np.int64(T))
    rec.put("meta_epochs", np.int64(epochs))
    for k, v in env_kw.items():
        rec.put(f"meta_env_{k}", np.float64(v))
    for k, v in {**cfg_saved, **cfg_over}.items():
        if not isinstance(v, (list, tuple)):
            rec.put(f"meta_cfg_{k}", np.float64(v))
    np.savez_compressed(os.path.join(OUT, fname), **rec.a)
    print(fname, len(rec.a), "arrays")


@_self_contained
def golden_pid():
    """PIDLagrangian multipliers for a fixed episode-cost sequence (reference safepo/common/lagrange.py:108-200,
    loaded straight from its file: the module has no third-party imports)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_lagrange", os.path.join(ref_shim.REF_ROOT, "safepo/common/lagrange.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(0)
    costs = np.concatenate([[0.0], rng.uniform(0, 60, 40), np.full(8, 25.0), rng.uniform(20, 30, 16)])
    out = {}
    for tag, kw in {"default": {}, "diffnorm": {"diff_norm": True}, "nosum": {"sum_norm": False, "penalty_max": 0.5}}.items():
        pid = m.PIDLagrangian(cost_limit=25.0, lagrangian_multiplier_init=0.001, **kw)
        vals = []
        for c in costs:
            pid.update_lagrange_multiplier(float(c))
            vals.append(pid.lagrangian_multiplier)
        out[tag] = np.asarray(vals)
    np.savez_compressed(os.path.join(OUT, "pid.npz"), costs=costs, **out)
    print("pid.npz")


@_self_contained
def golden_ma_gae():
    """Multi-agent masked GAE + PopArt: reference SeparatedReplayBuffer.compute_returns / compute_cost_returns
    (safepo/common/buffer.py:356-384) with a PopArt value normaliser (safepo/common/popart.py)."""
    if ref_shim.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_shim.REF_ROOT)
    from safepo.common.buffer import SeparatedReplayBuffer
    from safepo.common.popart import PopArt
    out = {}
    for tag, (T, N, p_done, seed) in {"a": (8, 5, 0.2, 1), "b": (50, 300, 0.05, 2), "c": (1000, 10, 0.01, 3)}.items():
        torch.manual_seed(seed)
        cfg = dict(episode_length=T, n_rollout_threads=N, hidden_size=8, recurrent_N=1, gamma=0.96, gae_lambda=0.95,
                   use_gae=True, use_popart=True, use_valuenorm=True, use_proper_time_limits=False,
                   algorithm_name="mappolag", device="cpu")
        buf = SeparatedReplayBuffer(cfg, Space(6), Space(9), Space(3))
        buf.rewards.copy_(torch.randn(T, N, 1))
        buf.costs.copy_((torch.rand(T, N, 1) < 0.3).float())
        buf.value_preds.copy_(torch.randn(T + 1, N, 1))
        buf.cost_preds.copy_(torch.randn(T + 1, N, 1))
        buf.masks.copy_((torch.rand(T + 1, N, 1) > p_done).float())
        norm = PopArt(1)
        for _ in range(3):
            norm(torch.randn(64, 1) * 2.5 + 1.0, train=True)
        next_v, next_c = torch.randn(N, 1), torch.randn(N, 1)
        buf.compute_returns(next_v, norm)
        buf.compute_cost_returns(next_c, norm)
        mean, var = norm.running_mean_var()
        for k, v in dict(rewards=buf.rewards, costs=buf.costs, value_preds=buf.value_preds, cost_preds=buf.cost_preds,
                         masks=buf.masks, returns=buf.returns, cost_returns=buf.cost_returns, next_v=next_v, next_c=next_c,
                         rm=norm.running_mean, rms=norm.running_mean_sq, deb=norm.debiasing_term.reshape(1),
                         mean=mean, var=var).items():
            out[f"{tag}_{k}"] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "ma_gae.npz"), **out)
    print("ma_gae.npz", len(out), "arrays")


@_self_contained
def golden_ma_mappolag():
    """MAPPO-L networks and trainer step: the reference MAPPO_L_Policy / MAPPO_L_Trainer.ppo_update
    (safepo/multi_agent/mappolag.py:45-199) run on fixed samples, for the default config (no active masks, entropy 0)
    and the mamujoco overrides (active masks, entropy 0.01)."""
    import importlib
    import yaml
    if ref_shim.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_shim.REF_ROOT)
    ref_shim._install_stubs()
    M = importlib.import_module("safepo.multi_agent.mappolag")
    base = yaml.safe_load(open(os.path.join(ref_shim.REF_ROOT, "safepo/multi_agent/marl_cfg/mappolag/config.yaml")))
    out = {}
    for tag, over in {"default": {}, "mamujoco": dict(base["mamujoco"])}.items():
        cfg = dict(base)
        cfg.update(over)
        cfg.update(device="cpu", hidden_size=64, cost_limit=0.5, lagrangian_coef_rate=0.05, actor_lr=3e-3, critic_lr=3e-3)
        torch.manual_seed(11)
        D, S, A, B = 20, 33, 5, 96
        pol = M.MAPPO_L_Policy(cfg, Space(D), Space(S), Space(A))
        with torch.no_grad():                      # default init has v_out == 0 and a tiny actor head: perturb so every path is live
            for net in (pol.actor, pol.critic, pol.cost_critic):
                for prm in net.parameters():
                    prm.add_(0.05 * torch.randn_like(prm))
        tr = M.MAPPO_L_Trainer(cfg, pol)
        for nm, net in (("actor", pol.actor), ("critic", pol.critic), ("cost_critic", pol.cost_critic)):
            for k, v in net.state_dict().items():
                out[f"{tag}_init_{nm}_{k}"] = v.detach().numpy().copy()
        share_obs, obs = torch.randn(B, S), torch.randn(B, D)
        with torch.no_grad():
            values, actions, logp, _, _, cost_preds, _ = pol.get_actions(share_obs, obs, torch.zeros(B, 1, 64), torch.zeros(B, 1, 64),
                                                                         torch.ones(B, 1), rnn_states_cost=torch.zeros(B, 1, 64))
            mean = pol.actor.act.action_out(pol.actor.base(obs)).mean
        out[f"{tag}_fwd_mean"], out[f"{tag}_fwd_values"], out[f"{tag}_fwd_cost_preds"] = mean.numpy(), values.numpy(), cost_preds.numpy()
        out[f"{tag}_fwd_logp"] = logp.numpy()
        old_logp = logp + 0.05 * torch.randn(B, A)
        active = (torch.rand(B, 1) > 0.2).float()
        sample = (share_obs, obs, torch.zeros(B, 1, 64), torch.zeros(B, 1, 64), actions, values + 0.3 * torch.randn(B, 1),
                  torch.randn(B, 1) * 2 + 0.5, torch.ones(B, 1), active, old_logp, torch.randn(B, 1), None,
                  torch.rand(B, 1) + 0.5, cost_preds + 0.3 * torch.randn(B, 1), torch.rand(B, 1) * 3, torch.zeros(B, 1, 64),
                  torch.randn(B, 1), torch.tensor(0.9))
        names = ["share_obs", "obs", None, None, "actions", "value_preds", "returns", None, "active_masks", "old_logp", "adv",
                 None, "factor", "cost_preds", "cost_returns", None, "cost_adv", "aver_episode_costs"]
        for nme, t in zip(names, sample):
            if nme:
                out[f"{tag}_{nme}"] = t.numpy().copy()
        steps = []
        for _ in range(3):
            vl, cgn, plo, ent, agn, imp, cl, cogn = tr.ppo_update(sample)
            steps.append([float(vl), float(cgn), float(plo), float(ent), float(agn), float(imp.mean()), float(cl), float(cogn),
                          float(tr.lamda_lagr), float(tr.value_normalizer.running_mean), float(tr.value_normalizer.running_mean_sq),
                          float(tr.value_normalizer.debiasing_term)])
        out[f"{tag}_steps"] = np.asarray(steps, np.float64)
        for nm, net in (("actor", pol.actor), ("critic", pol.critic), ("cost_critic", pol.cost_critic)):
            for k, v in net.state_dict().items():
                out[f"{tag}_final_{nm}_{k}"] = v.detach().numpy().copy()
        for k in ("clip_param", "entropy_coef", "huber_delta", "value_loss_coef", "max_grad_norm", "actor_lr", "critic_lr",
                  "opti_eps", "weight_decay", "lamda_lagr", "cost_limit", "gamma", "lagrangian_coef_rate", "std_x_coef",
                  "std_y_coef", "layer_N", "hidden_size"):
            out[f"{tag}_cfg_{k}"] = np.float64(cfg[k])
        out[f"{tag}_cfg_use_policy_active_masks"] = np.float64(cfg["use_policy_active_masks"])
    np.savez_compressed(os.path.join(OUT, "ma_mappolag.npz"), **out)
    print("ma_mappolag.npz", len(out), "arrays")


@_self_contained
def golden_ma_happo_mappo():
    """HAPPO / MAPPO trainers: the reference {HAPPO,MAPPO}_Trainer.ppo_update (happo.py:124-169, mappo.py:119-161) for three
    steps on a fixed sample, and one {HAPPO,MAPPO}_Trainer.train (happo.py:171-191, mappo.py:163-183) over a filled
    reference SeparatedReplayBuffer (advantage standardisation + learning_iters whole-buffer steps)."""
    import importlib
    import yaml
    if ref_shim.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_shim.REF_ROOT)
    ref_shim._install_stubs()
    RB = importlib.import_module("safepo.common.buffer")
    out = {}
    cases = {"happo_default": ("happo", {}), "happo_masked": ("happo", {"use_value_active_masks": True, "use_policy_active_masks": True,
                                                                       "entropy_coef": 0.01}),
             "mappo_default": ("mappo", {}), "mappo_mamujoco": ("mappo", "mamujoco")}
    for tag, (algo, over) in cases.items():
        M = importlib.import_module(f"safepo.multi_agent.{algo}")
        base = yaml.safe_load(open(os.path.join(ref_shim.REF_ROOT, f"safepo/multi_agent/marl_cfg/{algo}/config.yaml")))
        cfg = dict(base)
        cfg.update(base["mamujoco"] if over == "mamujoco" else over)
        D, S, A, B, T, N = 20, 33, 5, 96, 6, 8
        cfg.update(device="cpu", hidden_size=32, actor_lr=3e-3, critic_lr=3e-3, episode_length=T, n_rollout_threads=N,
                   learning_iters=3, num_mini_batch=1)
        Pol, Tr = (M.HAPPO_Policy, M.HAPPO_Trainer) if algo == "happo" else (M.MAPPO_Policy, M.MAPPO_Trainer)
        torch.manual_seed(17)

        def fresh(prefix):
            pol = Pol(cfg, Space(D), Space(S), Space(A))
            with torch.no_grad():
                for net in (pol.actor, pol.critic):
                    for prm in net.parameters():
                        prm.add_(0.05 * torch.randn_like(prm))
            for nm, net in (("actor", pol.actor), ("critic", pol.critic)):
                for k, v in net.state_dict().items():
                    out[f"{tag}_{prefix}_{nm}_{k}"] = v.detach().numpy().copy()
            return pol, Tr(cfg, pol)

        def final(prefix, pol):
            for nm, net in (("actor", pol.actor), ("critic", pol.critic)):
                for k, v in net.state_dict().items():
                    out[f"{tag}_{prefix}_{nm}_{k}"] = v.detach().numpy().copy()
        # ---- three ppo_update steps on one sample
        pol, tr = fresh("init")
        share_obs, obs = torch.randn(B, S), torch.randn(B, D)
        H = cfg["hidden_size"]
        with torch.no_grad():
            values, actions, logp, _, _ = pol.get_actions(share_obs, obs, torch.zeros(B, 1, H), torch.zeros(B, 1, H), torch.ones(B, 1))
        old_logp = logp + 0.05 * torch.randn(B, A)
        active = (torch.rand(B, 1) > 0.2).float()
        sample = (share_obs, obs, torch.zeros(B, 1, H), torch.zeros(B, 1, H), actions, values + 0.3 * torch.randn(B, 1),
                  torch.randn(B, 1) * 2 + 0.5, torch.ones(B, 1), active, old_logp, torch.randn(B, 1), None, torch.rand(B, 1) + 0.5)
        names = ["share_obs", "obs", None, None, "actions", "value_preds", "returns", None, "active_masks", "old_logp", "adv", None,
                 "factor"]
        for nme, t in zip(names, sample):
            if nme:
                out[f"{tag}_{nme}"] = t.numpy().copy()
        steps = []
        for _ in range(3):
            vl, cgn, plo, ent, agn, imp = tr.ppo_update(sample)
            vn = tr.value_normalizer
            steps.append([float(vl), float(cgn), float(plo), float(ent), float(agn), float(imp.mean()), float(vn.running_mean),
                          float(vn.running_mean_sq), float(vn.debiasing_term)])
        out[f"{tag}_steps"] = np.asarray(steps, np.float64)
        final("final", pol)
        # ---- one Trainer.train over a filled buffer
        pol, tr = fresh("tinit")
        buf = RB.SeparatedReplayBuffer(cfg, Space(D), Space(S), Space(A))
        with torch.no_grad():
            buf.share_obs.copy_(torch.randn_like(buf.share_obs)); buf.obs.copy_(torch.randn_like(buf.obs))
            flat_o, flat_s = buf.obs[:-1].reshape(T * N, D), buf.share_obs[:-1].reshape(T * N, S)
            v, a, lp, _, _ = pol.get_actions(flat_s, flat_o, torch.zeros(T * N, 1, H), torch.zeros(T * N, 1, H), torch.ones(T * N, 1))
            buf.actions.copy_(a.reshape(T, N, A)); buf.action_log_probs.copy_((lp + 0.03 * torch.randn_like(lp)).reshape(T, N, A))
            buf.value_preds.copy_(torch.randn_like(buf.value_preds) * 0.5)
            buf.returns.copy_(torch.randn_like(buf.returns) * 2 + 0.3)
            buf.active_masks.copy_((torch.rand_like(buf.active_masks) > 0.15).float())
            buf.update_factor(torch.rand(T, N, 1) + 0.5)
        for k in ("share_obs", "obs", "actions", "action_log_probs", "value_preds", "returns", "active_masks", "factor"):
            out[f"{tag}_buf_{k}"] = getattr(buf, k).numpy().copy()

        class _Log:
            rows = []

            def store(self, **kw):
                self.rows.append([kw["Loss/Loss_reward_critic"], kw["Misc/Reward_critic_norm"], kw["Loss/Loss_actor"],
                                  kw["Misc/Entropy"], kw["Misc/Ratio"]])
        lg = _Log()
        lg.rows = []
        tr.train(buf, lg)
        out[f"{tag}_train_rows"] = np.asarray(lg.rows, np.float64)
        vn = tr.value_normalizer
        out[f"{tag}_train_popart"] = np.asarray([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        final("tfinal", pol)
        for k in ("clip_param", "entropy_coef", "huber_delta", "value_loss_coef", "max_grad_norm", "actor_lr", "critic_lr",
                  "opti_eps", "weight_decay", "gamma", "std_x_coef", "std_y_coef", "layer_N", "hidden_size", "learning_iters",
                  "num_mini_batch", "use_policy_active_masks", "use_value_active_masks"):
            out[f"{tag}_cfg_{k}"] = np.float64(cfg[k])
    np.savez_compressed(os.path.join(OUT, "ma_happo_mappo.npz"), **out)
    print("ma_happo_mappo.npz", len(out), "arrays")


@_self_contained
def golden_ma_macpo():
    """MACPO trainer: the reference MACPO_Trainer.trpo_update (safepo/multi_agent/macpo.py:201-371) for two consecutive
    steps on fixed samples, in three settings that reach different branches of its case analysis (average episode cost
    below / above the limit; the mamujoco block)."""
    import importlib
    import yaml
    if ref_shim.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_shim.REF_ROOT)
    ref_shim._install_stubs()
    M = importlib.import_module("safepo.multi_agent.macpo")
    base = yaml.safe_load(open(os.path.join(ref_shim.REF_ROOT, "safepo/multi_agent/marl_cfg/macpo/config.yaml")))
    out = {}
    cases = {"safe": ({}, 0.2, 0.5), "unsafe": ({}, 0.9, 0.5), "mamujoco": (dict(base["mamujoco"]), 0.7, 0.5),
             "recover": ({}, 30.0, 0.5), "deep_safe": ({}, 0.1, 30.0)}
    for tag, (over, aver_cost, limit) in cases.items():
        cfg = dict(base)
        cfg.update(over)
        cfg.update(device="cpu", hidden_size=32, cost_limit=limit, actor_lr=3e-3, critic_lr=3e-3, algorithm_name="macpo")
        torch.manual_seed(23)
        D, S, A, B, H = 20, 33, 5, 96, 32
        pol = M.MACPO_Policy(cfg, Space(D), Space(S), Space(A))
        with torch.no_grad():
            for net in (pol.actor, pol.critic, pol.cost_critic):
                for prm in net.parameters():
                    prm.add_(0.05 * torch.randn_like(prm))
        tr = M.MACPO_Trainer(cfg, pol)
        for nm, net in (("actor", pol.actor), ("critic", pol.critic), ("cost_critic", pol.cost_critic)):
            for k, v in net.state_dict().items():
                out[f"{tag}_init_{nm}_{k}"] = v.detach().numpy().copy()
        share_obs, obs = torch.randn(B, S), torch.randn(B, D)
        with torch.no_grad():
            values, actions, logp, _, _, cost_preds, _ = pol.get_actions(share_obs, obs, torch.zeros(B, 1, H), torch.zeros(B, 1, H),
                                                                         torch.ones(B, 1), rnn_states_cost=torch.zeros(B, 1, H))
        old_logp = logp + 0.02 * torch.randn(B, A)
        active = (torch.rand(B, 1) > 0.2).float()
        sample = (share_obs, obs, torch.zeros(B, 1, H), torch.zeros(B, 1, H), actions, values + 0.3 * torch.randn(B, 1),
                  torch.randn(B, 1) * 2 + 0.5, torch.ones(B, 1), active, old_logp, torch.randn(B, 1), None,
                  torch.rand(B, 1) + 0.5, cost_preds + 0.3 * torch.randn(B, 1), torch.rand(B, 1) * 3, torch.zeros(B, 1, H),
                  torch.randn(B, 1), torch.tensor(aver_cost))
        names = ["share_obs", "obs", None, None, "actions", "value_preds", "returns", None, "active_masks", "old_logp", "adv",
                 None, "factor", "cost_preds", "cost_returns", None, "cost_adv", "aver_episode_costs"]
        for nme, t in zip(names, sample):
            if nme:
                out[f"{tag}_{nme}"] = t.numpy().copy()
        rows = []
        for it in range(2):
            r = tr.trpo_update(sample)
            (vl, cgn, kl, improve, expected, _ent, _ratio, cost_loss, cost_gn, wrp, _cp, _cr, bgrad, lam, nu, g_dir, b_dir, x, _mu,
             _std, bb) = r
            vn = tr.value_normalizer
            rows.append([float(vl), float(cgn), float(kl), float(improve), float(expected), float(cost_loss), float(cost_gn),
                         float(wrp), float(lam), float(nu), float(bb), float(vn.running_mean), float(vn.running_mean_sq),
                         float(vn.debiasing_term)])
            out[f"{tag}_s{it}_g_step_dir"] = g_dir.detach().numpy().copy()
            out[f"{tag}_s{it}_b_step_dir"] = (b_dir.detach().numpy().copy() if b_dir.dim() else np.zeros_like(g_dir.numpy()))
            out[f"{tag}_s{it}_x"] = x.detach().numpy().copy()
            out[f"{tag}_s{it}_cost_grad"] = bgrad.detach().numpy().copy()
            out[f"{tag}_s{it}_actor_after"] = torch.cat([p.detach().reshape(-1) for p in pol.actor.parameters()]).numpy().copy()
        out[f"{tag}_steps"] = np.asarray(rows, np.float64)
        for nm, net in (("actor", pol.actor), ("critic", pol.critic), ("cost_critic", pol.cost_critic)):
            for k, v in net.state_dict().items():
                out[f"{tag}_final_{nm}_{k}"] = v.detach().numpy().copy()
        for k in ("clip_param", "entropy_coef", "huber_delta", "value_loss_coef", "max_grad_norm", "actor_lr", "critic_lr",
                  "opti_eps", "weight_decay", "cost_limit", "gamma", "std_x_coef", "std_y_coef", "layer_N", "hidden_size",
                  "target_kl", "searching_steps", "conjugate_gradient_iters", "step_fraction", "fraction_coef",
                  "use_policy_active_masks"):
            out[f"{tag}_cfg_{k}"] = np.float64(cfg[k])
    np.savez_compressed(os.path.join(OUT, "ma_macpo.npz"), **out)
    print("ma_macpo.npz", len(out), "arrays")
    for tag in cases:
        print(tag, out[f"{tag}_steps"][:, [2, 3, 7, 8, 9]])


@_self_contained
def golden_ma_runner_trace(algo: str = "mappolag", fname: str = "ma_runner_trace.npz", N: int = 6, T: int = 12, EP: int = 3):
    """Episodes of the reference multi-agent Runner.run() (safepo/multi_agent/{mappolag,happo,macpo}.py) on SynthMAEnv:
    buffers before compute(), returns after it, the agent order and minibatch permutations (recorded from torch.randperm),
    logger rows, multipliers, PopArt statistics and all networks after every episode."""
    import importlib
    import shutil
    import yaml
    from oracle.synth_env import SynthMAEnv
    if ref_shim.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_shim.REF_ROOT)
    ref_shim._install_stubs()
    M = importlib.import_module(f"safepo.multi_agent.{algo}")
    cfg = yaml.safe_load(open(os.path.join(ref_shim.REF_ROOT, f"safepo/multi_agent/marl_cfg/{algo}/config.yaml")))
    cfg.update(cfg["mamujoco"])
    use_cost = algo in ("mappolag", "macpo")
    log_dir = f"/tmp/oracle_runs/ma_runner_{algo}"
    shutil.rmtree(log_dir, ignore_errors=True)
    cfg.update(device="cpu", hidden_size=32, n_rollout_threads=N, n_eval_rollout_threads=2, episode_length=T,
               num_env_steps=N * T * EP, learning_iters=3, num_mini_batch=2, use_eval=False, cost_limit=0.3,
               actor_lr=2e-3, critic_lr=2e-3, log_dir=log_dir, seed=0, algorithm_name=algo, env_name="SynthMA")
    if algo == "mappolag":
        cfg.update(lagrangian_coef_rate=0.05)
    torch.manual_seed(5)
    env = SynthMAEnv(N, seed=3, trunc_len=6)
    rec = Recorder()
    runner = M.Runner(env, None, cfg)
    A = runner.num_agents

    def nets_of(pol):
        out = [("actor", pol.actor), ("critic", pol.critic)]
        if use_cost:
            out.append(("cost_critic", pol.cost_critic))
        return out
    for a in range(A):
        for nm, net in nets_of(runner.policy[a]):
            for k, v in net.state_dict().items():
                rec.put(f"init_a{a}_{nm}_{k}", v)
    state = {"ep": 0, "perms": []}
    real_randperm = torch.randperm

    def randperm(n, *a, **k):
        out = real_randperm(n, *a, **k)
        state["perms"].append(out.clone())
        return out
    real_compute, real_train = runner.compute, runner.train
    BUF = ["share_obs", "obs", "actions", "action_log_probs", "value_preds", "rewards", "masks", "active_masks"]
    if use_cost:
        BUF += ["cost_preds", "costs"]

    def compute():
        e = state["ep"]
        for a in range(A):
            for k in BUF:
                rec.put(f"e{e}_a{a}_{k}", getattr(runner.buffer[a], k))
            if use_cost:
                rec.put(f"e{e}_a{a}_aver_episode_costs", runner.buffer[a].aver_episode_costs)
        real_compute()
        for a in range(A):
            rec.put(f"e{e}_a{a}_returns", runner.buffer[a].returns)
            if use_cost:
                rec.put(f"e{e}_a{a}_cost_returns", runner.buffer[a].cost_returns)

    def train():
        e = state["ep"]
        state["perms"] = []
        torch.randperm = randperm
        try:
            real_train()
        finally:
            torch.randperm = real_randperm
        rec.put(f"e{e}_agent_order", state["perms"][0])
        for i, pm in enumerate(state["perms"][1:]):
            rec.put(f"e{e}_perm{i}", pm)
        for k, v in runner.logger.epoch_dict.items():
            if k.startswith(("Loss/", "Misc/")):
                rec.put(f"e{e}_stored_{k.replace('/', '_')}", np.asarray(v, np.float64))
        for a in range(A):
            tr = runner.trainer[a]
            if algo == "mappolag":
                rec.put(f"e{e}_a{a}_lamda", np.float64(float(tr.lamda_lagr)))
            vn = tr.value_normalizer
            rec.put(f"e{e}_a{a}_popart", np.asarray([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)]))
            for nm, net in nets_of(tr.policy):
                for k, v in net.state_dict().items():
                    rec.put(f"e{e}_a{a}_after_{nm}_{k}", v)
        state["ep"] += 1
    runner.compute, runner.train = compute, train
    runner.run()
    keys = ["clip_param", "entropy_coef", "huber_delta", "value_loss_coef", "max_grad_norm", "actor_lr", "critic_lr", "opti_eps",
            "weight_decay", "cost_limit", "gamma", "gae_lambda", "std_x_coef", "std_y_coef", "layer_N", "hidden_size",
            "learning_iters", "num_mini_batch", "use_policy_active_masks", "use_value_active_masks", "episode_length",
            "n_rollout_threads"]
    if algo == "mappolag":
        keys += ["lamda_lagr", "lagrangian_coef_rate"]
    if algo == "macpo":
        keys += ["target_kl", "searching_steps", "conjugate_gradient_iters", "step_fraction", "fraction_coef"]
    for k in keys:
        rec.put(f"cfg_{k}", np.float64(cfg[k]))
    rec.put("meta_agents", np.int64(A))
    rec.put("meta_episodes", np.int64(EP))
    np.savez_compressed(os.path.join(OUT, fname), **rec.a)
    print(fname, len(rec.a), "arrays", "perms per episode:", len(state["perms"]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    golden_gae()
    golden_model()
    golden_pid()
    golden_ma_gae()
    golden_ma_mappolag()
    golden_ma_happo_mappo()
    golden_ma_macpo()
    golden_ma_runner_trace()
    golden_ma_runner_trace("happo", "ma_runner_trace_happo.npz", N=4, T=8, EP=2)
    golden_ma_runner_trace("macpo", "ma_runner_trace_macpo.npz", N=4, T=8, EP=2)
    env_kw = dict(obs_dim=60, act_dim=8, p_term=0.03, p_cost=0.3, trunc_len=20)
    golden_trace("ppo_lag", "ppo_lag_trace.npz", num_envs=4, T=48, epochs=3, env_kw=env_kw,
                 cfg_over={"learning_iters": 6, "target_kl": 0.004},
                 args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
    golden_trace("cpo", "cpo_trace.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw,
                 cfg_over={"learning_iters": 2, "batch_size": 64}, fvp_calls=3,
                 args_over={"cost_limit": 3.0})
    golden_trace("pcpo", "pcpo_trace.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw,
                 cfg_over={"learning_iters": 2, "batch_size": 64}, args_over={"cost_limit": 3.0})
    golden_trace("focops", "focops_trace.npz", num_envs=4, T=48, epochs=3, env_kw=env_kw,
                 cfg_over={"learning_iters": 8, "target_kl": 0.0006},
                 args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
    golden_trace("cup", "cup_trace.npz", num_envs=4, T=48, epochs=3, env_kw=env_kw,
                 cfg_over={"learning_iters": 6, "target_kl": 0.002},
                 args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
    for _algo in ("natural_pg", "trpo"):
        golden_trace(_algo, f"{_algo}_trace.npz", num_envs=4, T=48, epochs=2, env_kw=env_kw,
                     cfg_over={"learning_iters": 2, "batch_size": 64})
    # the Lagrangian trust-region siblings: a multiplier that starts away from zero and a binding cost limit, so the
    # advantage mix (rcpo.py:325-326, trpo_lag.py:326-327) and the multiplier update both matter in every epoch
    for _algo in ("rcpo", "trpo_lag"):
        golden_trace(_algo, f"{_algo}_trace.npz", num_envs=4, T=48, epochs=3, env_kw=env_kw,
                     cfg_over={"learning_iters": 2, "batch_size": 64},
                     args_over={"cost_limit": 1.0, "lagrangian_multiplier_init": 0.5})
