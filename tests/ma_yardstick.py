"""Test helper (NOT a test module): the fp64 yardstick for the multi-agent rows, the same gate the single-agent
path is held to (tests/envelope.py).

A blanket relative tolerance either fails on rounding noise or hides defects.  Instead the oracle restatement
(oracle/ma_restatement.py, pinned to the reference's own outputs at 2e-5 by tests/test_oracle_golden.py) is evaluated twice on
the same inputs -- in float32 (the reference's arithmetic; where a golden fixture exists, the reference's OWN recorded output
is used as this leg) and in float64 (the yardstick) -- and the HIP result must satisfy

    dist(HIP, f64) <= C * dist(f32, f64) + floor          (C = 3)

in max-norm and in L2: a deviation has to be SHOWN to be rounding of the size the reference itself has.  `floor` is relative
to the scale of the quantity (1e-6 for single evaluations, 1e-5 for quantities after optimiser steps: north_star's bar)."""
from __future__ import annotations

import copy

import numpy as np
import torch

from oracle import ma_restatement as MR

C = 3.0


def gate(hip, f32, f64, rel_floor=1e-6, what="", scale=None):
    """Array gate in max-norm and L2.  Returns (d_hip_max, d_32_max) for reporting."""
    hip, f32, f64 = (np.asarray(t, np.float64).reshape(-1) for t in (hip, f32, f64))
    assert hip.shape == f32.shape == f64.shape, (what, hip.shape, f32.shape, f64.shape)
    assert np.isfinite(hip).all(), f"{what}: non-finite HIP values"
    sc = float(np.abs(f64).max()) if scale is None else float(scale)
    sc = max(sc, 1e-30)
    d_hip, d_32 = np.abs(hip - f64), np.abs(f32 - f64)
    assert d_hip.max() <= C * d_32.max() + rel_floor * sc, \
        f"{what}: max |hip - f64| {d_hip.max():.3e} > {C} x |f32 - f64| {d_32.max():.3e} + {rel_floor:g} x scale {sc:.3e}"
    n = np.sqrt(hip.size)
    assert np.linalg.norm(d_hip) <= C * np.linalg.norm(d_32) + rel_floor * sc * n, \
        f"{what}: L2 |hip - f64| {np.linalg.norm(d_hip):.3e} > {C} x |f32 - f64| {np.linalg.norm(d_32):.3e} + floor"
    return float(d_hip.max()), float(d_32.max())


def gate_rows(hip_rows, f32_rows, f64_rows, rel_floor=1e-5, what="", names=None):
    """Logged scalars of several steps ([steps, columns]): every COLUMN (a loss, a norm, ...) is gated over the steps with
    its own scale, so a small column is not hidden behind a large one."""
    hip_rows, f32_rows, f64_rows = (np.asarray(t, np.float64) for t in (hip_rows, f32_rows, f64_rows))
    assert hip_rows.shape == f32_rows.shape == f64_rows.shape, (what, hip_rows.shape, f32_rows.shape, f64_rows.shape)
    for c in range(hip_rows.shape[1]):
        gate(hip_rows[:, c], f32_rows[:, c], f64_rows[:, c], rel_floor, f"{what} column {names[c] if names else c}")


def to_dtype(obj, dtype):
    """Deep copy of a net / dict of tensors in `dtype`."""
    if isinstance(obj, torch.nn.Module):
        return copy.deepcopy(obj).to(dtype)
    if isinstance(obj, dict):
        return {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in obj.items()}
    raise TypeError(type(obj))


def oracle_trainer(cfg, nets: dict, algo: str, dtype):
    """OracleMATrainer over copies of `nets` in `dtype` (its PopArt statistics and multiplier included)."""
    n = {k: to_dtype(v, dtype) for k, v in nets.items()}
    tr = MR.OracleMATrainer(cfg, n["actor"], n["critic"], n.get("cost_critic"), algo=algo)
    p = tr.popart
    p.running_mean, p.running_mean_sq, p.debiasing_term = p.running_mean.to(dtype), p.running_mean_sq.to(dtype), p.debiasing_term.to(dtype)
    tr.lamda = tr.lamda.to(dtype)
    return tr, n


def oracle_steps(cfg, nets: dict, sample: dict, algo: str, nsteps: int, dtype, snapshots=()):
    """`nsteps` ppo_update (/ trpo_update) steps of the oracle in `dtype` on one sample.  Returns (records, nets,
    {step: {name: flat parameters}} for the 1-based steps in `snapshots`)."""
    tr, n = oracle_trainer(cfg, nets, algo, dtype)
    s = to_dtype(sample, dtype)
    recs, snaps = [], {}
    for k in range(nsteps):
        recs.append(tr.ppo_update(s))
        if k + 1 in snapshots:
            snaps[k + 1] = {nm: net.flat().double().numpy().copy() for nm, net in n.items()}
    return recs, n, snaps


def nets_like(policy, std_x_coef=1.0, std_y_coef=0.5):
    """Oracle nets (float32) holding the parameters of a HIP MAPPO-style policy (actor / critic / cost critic)."""
    out = {}
    for nm, net in (("actor", policy.actor), ("critic", policy.critic), ("cost_critic", getattr(policy, "cost_critic", None))):
        if net is None:
            continue
        g = net._net
        ref = MR.MANet(g.in_dim, g.hidden, g.n_blocks, g.out_dim, bool(g.is_actor), std_x_coef, std_y_coef)
        ref.load_reference_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
        assert torch.equal(ref.flat(), net.theta.cpu())
        out[nm] = ref
    return out


def runner_replay(z, algo: str, dtype):
    """Episodes of the reference multi-agent Runner (ma_runner_trace*.npz) replayed through the restatement in `dtype`:
    compute() (bootstrap values, masked GAE with PopArt de-normalisation) then the HAPPO-sequential train() with the recorded
    agent order and shuffles -- the replay of tests/test_oracle_golden.py::test_ma_runner_restatement_vs_reference_runner_trace,
    parameterised by the arithmetic.  Returns one dict per episode: returns / cost_returns per agent, the stored rows, and per
    agent the multiplier, the PopArt state and the flat parameters of every network after the episode."""
    use_cost = algo in ("mappolag", "macpo")
    A, EP = int(z["meta_agents"]), int(z["meta_episodes"])
    cfg = {k[4:]: float(z[k]) for k in z.files if k.startswith("cfg_")}
    cfg["use_policy_active_masks"] = bool(cfg["use_policy_active_masks"])
    cfg["use_value_active_masks"] = bool(cfg.get("use_value_active_masks", 0))
    H, nb = int(cfg["hidden_size"]), 1 + int(cfg["layer_N"])
    D, S, Ad = z["e0_a0_obs"].shape[-1], z["e0_a0_share_obs"].shape[-1], z["e0_a0_actions"].shape[-1]
    trainers = []
    for a in range(A):
        n = {"actor": MR.MANet(D, H, nb, Ad, True, cfg["std_x_coef"], cfg["std_y_coef"]), "critic": MR.MANet(S, H, nb, 1, False)}
        if use_cost:
            n["cost_critic"] = MR.MANet(S, H, nb, 1, False)
        for nm, net in n.items():
            pre = f"init_a{a}_{nm}_"
            net.load_reference_state_dict({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
        trainers.append(oracle_trainer(cfg, n, algo, dtype)[0])
    iters = 1 if algo == "macpo" else int(cfg["learning_iters"])
    buf_keys = ["share_obs", "obs", "actions", "action_log_probs", "value_preds", "rewards", "masks", "active_masks"]
    buf_keys += ["cost_preds", "costs"] if use_cost else []
    out = []
    for e in range(EP):
        bufs, ep = [], {"returns": [], "cost_returns": []}
        for a in range(A):
            b = {k: torch.from_numpy(z[f"e{e}_a{a}_{k}"].copy()).to(dtype) for k in buf_keys}
            tr = trainers[a]
            with torch.no_grad():
                b["value_preds"][-1] = tr.critic(b["share_obs"][-1])
            b["returns"] = MR.masked_gae(b["rewards"], b["value_preds"], b["masks"], tr.popart, cfg["gamma"], cfg["gae_lambda"])
            ep["returns"].append(b["returns"].double().numpy().copy())
            if use_cost:
                b["aver_episode_costs"] = torch.from_numpy(z[f"e{e}_a{a}_aver_episode_costs"].copy()).to(dtype)
                with torch.no_grad():
                    b["cost_preds"][-1] = tr.cost_critic(b["share_obs"][-1])
                b["cost_returns"] = MR.masked_gae(b["costs"], b["cost_preds"], b["masks"], tr.popart, cfg["gamma"], cfg["gae_lambda"])
                ep["cost_returns"].append(b["cost_returns"].double().numpy().copy())
            bufs.append(b)
        order = [int(i) for i in z[f"e{e}_agent_order"]]
        perms_of = {a: [z[f"e{e}_perm{pos * iters + it}"] for it in range(iters)] for pos, a in enumerate(order)}
        ep["rows"] = np.asarray(MR.runner_train(trainers, bufs, order, perms_of, cfg), np.float64)
        ep["lamda"] = [float(tr.lamda) for tr in trainers]
        ep["popart"] = [[float(tr.popart.running_mean), float(tr.popart.running_mean_sq), float(tr.popart.debiasing_term)] for tr in trainers]
        ep["theta"] = [{nm: getattr(tr, nm).flat().double().numpy().copy() for nm in (("actor", "critic", "cost_critic") if use_cost
                                                                                       else ("actor", "critic"))} for tr in trainers]
        out.append(ep)
    return out, cfg


# columns of the oracle's per-update row -> keys the Runner stores (test_ma_runner_restatement_vs_reference_runner_trace)
RUNNER_COLS = {
    "mappolag": ((0, "Loss/Loss_reward_critic"), (6, "Loss/Loss_cost_critic"), (2, "Loss/Loss_actor"), (1, "Misc/Reward_critic_norm"),
                 (7, "Misc/Cost_critic_norm"), (3, "Misc/Entropy"), (5, "Misc/Ratio")),
    "happo": ((0, "Loss/Loss_reward_critic"), (2, "Loss/Loss_actor"), (1, "Misc/Reward_critic_norm"), (3, "Misc/Entropy"), (5, "Misc/Ratio")),
    "macpo": ((0, "Loss/Loss_reward_critic"), (5, "Loss/Loss_cost_critic"), (3, "Loss/Loss_actor_improve"),
              (4, "Loss/Loss_actor_expected_improve"), (1, "Misc/Reward_critic_norm"), (6, "Misc/Cost_critic_norm"), (2, "Misc/KL")),
}
