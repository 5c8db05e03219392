"""Pins the CPU oracle (oracle/restatement.py) to outputs of the UNMODIFIED reference
captured by oracle/make_golden.py (tests/golden/*.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R

torch.set_num_threads(1)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _policy_from(z, prefix):
    obs_dim, act_dim = z[prefix + "actor.mean.0.weight"].shape[1], z[prefix + "actor.log_std"].shape[0]
    pol = R.OraclePolicy(obs_dim, act_dim)
    sd = {k[len(prefix):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(prefix)}
    pol.load_state_dict(sd)
    return pol


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_gae_bit_exact_vs_reference_buffer(golden_dir, tag):
    z = _load(golden_dir, "gae.npz")
    i = lambda k: z[f"{tag}_in_{k}"]
    adv_r, adv_c, tgt_r, tgt_c = R.gae_dense(i("reward"), i("cost"), i("value_r"), i("value_c"), i("seg_end"),
                                              i("boot_r"), i("boot_c"), 0.99, 0.95, 0.95)
    for got, key in ((adv_r, "adv_r"), (adv_c, "adv_c"), (tgt_r, "target_value_r"), (tgt_c, "target_value_c")):
        ref = z[f"{tag}_raw_{key}"]
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), key
    # per-path scalar restatement agrees too (first env, first path)
    seg = i("seg_end")[0]
    e = int(np.argmax(seg))
    vals = np.concatenate([i("value_r")[0, :e + 1], [i("boot_r")[0, e]]])
    rews = np.concatenate([i("reward")[0, :e + 1], [0.0]])
    a, t = R.gae_path(vals, rews, 0.99, 0.95)
    assert np.array_equal(a, z[f"{tag}_raw_adv_r"][0, :e + 1])
    assert np.array_equal(t, z[f"{tag}_raw_target_value_r"][0, :e + 1])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_get_standardisation_vs_reference(golden_dir, tag):
    z = _load(golden_dir, "gae.npz")
    N, T = z[f"{tag}_in_reward"].shape
    adv_r = torch.from_numpy(z[f"{tag}_raw_adv_r"].reshape(-1))
    adv_c = torch.from_numpy(z[f"{tag}_raw_adv_c"].reshape(-1))
    sr, sc = R.adv_standardize(adv_r, adv_c)
    assert np.array_equal(sr.numpy(), z[f"{tag}_get_adv_r"])
    assert np.array_equal(sc.numpy(), z[f"{tag}_get_adv_c"])
    # env-major flattening of get() (buffer.py:149-153)
    assert np.array_equal(z[f"{tag}_get_obs"], z[f"{tag}_in_obs"].reshape(N * T, -1))
    assert np.array_equal(z[f"{tag}_get_reward"], z[f"{tag}_in_reward"].reshape(-1))


def test_model_step_vs_reference(golden_dir):
    z = _load(golden_dir, "model.npz")
    pol = _policy_from(z, "sd_")
    # parameter registration order = reference order
    names = [n for n, _ in pol.named_parameters()]
    assert names[0] == "reward_critic.critic.0.weight" and names[12] == "actor.log_std"
    assert sum(p.numel() for p in pol.parameters()) == 24850
    with torch.no_grad():
        act, logp, v_r, v_c = pol.step_with_eps(torch.from_numpy(z["obs"]), torch.from_numpy(z["eps"]))
    np.testing.assert_allclose(act.numpy(), z["act"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(logp.numpy(), z["logp"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(v_r.numpy(), z["v_r"]) and np.array_equal(v_c.numpy(), z["v_c"])
    with torch.no_grad():
        _, _, r5, c5 = pol.step_with_eps(torch.from_numpy(z["obs"][5]), torch.zeros(8))
    np.testing.assert_allclose(r5.numpy(), z["row5_v_r"], rtol=1e-6)
    np.testing.assert_allclose(c5.numpy(), z["row5_v_c"], rtol=1e-6)


def _epoch_data(z, e):
    raw = lambda k: z[f"e{e}_raw_{k}"]
    N, T = raw("reward").shape
    adv_r, adv_c, tgt_r, tgt_c = R.gae_dense(raw("reward"), raw("cost"), raw("value_r"), raw("value_c"),
                                              z[f"e{e}_seg_end"], z[f"e{e}_boot_r"], z[f"e{e}_boot_c"],
                                              float(z["meta_cfg_gamma"]), 0.95, 0.95)
    return N, T, adv_r, adv_c, tgt_r, tgt_c


@pytest.mark.parametrize("fname", ["ppo_lag_trace.npz", "ppo_lag_trace_humanoid.npz", "ppo_lag_trace_car.npz"])
def test_ppo_lag_main_trace(golden_dir, fname):
    """Replays the epochs of the reference ppo_lag.main() (recorded buffers, shuffles and initial
    weights) through the oracle: GAE bits, get() standardisation, lambda, per-minibatch losses,
    early-stop iteration, KL and parameters after every epoch.  `_humanoid`: the same run at HumanoidVelocity's dims,
    ActorVCritic(376, 17) (oracle/make_golden_humanoid.py, round 5)."""
    z = _load(golden_dir, fname)
    epochs = int(z["meta_epochs"])
    pol = _policy_from(z, "init_sd_")
    upd = R.PPOLagUpdater(pol, epochs=epochs)
    lag = R.OracleLagrange(float(z["meta_arg_cost_limit"]), float(z["meta_arg_lagrangian_multiplier_init"]),
                           float(z["meta_arg_lagrangian_multiplier_lr"]))
    for e in range(epochs):
        for k, v in pol.state_dict().items():
            np.testing.assert_allclose(v.numpy(), z[f"e{e}_sd_before_{k}"], rtol=2e-6, atol=1e-7, err_msg=k)
        N, T, adv_r, adv_c, tgt_r, tgt_c = _epoch_data(z, e)
        for got, key in ((adv_r, "adv_r"), (adv_c, "adv_c"), (tgt_r, "target_value_r"), (tgt_c, "target_value_c")):
            assert np.array_equal(got, z[f"e{e}_raw_{key}"]), (e, key)
        # every row's last step ends a path; boundary flags recorded from finish_path calls
        assert z[f"e{e}_seg_end"][:, -1].all()
        lag.update_lagrange_multiplier(float(z[f"e{e}_get_stats_Metrics_EpCost"]))
        assert lag.lagrangian_multiplier == pytest.approx(float(z[f"e{e}_row_Train_LagragianMultiplier"]), rel=1e-6)
        sr, sc = R.adv_standardize(torch.from_numpy(adv_r.reshape(-1)), torch.from_numpy(adv_c.reshape(-1)))
        assert np.array_equal(sr.numpy(), z[f"e{e}_get_adv_r"])
        assert np.array_equal(sc.numpy(), z[f"e{e}_get_adv_c"])
        flat = lambda k: torch.from_numpy(z[f"e{e}_raw_{k}"].reshape(N * T, *z[f"e{e}_raw_{k}"].shape[2:]))
        data = {"obs": flat("obs"), "act": flat("act"), "log_prob": flat("log_prob"),
                "target_value_r": torch.from_numpy(tgt_r.reshape(-1)),
                "target_value_c": torch.from_numpy(tgt_c.reshape(-1)), "adv_r": sr, "adv_c": sc}
        n_perm = len([k for k in z.files if k.startswith(f"e{e}_perm")])
        perms = [z[f"e{e}_perm{i}"] for i in range(n_perm)]
        perms += [perms[-1]] * (int(z["meta_cfg_learning_iters"]) - n_perm)
        out = R.ppo_lag_update(pol, upd, data, lag.lagrangian_multiplier, perms,
                               learning_iters=int(z["meta_cfg_learning_iters"]),
                               batch_size=int(z[f"e{e}_batch_size"]), target_kl=float(z["meta_cfg_target_kl"]))
        assert out["stop_iter"] == int(z[f"e{e}_row_Train_StopIter"]) == n_perm
        assert out["kl"] == pytest.approx(float(z[f"e{e}_row_Train_KL"]), rel=1e-4)
        np.testing.assert_allclose(out["losses"], z[f"e{e}_mb_losses"], rtol=2e-5, atol=1e-7)
    for k, v in pol.state_dict().items():
        np.testing.assert_allclose(v.numpy(), z[f"final_sd_{k}"], rtol=2e-5, atol=2e-7, err_msg=k)


@pytest.mark.parametrize("fname", ["cpo_trace.npz", "cpo_trace_humanoid.npz", "cpo_trace_car.npz", "cpo_trace_humanoid_b128.npz"])
def test_cpo_main_trace(golden_dir, fname):
    """Replays the reference cpo.main(): FVP known answers, the actor update (CG, case analysis,
    line search) and the critic fit, epoch by epoch.  `_humanoid`: the same run with ActorVCritic(376, 17) (no FVP vectors)."""
    z = _load(golden_dir, fname)
    # FVP known-answer vectors recorded from the reference's fvp()
    for i in range(3 if "fvp_in0" in z.files else 0):
        pol = R.OraclePolicy(60, 8)
        pol.actor.load_state_dict({k[len(f"fvp_sd{i}_"):]: torch.from_numpy(z[k].copy())
                                   for k in z.files if k.startswith(f"fvp_sd{i}_")})
        obs = torch.from_numpy(z["e0_raw_obs"].reshape(-1, 60))
        got = R.cpo_fvp(torch.from_numpy(z[f"fvp_in{i}"]), pol, obs)
        np.testing.assert_allclose(got.numpy(), z[f"fvp_out{i}"], rtol=1e-5, atol=1e-8)
    epochs = int(z["meta_epochs"])
    pol = _policy_from(z, "init_sd_")
    fit = R.CriticFitter(pol)
    cases = []
    for e in range(epochs):
        for k, v in pol.state_dict().items():
            np.testing.assert_allclose(v.numpy(), z[f"e{e}_sd_before_{k}"], rtol=1e-4, atol=2e-6, err_msg=k)
        N, T, adv_r, adv_c, tgt_r, tgt_c = _epoch_data(z, e)
        assert np.array_equal(adv_r, z[f"e{e}_raw_adv_r"]) and np.array_equal(tgt_c, z[f"e{e}_raw_target_value_c"])
        sr, sc = R.adv_standardize(torch.from_numpy(adv_r.reshape(-1)), torch.from_numpy(adv_c.reshape(-1)))
        flat = lambda k: torch.from_numpy(z[f"e{e}_raw_{k}"].reshape(N * T, *z[f"e{e}_raw_{k}"].shape[2:]))
        data = {"obs": flat("obs"), "act": flat("act"), "log_prob": flat("log_prob"),
                "target_value_r": torch.from_numpy(tgt_r.reshape(-1)),
                "target_value_c": torch.from_numpy(tgt_c.reshape(-1)), "adv_r": sr, "adv_c": sc}
        ep_costs = float(z[f"e{e}_get_stats_Metrics_EpCost"]) - float(z["meta_arg_cost_limit"])
        out = R.cpo_policy_update(pol, data, ep_costs, target_kl=float(z["meta_cfg_target_kl"]))
        cases.append(out["case"])
        assert out["accept"] == int(z[f"e{e}_Misc_AcceptanceStep"])
        assert float(out["xHx"]) == pytest.approx(float(z[f"e{e}_Misc_xHx"]), rel=1e-3)
        assert float(out["x"].norm()) == pytest.approx(float(z[f"e{e}_Misc_H_inv_g"]), rel=1e-3)
        assert float(out["g"].norm()) == pytest.approx(float(z[f"e{e}_Misc_gradient_norm"]), rel=1e-4)
        assert float(out["step_direction"].norm()) == pytest.approx(float(z[f"e{e}_Misc_FinalStepNorm"]), rel=1e-3)
        assert out["kl"] == pytest.approx(float(z[f"e{e}_Train_KL"]), rel=1e-3)
        for k, v in pol.actor.state_dict().items():
            np.testing.assert_allclose(v.numpy(), z[f"e{e}_actor_after_{k}"], rtol=1e-3, atol=2e-6, err_msg=k)
        # critic fit with the recorded shuffles
        bs = int(z[f"e{e}_batch_size"])
        # the actor's stale .grad (= cost gradient b) takes part in clip_grad_norm_ (cpo.py:557)
        R.actor_set_flat_params(pol.actor, R.actor_flat_params(pol.actor))
        i = 0
        for (_, prm) in pol.actor.named_parameters():
            prm.grad = out["b"][i:i + prm.numel()].view(prm.shape).clone()
            i += prm.numel()
        losses = []
        for it in range(int(z["meta_cfg_learning_iters"])):
            perm = torch.from_numpy(z[f"e{e}_perm{it}"])
            for s in range(0, N * T, bs):
                idx = perm[s:s + bs]
                losses.append(fit.minibatch_step(data["obs"][idx], data["target_value_r"][idx],
                                                 data["target_value_c"][idx]))
        np.testing.assert_allclose(np.asarray(losses), z[f"e{e}_mb_losses"][:, :2], rtol=1e-4, atol=1e-7)
    if fname == "cpo_trace.npz":
        assert cases[1] in (0, 1), "epoch 1 of the fixture is an infeasible-recovery case"
    for k, v in pol.state_dict().items():
        np.testing.assert_allclose(v.numpy(), z[f"final_sd_{k}"], rtol=1e-3, atol=5e-6, err_msg=k)


def _trace_epoch_inputs(z, e):
    N, T, adv_r, adv_c, tgt_r, tgt_c = _epoch_data(z, e)
    sr, sc = R.adv_standardize(torch.from_numpy(adv_r.reshape(-1)), torch.from_numpy(adv_c.reshape(-1)))
    flat = lambda k: torch.from_numpy(z[f"e{e}_raw_{k}"].reshape(N * T, *z[f"e{e}_raw_{k}"].shape[2:]))
    data = {"obs": flat("obs"), "act": flat("act"), "log_prob": flat("log_prob"),
            "target_value_r": torch.from_numpy(tgt_r.reshape(-1)), "target_value_c": torch.from_numpy(tgt_c.reshape(-1)),
            "adv_r": sr, "adv_c": sc}
    n_perm = len([k for k in z.files if k.startswith(f"e{e}_perm")])
    return data, [z[f"e{e}_perm{i}"] for i in range(n_perm)]


@pytest.mark.parametrize("algo,upper,suffix", [("focops", 2.0, ""), ("cup", 0.2, ""), ("focops", 2.0, "_humanoid"),
                                               ("cup", 0.2, "_humanoid")])
def test_kl_penalty_family_main_trace(golden_dir, algo, upper, suffix):
    """Replays the reference focops.main() / cup.main() through the restatement: the [B,1] x [B] broadcast of the
    KL-penalty losses, the per-sample indicator, CUP's actor-only second stage with its own optimiser clock.
    `_humanoid`: focops.main() / cup.main() with ActorVCritic(376, 17)."""
    z = _load(golden_dir, f"{algo}_trace{suffix}.npz")
    epochs, iters = int(z["meta_epochs"]), int(z["meta_cfg_learning_iters"])
    pol = _policy_from(z, "init_sd_")
    upd = R.KLPenaltyUpdater(pol, epochs=epochs)
    lag = R.OracleLagrange(float(z["meta_arg_cost_limit"]), float(z["meta_arg_lagrangian_multiplier_init"]),
                           float(z["meta_arg_lagrangian_multiplier_lr"]), lagrangian_upper_bound=upper)
    masked = 0
    for e in range(epochs):
        for k, v in pol.state_dict().items():
            np.testing.assert_allclose(v.numpy(), z[f"e{e}_sd_before_{k}"], rtol=2e-6, atol=1e-7, err_msg=k)
        data, perms = _trace_epoch_inputs(z, e)
        lag.update_lagrange_multiplier(float(z[f"e{e}_get_stats_Metrics_EpCost"]))
        assert lag.lagrangian_multiplier == pytest.approx(float(z[f"e{e}_row_Train_LagragianMultiplier"]), rel=1e-6)
        kw = dict(learning_iters=iters, batch_size=int(z[f"e{e}_batch_size"]), target_kl=float(z["meta_cfg_target_kl"]))
        perms = perms + [perms[-1]] * (2 * iters)
        if algo == "focops":
            out = R.focops_update(pol, upd, data, lag.lagrangian_multiplier, perms, **kw)
        else:
            out = R.cup_update(pol, upd, data, lag.lagrangian_multiplier, perms, float(z["meta_cfg_gamma"]), **kw)
            assert out["second_stage_stop_iter"] == int(z[f"e{e}_row_Train_SeconStageStopIter"])
            if e > 0:
                pass
        assert out["stop_iter"] == int(z[f"e{e}_row_Train_StopIter"])
        assert out["kl"] == pytest.approx(float(z[f"e{e}_row_Train_KL"]), rel=1e-4)
        np.testing.assert_allclose(out["losses"], z[f"e{e}_mb_losses"], rtol=2e-5, atol=1e-7)
    for k, v in pol.state_dict().items():
        np.testing.assert_allclose(v.numpy(), z[f"final_sd_{k}"], rtol=2e-5, atol=2e-7, err_msg=k)


@pytest.mark.parametrize("tag", ["default", "mamujoco"])
def test_ma_mappolag_restatement_vs_reference(golden_dir, tag):
    """Networks (forward, per-dimension log-probs) and three MAPPO_L_Trainer.ppo_update steps of the reference:
    losses, gradient norms, entropy, ratio, in-loop lambda, PopArt statistics, parameters after."""
    from oracle import ma_restatement as MR
    z = _load(golden_dir, "ma_mappolag.npz")
    nets, cfg, s = MR.nets_from_golden(z, tag), MR.cfg_from_golden(z, tag), MR.sample_from_golden(z, tag)
    with torch.no_grad():
        mean = nets["actor"](s["obs"])
        np.testing.assert_allclose(mean.numpy(), z[f"{tag}_fwd_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(MR.log_probs(mean, nets["actor"].std(), s["actions"]).numpy(), z[f"{tag}_fwd_logp"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(nets["critic"](s["share_obs"]).numpy(), z[f"{tag}_fwd_values"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(nets["cost_critic"](s["share_obs"]).numpy(), z[f"{tag}_fwd_cost_preds"], rtol=1e-5, atol=1e-6)
    tr = MR.OracleMATrainer(cfg, nets["actor"], nets["critic"], nets["cost_critic"])
    rows = [tr.ppo_update(s)["row"] for _ in range(3)]
    np.testing.assert_allclose(np.asarray(rows), z[f"{tag}_steps"], rtol=2e-5, atol=1e-7)
    fin = MR.nets_from_golden(z, tag, "final")
    for nm in nets:
        np.testing.assert_allclose(nets[nm].flat().numpy(), fin[nm].flat().numpy(), rtol=2e-5, atol=2e-7, err_msg=nm)


@pytest.mark.parametrize("tag", ["happo_default", "happo_masked", "mappo_default", "mappo_mamujoco"])
def test_ma_happo_mappo_restatement_vs_reference(golden_dir, tag):
    """HAPPO / MAPPO: three reference Trainer.ppo_update steps on a fixed sample, and one Trainer.train over a filled
    reference buffer (plain mean/std standardisation + learning_iters whole-buffer steps)."""
    from oracle import ma_restatement as MR
    z = _load(golden_dir, "ma_happo_mappo.npz")
    algo = tag.split("_")[0]
    nets, cfg, s = MR.nets_from_golden(z, tag), MR.cfg_from_golden(z, tag), MR.sample_from_golden(z, tag)
    assert set(nets) == {"actor", "critic"}
    tr = MR.OracleMATrainer(cfg, nets["actor"], nets["critic"], None, algo=algo)
    rows = [tr.ppo_update(s)["row"] for _ in range(3)]
    np.testing.assert_allclose(np.asarray(rows), z[f"{tag}_steps"], rtol=2e-5, atol=1e-7)
    fin = MR.nets_from_golden(z, tag, "final")
    for nm in nets:
        np.testing.assert_allclose(nets[nm].flat().numpy(), fin[nm].flat().numpy(), rtol=2e-5, atol=2e-7, err_msg=nm)
    # Trainer.train: the reference shuffles the single whole-buffer minibatch, which only reorders the sums
    nets = MR.nets_from_golden(z, tag, "tinit")
    tr = MR.OracleMATrainer(cfg, nets["actor"], nets["critic"], None, algo=algo)
    buf = {k: torch.from_numpy(z[f"{tag}_buf_{k}"].copy()) for k in ("share_obs", "obs", "actions", "action_log_probs",
                                                                     "value_preds", "returns", "active_masks", "factor")}
    buf["rewards"] = torch.zeros_like(buf["factor"])
    iters = int(cfg["learning_iters"])
    n = buf["factor"].numel()
    got = MR.train_agent(tr, buf, [torch.arange(n)] * iters, cfg)
    want = z[f"{tag}_train_rows"]                 # value loss, critic norm, policy loss, entropy, ratio
    np.testing.assert_allclose(np.asarray(got)[:, [0, 1, 2, 3, 5]], want, rtol=5e-5, atol=2e-7)
    np.testing.assert_allclose(np.asarray(got)[-1, 6:9], z[f"{tag}_train_popart"], rtol=1e-5)
    fin = MR.nets_from_golden(z, tag, "tfinal")
    for nm in nets:
        np.testing.assert_allclose(nets[nm].flat().numpy(), fin[nm].flat().numpy(), rtol=5e-5, atol=5e-7, err_msg=nm)


@pytest.mark.parametrize("tag,case", [("safe", 2), ("unsafe", 1), ("mamujoco", 1), ("recover", 0), ("deep_safe", 3)])
def test_ma_macpo_restatement_vs_reference(golden_dir, tag, case):
    """MACPO: two reference MACPO_Trainer.trpo_update steps (critic steps, surrogate gradients, two conjugate-gradient solves
    with the double-backward Fisher product, case analysis, line search) in five settings covering optim cases 0-3."""
    from oracle import ma_restatement as MR
    z = _load(golden_dir, "ma_macpo.npz")
    nets, cfg, s = MR.nets_from_golden(z, tag), MR.cfg_from_golden(z, tag), MR.sample_from_golden(z, tag)
    tr = MR.OracleMATrainer(cfg, nets["actor"], nets["critic"], nets["cost_critic"], algo="macpo")
    for it in range(2):
        rec = tr.ppo_update(s)
        assert rec["case"] == case
        np.testing.assert_allclose(rec["row"], z[f"{tag}_steps"][it], rtol=2e-3, atol=2e-6)
        for k in ("g_dir", "b_dir", "x"):
            want = z[f"{tag}_s{it}_{'g_step_dir' if k == 'g_dir' else 'b_step_dir' if k == 'b_dir' else 'x'}"]
            np.testing.assert_allclose(rec[k].numpy(), want, rtol=5e-3, atol=2e-4 * max(np.abs(want).max(), 1e-6), err_msg=k)
        np.testing.assert_allclose(nets["actor"].flat().numpy(), z[f"{tag}_s{it}_actor_after"], rtol=1e-3, atol=2e-5)
    fin = MR.nets_from_golden(z, tag, "final")
    for nm in ("critic", "cost_critic"):
        np.testing.assert_allclose(nets[nm].flat().numpy(), fin[nm].flat().numpy(), rtol=2e-5, atol=2e-7, err_msg=nm)


@pytest.mark.parametrize("algo,fname", [("mappolag", "ma_runner_trace.npz"), ("happo", "ma_runner_trace_happo.npz"),
                                        ("macpo", "ma_runner_trace_macpo.npz")])
def test_ma_runner_restatement_vs_reference_runner_trace(golden_dir, algo, fname):
    """Episodes of the reference multi-agent Runner (compute() with PopArt-denormalised masked GAE, then HAPPO-sequential
    train()) replayed through the restatement with the recorded buffers, agent order and shuffles: mappolag, happo (no cost
    side) and macpo (trust-region step, one pass)."""
    from oracle import ma_restatement as MR
    z = _load(golden_dir, fname)
    use_cost = algo in ("mappolag", "macpo")
    A, EP = int(z["meta_agents"]), int(z["meta_episodes"])
    cfg = {k[4:]: float(z[k]) for k in z.files if k.startswith("cfg_")}
    cfg["use_policy_active_masks"] = bool(cfg["use_policy_active_masks"])
    cfg["use_value_active_masks"] = bool(cfg.get("use_value_active_masks", 0))
    H, nb = int(cfg["hidden_size"]), 1 + int(cfg["layer_N"])
    D, S, Ad = z["e0_a0_obs"].shape[-1], z["e0_a0_share_obs"].shape[-1], z["e0_a0_actions"].shape[-1]
    names = ("actor", "critic", "cost_critic") if use_cost else ("actor", "critic")

    def nets(prefix):
        out = {"actor": MR.MANet(D, H, nb, Ad, True, cfg["std_x_coef"], cfg["std_y_coef"]), "critic": MR.MANet(S, H, nb, 1, False)}
        if use_cost:
            out["cost_critic"] = MR.MANet(S, H, nb, 1, False)
        for nm, net in out.items():
            pre = f"{prefix}_{nm}_"
            net.load_reference_state_dict({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
        return out
    trainers = []
    for a in range(A):
        n = nets(f"init_a{a}")
        trainers.append(MR.OracleMATrainer(cfg, n["actor"], n["critic"], n.get("cost_critic"), algo=algo))
    iters = 1 if algo == "macpo" else int(cfg["learning_iters"])
    buf_keys = ["share_obs", "obs", "actions", "action_log_probs", "value_preds", "rewards", "masks", "active_masks"]
    buf_keys += ["cost_preds", "costs"] if use_cost else []
    for e in range(EP):
        bufs = []
        for a in range(A):
            b = {k: torch.from_numpy(z[f"e{e}_a{a}_{k}"].copy()) for k in buf_keys}
            tr = trainers[a]
            # Runner.compute: bootstrap values from the critics, then the masked GAE with PopArt de-normalisation
            with torch.no_grad():
                b["value_preds"][-1] = tr.critic(b["share_obs"][-1])
            b["returns"] = MR.masked_gae(b["rewards"], b["value_preds"], b["masks"], tr.popart, cfg["gamma"], cfg["gae_lambda"])
            np.testing.assert_allclose(b["returns"].numpy()[:-1], z[f"e{e}_a{a}_returns"][:-1], rtol=2e-5, atol=2e-6)
            if use_cost:
                b["aver_episode_costs"] = torch.from_numpy(z[f"e{e}_a{a}_aver_episode_costs"].copy())
                with torch.no_grad():
                    b["cost_preds"][-1] = tr.cost_critic(b["share_obs"][-1])
                b["cost_returns"] = MR.masked_gae(b["costs"], b["cost_preds"], b["masks"], tr.popart, cfg["gamma"], cfg["gae_lambda"])
                np.testing.assert_allclose(b["cost_returns"].numpy()[:-1], z[f"e{e}_a{a}_cost_returns"][:-1], rtol=2e-5, atol=2e-6)
            bufs.append(b)
        order = [int(i) for i in z[f"e{e}_agent_order"]]
        perms_of = {a: [z[f"e{e}_perm{pos * iters + it}"] for it in range(iters)] for pos, a in enumerate(order)}
        rows = np.asarray(MR.runner_train(trainers, bufs, order, perms_of, cfg))
        if algo == "mappolag":
            cols = ((0, "Loss_Loss_reward_critic"), (6, "Loss_Loss_cost_critic"), (2, "Loss_Loss_actor"), (1, "Misc_Reward_critic_norm"),
                    (7, "Misc_Cost_critic_norm"), (3, "Misc_Entropy"), (5, "Misc_Ratio"))
        elif algo == "happo":
            cols = ((0, "Loss_Loss_reward_critic"), (2, "Loss_Loss_actor"), (1, "Misc_Reward_critic_norm"), (3, "Misc_Entropy"),
                    (5, "Misc_Ratio"))
        else:   # macpo rows: value loss, critic norm, kl, improve, expected improve, cost surrogate, cost critic norm, ...
            cols = ((0, "Loss_Loss_reward_critic"), (5, "Loss_Loss_cost_critic"), (3, "Loss_Loss_actor_improve"),
                    (4, "Loss_Loss_actor_expected_improve"), (1, "Misc_Reward_critic_norm"), (6, "Misc_Cost_critic_norm"), (2, "Misc_KL"))
        for col, key in cols:
            np.testing.assert_allclose(rows[:, col], z[f"e{e}_stored_{key}"], rtol=2e-3 if algo == "macpo" else 2e-4, atol=2e-6,
                                       err_msg=f"episode {e} {key}")
        for a in range(A):
            if algo == "mappolag":
                assert float(trainers[a].lamda) == pytest.approx(float(z[f"e{e}_a{a}_lamda"]), rel=1e-5)
            fin = nets(f"e{e}_a{a}_after")
            for nm in names:
                net = getattr(trainers[a], nm)
                np.testing.assert_allclose(net.flat().numpy(), fin[nm].flat().numpy(), rtol=2e-3 if algo == "macpo" else 5e-4,
                                           atol=2e-5 if algo == "macpo" else 5e-6, err_msg=f"e{e} a{a} {nm}")


def test_boundary_logic_matches_trace(golden_dir):
    """a-4: done -> bootstrap 0; epoch end and time-out both end a path (ppo_lag.py:198-234)."""
    z = _load(golden_dir, "ppo_lag_trace.npz")
    from oracle.synth_env import SynthEnv
    env = SynthEnv(int(z["meta_num_envs"]), seed=0, obs_dim=60, act_dim=8, p_term=float(z["meta_env_p_term"]),
                   p_cost=float(z["meta_env_p_cost"]), trunc_len=int(z["meta_env_trunc_len"]))
    env.reset()
    T = int(z["meta_T"])
    for e in range(int(z["meta_epochs"])):
        for t in range(T):
            obs, rew, cost, term, trunc, info = env.step(None)
            dummy = np.full(env.num_envs, 7.0, np.float32)
            seg, br, bc = R.boundary_step(term, trunc, t == T - 1, dummy, dummy, dummy + 1, dummy + 1)
            assert np.array_equal(seg.astype(np.uint8), z[f"e{e}_seg_end"][:, t]), (e, t)
            assert np.all(br[term] == 0) and np.all(br[~seg] == 0), "terminated envs bootstrap with 0"
            assert np.array_equal(rew, z[f"e{e}_raw_reward"][:, t])
            # recorded bootstrap is zero exactly where the episode terminated
            assert np.all(z[f"e{e}_boot_r"][:, t][term] == 0)
            assert np.all(br[trunc] == 8.0) and np.all(br[seg & ~term & ~trunc] == 7.0)


@pytest.mark.parametrize("algo", ["rcpo", "trpo_lag"])
def test_lagrangian_trust_region_traces_multiplier_and_mix(golden_dir, algo):
    """The Lagrangian trust-region siblings (rcpo.py:320-327, trpo_lag.py:320-327): the multiplier the reference logged in
    every epoch follows from the recorded EpCost statistics through the oracle's Lagrange restatement, and it moves."""
    z = _load(golden_dir, f"{algo}_trace.npz")
    lag = R.OracleLagrange(float(z["meta_arg_cost_limit"]), float(z["meta_arg_lagrangian_multiplier_init"]),
                           float(z["meta_arg_lagrangian_multiplier_lr"]))
    lams = []
    for e in range(int(z["meta_epochs"])):
        lag.update_lagrange_multiplier(float(z[f"e{e}_get_stats_Metrics_EpCost"]))
        assert lag.lagrangian_multiplier == pytest.approx(float(z[f"e{e}_row_Train_LagragianMultiplier"]), rel=1e-6)
        lams.append(lag.lagrangian_multiplier)
    assert len(set(round(l, 6) for l in lams)) == len(lams) and all(l > 0 for l in lams)


@pytest.mark.parametrize("algo,line_search", [("natural_pg", False), ("trpo", True), ("rcpo", False), ("trpo_lag", True),
                                              ("trpo_lag_humanoid", True)])
def test_trust_region_restatement_vs_reference_main_trace(golden_dir, algo, line_search):
    """R.trust_region_policy_update (natural_pg.py:350-381, trpo.py:366-428; rcpo / trpo_lag on the mixed advantage) against
    the reference mains' own traces, every epoch from the reference's recorded state: curvature, step length, norms, the KL
    and the loss it logs, the accepted line-search candidate and the actor after the step.  `_humanoid`: trpo_lag.main() with
    ActorVCritic(376, 17)."""
    fname = f"{algo}_trace.npz" if not algo.endswith("_humanoid") else f"{algo[:-9]}_trace_humanoid.npz"
    algo = algo[:-9] if algo.endswith("_humanoid") else algo
    z = _load(golden_dir, fname)
    tkl = float(z["meta_cfg_target_kl"])
    for e in range(int(z["meta_epochs"])):
        pol = _policy_from(z, f"e{e}_sd_before_")
        data, _ = _trace_epoch_inputs(z, e)
        adv = data["adv_r"]
        if algo in ("rcpo", "trpo_lag"):
            adv = R.adv_mix(data["adv_r"], data["adv_c"], float(z[f"e{e}_row_Train_LagragianMultiplier"]))
        out = R.trust_region_policy_update(pol, data, adv, target_kl=tkl, line_search=line_search)
        assert float(out["xHx"]) == pytest.approx(float(z[f"e{e}_Misc_xHx"]), rel=1e-3)
        assert float(out["alpha"]) == pytest.approx(float(z[f"e{e}_Misc_Alpha"]), rel=1e-3)
        assert float(out["x"].norm()) == pytest.approx(float(z[f"e{e}_Misc_H_inv_g"]), rel=1e-3)
        assert float(out["g"].norm()) == pytest.approx(float(z[f"e{e}_Misc_gradient_norm"]), rel=1e-4)
        assert float(out["step_direction"].norm()) == pytest.approx(float(z[f"e{e}_Misc_FinalStepNorm"]), rel=1e-3)
        assert out["kl"] == pytest.approx(float(z[f"e{e}_Train_KL"]), rel=2e-3)
        assert out["loss_actor"] == pytest.approx(float(z[f"e{e}_Loss_Loss_actor"]), rel=1e-4, abs=1e-7)
        if line_search:
            assert out["accept"] == int(z[f"e{e}_Misc_AcceptanceStep"])
        for k, v in pol.actor.state_dict().items():
            np.testing.assert_allclose(v.numpy(), z[f"e{e}_actor_after_{k}"], rtol=1e-3, atol=2e-6, err_msg=f"epoch {e} {k}")


@pytest.mark.parametrize("fname", ["pcpo_trace.npz", "pcpo_trace_humanoid.npz"])
def test_pcpo_restatement_vs_reference_main_trace(golden_dir, fname):
    """R.pcpo_policy_update (pcpo.py:352-470) against the reference's pcpo.main() trace, every epoch from the recorded state
    (incl. the epoch whose line search backtracks six times).  `_humanoid`: pcpo.main() with ActorVCritic(376, 17)."""
    z = _load(golden_dir, fname)
    tkl = float(z["meta_cfg_target_kl"])
    steps = []
    for e in range(int(z["meta_epochs"])):
        pol = _policy_from(z, f"e{e}_sd_before_")
        data, _ = _trace_epoch_inputs(z, e)
        ep_costs = float(z[f"e{e}_get_stats_Metrics_EpCost"]) - float(z["meta_arg_cost_limit"])
        out = R.pcpo_policy_update(pol, data, ep_costs, target_kl=tkl)
        steps.append(out["accept"])
        assert out["accept"] == int(z[f"e{e}_Misc_AcceptanceStep"])
        assert float(out["xHx"]) == pytest.approx(float(z[f"e{e}_Misc_xHx"]), rel=1e-3)
        assert float(out["alpha"]) == pytest.approx(float(z[f"e{e}_Misc_Alpha"]), rel=1e-3)
        assert float(out["x"].norm()) == pytest.approx(float(z[f"e{e}_Misc_H_inv_g"]), rel=1e-3)
        assert float(out["g"].norm()) == pytest.approx(float(z[f"e{e}_Misc_gradient_norm"]), rel=1e-4)
        assert float(out["step_direction"].norm()) == pytest.approx(float(z[f"e{e}_Misc_FinalStepNorm"]), rel=1e-3)
        assert out["kl"] == pytest.approx(float(z[f"e{e}_Train_KL"]), rel=2e-3)
        assert out["loss_r_before"] + out["loss_c_before"] == pytest.approx(float(z[f"e{e}_Loss_Loss_actor"]), rel=1e-4, abs=1e-7)
        for k, v in pol.actor.state_dict().items():
            np.testing.assert_allclose(v.numpy(), z[f"e{e}_actor_after_{k}"], rtol=1e-3, atol=2e-6, err_msg=f"epoch {e} {k}")
    if fname == "pcpo_trace.npz":
        assert max(steps) > 1


def test_running_mean_std_restatement_vs_independent_two_pass():
    """a-2 (parity unpinned by the reference: gymnasium is not vendored / installed).  The restated RunningMeanStd -- the
    incremental parallel-variance merge, mean 0 / var 1 / count 1e-4 prior -- is pinned here against an
    INDEPENDENT float64 evaluation written from the definition: the prior is a pseudo-batch of weight 1e-4 with mean 0 and
    variance 1; merging batches must equal the weighted two-pass mean / variance of everything seen so far."""
    rng = np.random.default_rng(3)
    D = 7
    rms = R.RunningMeanStd((D,))
    seen = []
    for n in (5, 1, 1000, 64, 3):
        x = rng.standard_normal((n, D)) * (1.0 + np.arange(D)) + 10.0 * np.arange(D)
        rms.update(x)
        seen.append(x)
        allx = np.concatenate(seen, 0)
        w0 = 1e-4
        tot = w0 + allx.shape[0]
        mean = allx.sum(0) / tot                                   # prior mean 0 contributes nothing to the sum
        var = (w0 * (1.0 + mean ** 2) + ((allx - mean) ** 2).sum(0)) / tot
        np.testing.assert_allclose(rms.mean, mean, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(rms.var, var, rtol=1e-10)
        assert rms.count == pytest.approx(tot, rel=1e-15)
    # normalize() = update with the batch, then (x - mean) / sqrt(var + 1e-8) with the UPDATED statistics
    y = rng.standard_normal((4, D))
    out = rms.normalize(y.copy())
    allx = np.concatenate(seen + [y], 0)
    tot = 1e-4 + allx.shape[0]
    mean = allx.sum(0) / tot
    var = (1e-4 * (1.0 + mean ** 2) + ((allx - mean) ** 2).sum(0)) / tot
    np.testing.assert_allclose(out, (y - mean) / np.sqrt(var + 1e-8), rtol=1e-9, atol=1e-12)
