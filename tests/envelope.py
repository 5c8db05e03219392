"""Drift-envelope checker for optimiser trajectories (test infrastructure; uses the CPU oracle).

A chain of Adam steps is not contractive: two correct fp32 implementations of the same minibatch
sequence (ppo_lag.py:297-336) drift apart, because 1/sqrt(v) amplifies rounding-level gradient
differences.  A blanket rtol along a trajectory therefore cannot tell rounding from a slow kernel
bug.  This module measures the drift against a yardstick instead:

    T64 = the oracle evaluated in float64 on the same inputs / shuffles (the "true" trajectory),
    T32 = the reference arithmetic (torch fp32 on CPU: the oracle, or a recorded reference trace),
    TH  = the HIP path,

and asserts  dist(TH, T64) <= c * dist(T32, T64) + floor  for the per-minibatch losses (RMS over
windows of consecutive steps) and for the parameter vector at checkpoints (L2 norm and max-abs).
If the HIP kernel's deviation were anything but rounding (a stale prefetch, a wrong parity buffer, a
mis-applied speculative Adam), its distance from T64 would exceed the fp32 reference's own distance by
orders of magnitude: the reference's distance is ~1e-7 relative per step.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import restatement as R


def oracle_trajectory(state_dict, problem, perm, batch, nsteps, dtype, checkpoints=(), lr=3e-4, max_grad_norm=40.0,
                      lr_factor=1.0, threads=4, hidden_sizes=(64, 64), **loss_kw):
    """Runs `nsteps` consecutive minibatch steps of the oracle (R.PPOLagUpdater.minibatch_step, i.e.
    ppo_lag.py:306-329) in `dtype` from `state_dict` over consecutive chunks of `perm`.
    Returns (losses [nsteps,3] float64, {k: flat theta after k steps, float64})."""
    torch.set_num_threads(threads)
    obs, act, logp, tgt_r, tgt_c, adv = [t.to(dtype) for t in problem]
    D, A = obs.shape[1], act.shape[1]
    pol = R.OraclePolicy(D, A, tuple(hidden_sizes))
    pol.load_state_dict({k: v.detach().cpu().clone() for k, v in state_dict.items()})
    pol = pol.to(dtype)
    upd = R.PPOLagUpdater(pol, epochs=1, lr=lr, max_grad_norm=max_grad_norm, **loss_kw)
    for g in upd.opt_a.param_groups:
        g["lr"] = lr * lr_factor
    perm = torch.as_tensor(perm, dtype=torch.long)
    cps = set(int(k) for k in checkpoints)
    thetas, losses = {}, np.zeros((nsteps, 3))
    for s in range(nsteps):
        ii = perm[s * batch:(s + 1) * batch]
        losses[s] = upd.minibatch_step(obs[ii], act[ii], logp[ii], tgt_r[ii], tgt_c[ii], adv[ii])
        if (s + 1) in cps:
            thetas[s + 1] = R.flat_params(pol).double().numpy().copy()
    return losses, thetas


def _windows(n, w):
    edges = list(range(0, n, w))
    return [(a, min(a + w, n)) for a in edges]


def loss_envelope(hip, f32, f64, c=3.0, floor_rel=3e-7, window=64):
    """Per window of `window` consecutive steps and per loss column: RMS(hip - f64) against
    c * RMS(f32 - f64) + floor_rel * RMS(f64).  Returns the worst ratio lhs / rhs and its location."""
    hip, f32, f64 = (np.asarray(x, np.float64) for x in (hip, f32, f64))
    worst = (0.0, None)
    for a, b in _windows(len(f64), window):
        for col in range(f64.shape[1]):
            dh = np.sqrt(np.mean((hip[a:b, col] - f64[a:b, col]) ** 2))
            d32 = np.sqrt(np.mean((f32[a:b, col] - f64[a:b, col]) ** 2))
            scale = np.sqrt(np.mean(f64[a:b, col] ** 2))
            ratio = dh / (c * d32 + floor_rel * scale + 1e-30)
            if ratio > worst[0]:
                worst = (ratio, (a, b, col, dh, d32, scale))
    return worst


def adam_noise_directions(dh, d32, frac=1e-4, factor=30.0, slack=10.0):
    """Elements on which THE REFERENCE ITSELF is an outlier against float64: Adam divides by sqrt(v) + eps, so an element whose
    gradient is of the size of eps (1e-8) turns rounding noise of the gradient into steps of either sign -- in the reference's
    float32 as much as anywhere (measured on the reference's own ppo_lag.main() at 376 / 17, tests/golden/ppo_lag_trace_humanoid:
    ONE of 86 116 parameters, a first-layer weight of the cost critic, sits 4.7e-6 from float64 in the reference's recorded
    run after 6 steps where every other element sits ~1e-8; the launch-per-layer HIP path lands 9.2e-6 away on it, the feature-split
    kernel 2.8e-5).  Such an element says nothing about rounding quality, and one of them can dominate an L2 norm.  They are
    identified by the reference's behaviour alone -- at most `frac` of the elements, each at least `factor` x the RMS distance of
    the rest -- gated on their own (|hip - f64| <= slack x the reference's distance on THAT element) and left out of the norms.
    Returns (mask, ok)."""
    n = d32.size
    k = max(1, int(np.ceil(frac * n)))
    order = np.argsort(d32)
    typ = float(np.sqrt(np.mean(d32[order[:n - k]] ** 2))) if n > k else 0.0
    mask = np.zeros(n, bool)
    cand = order[n - k:]
    mask[cand] = d32[cand] > factor * max(typ, 1e-12)
    return mask, bool((dh[mask] <= slack * d32[mask]).all())


def theta_envelope(hip, f32, f64, c=3.0, floor_abs=2e-7, floor_abs_max=None, noise_directions=False):
    """L2 and max-abs distance of the parameter vector from the fp64 trajectory against c x the fp32 reference's own
    distance (+ floor_abs per element: half an fp32 ulp of an O(1) parameter; floor_abs_max: a separate floor for the max-norm
    gate, see tests/test_gpu_wide_dims.py::_theta_floor).  noise_directions: adam_noise_directions() first.  Returns the worse
    of the two ratios."""
    hip, f32, f64 = (np.asarray(x, np.float64).reshape(-1) for x in (hip, f32, f64))
    info = {}
    if noise_directions:
        # round 6: the exemption is taken only where the gate fails without it, and the measured numbers are printed
        strict, _ = theta_envelope(hip, f32, f64, c=c, floor_abs=floor_abs, floor_abs_max=floor_abs_max, noise_directions=False)
        if strict <= 1.0:
            return strict, {"noise_directions": 0, "strict_ratio": strict}
        mask, ok = adam_noise_directions(np.abs(hip - f64), np.abs(f32 - f64))
        info["noise_directions"] = int(mask.sum())
        info["strict_ratio"] = strict
        worst = float((np.abs(hip - f64)[mask] / np.abs(f32 - f64)[mask]).max()) if mask.any() else 0.0
        print(f"[envelope] Adam-noise-direction exemption IN USE: ratio without it {strict:.2f}; {int(mask.sum())} of {hip.size} "
              f"elements singled out by the reference's own distance, worst |hip-f64| / |f32-f64| on them {worst:.2f} (gate 10)")
        if not ok:
            return float("inf"), {"noise_directions": int(mask.sum()), "worst": worst}
        hip, f32, f64 = hip[~mask], f32[~mask], f64[~mask]
    n = f64.size
    dh2, d322 = np.linalg.norm(hip - f64), np.linalg.norm(f32 - f64)
    dhm, d32m = np.abs(hip - f64).max(), np.abs(f32 - f64).max()
    r2 = dh2 / (c * d322 + floor_abs * np.sqrt(n))
    rm = dhm / (c * d32m + (floor_abs if floor_abs_max is None else floor_abs_max))
    return max(r2, rm), dict(info, l2_hip=dh2, l2_f32=d322, max_hip=dhm, max_f32=d32m)


def assert_loss_envelope(hip, f32, f64, what, **kw):
    ratio, where = loss_envelope(hip, f32, f64, **kw)
    assert ratio <= 1.0, (f"{what}: HIP losses leave the rounding envelope around the fp64 trajectory: ratio {ratio:.2f} at "
                          f"(steps {where[0]}..{where[1]}, column {where[2]}): rms|hip-f64|={where[3]:.3e}, "
                          f"rms|f32-f64|={where[4]:.3e}, rms|f64|={where[5]:.3e}")
    return ratio


def assert_theta_envelope(hip, f32, f64, what, **kw):
    ratio, info = theta_envelope(hip, f32, f64, **kw)
    assert ratio <= 1.0, f"{what}: HIP parameters leave the rounding envelope around the fp64 trajectory: ratio {ratio:.2f}, {info}"
    return ratio, info


def permuted_hidden_state(state_dict, seed=0):
    """The same function with the hidden units of every MLP renumbered: an independent but equally valid fp32
    evaluation order (dot products are summed in a different order).  Used on CPU to calibrate the envelope: it plays the
    part of "another correct fp32 implementation".  Returns (state_dict', unpermute(flat theta') -> flat theta)."""
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.detach().cpu().clone() for k, v in state_dict.items()}
    perms = {}
    for net, pre in (("reward_critic", "reward_critic.critic"), ("cost_critic", "cost_critic.critic"), ("actor", "actor.mean")):
        p1 = torch.randperm(sd[f"{pre}.0.weight"].shape[0], generator=g)
        p2 = torch.randperm(sd[f"{pre}.2.weight"].shape[0], generator=g)
        perms[pre] = (p1, p2)
        sd[f"{pre}.0.weight"] = sd[f"{pre}.0.weight"][p1]
        sd[f"{pre}.0.bias"] = sd[f"{pre}.0.bias"][p1]
        sd[f"{pre}.2.weight"] = sd[f"{pre}.2.weight"][p2][:, p1]
        sd[f"{pre}.2.bias"] = sd[f"{pre}.2.bias"][p2]
        sd[f"{pre}.4.weight"] = sd[f"{pre}.4.weight"][:, p2]
    shapes = [(k, tuple(v.shape)) for k, v in sd.items()]

    def unpermute(flat):
        flat = np.asarray(flat)
        out, off = [], 0
        parts = {}
        for k, shp in shapes:
            n = int(np.prod(shp))
            parts[k] = flat[off:off + n].reshape(shp)
            off += n
        for pre, (p1, p2) in perms.items():
            i1, i2 = np.argsort(p1.numpy()), np.argsort(p2.numpy())
            parts[f"{pre}.0.weight"] = parts[f"{pre}.0.weight"][i1]
            parts[f"{pre}.0.bias"] = parts[f"{pre}.0.bias"][i1]
            parts[f"{pre}.2.weight"] = parts[f"{pre}.2.weight"][:, i1][i2]
            parts[f"{pre}.2.bias"] = parts[f"{pre}.2.bias"][i2]
            parts[f"{pre}.4.weight"] = parts[f"{pre}.4.weight"][:, i2]
        for k, _ in shapes:
            out.append(parts[k].reshape(-1))
        return np.concatenate(out)
    return sd, unpermute


def replay_ppo_lag_trace(z, dtype=torch.float64):
    """Replays the recorded epochs of the reference's ppo_lag.main() (tests/golden/ppo_lag_trace.npz: buffers, shuffles,
    multipliers, initial weights, number of learning iterations actually run) through the oracle in `dtype`.  The inputs
    are the reference's recorded fp32 tensors (get() output); only the update arithmetic changes precision.
    Returns {"losses": [per epoch [steps,3]], "theta_before": [per epoch flat], "theta_final": flat} (float64 numpy)."""
    epochs = int(z["meta_epochs"])
    names = [k[len("init_sd_"):] for k in z.files if k.startswith("init_sd_")]
    D, A = z["init_sd_actor.mean.0.weight"].shape[1], z["init_sd_actor.log_std"].shape[0]
    pol = R.OraclePolicy(D, A)
    pol.load_state_dict({k: torch.from_numpy(z["init_sd_" + k].copy()) for k in names})
    pol = pol.to(dtype)
    upd = R.PPOLagUpdater(pol, epochs=epochs, max_grad_norm=float(z["meta_cfg_max_grad_norm"]))
    out = {"losses": [], "theta_before": []}
    for e in range(epochs):
        out["theta_before"].append(R.flat_params(pol).double().numpy().copy())
        raw = lambda k: z[f"e{e}_raw_{k}"]
        N, T = raw("reward").shape
        flat = lambda k: torch.from_numpy(raw(k).reshape(N * T, *raw(k).shape[2:])).to(dtype)
        data = {"obs": flat("obs"), "act": flat("act"), "log_prob": flat("log_prob"),
                "target_value_r": flat("target_value_r"), "target_value_c": flat("target_value_c"),
                "adv_r": torch.from_numpy(z[f"e{e}_get_adv_r"]).to(dtype), "adv_c": torch.from_numpy(z[f"e{e}_get_adv_c"]).to(dtype)}
        n_perm = len([k for k in z.files if k.startswith(f"e{e}_perm")])
        perms = [z[f"e{e}_perm{i}"] for i in range(n_perm)]
        lam = float(z[f"e{e}_row_Train_LagragianMultiplier"])
        res = R.ppo_lag_update(pol, upd, data, lam, perms, learning_iters=n_perm, batch_size=int(z[f"e{e}_batch_size"]),
                               target_kl=float("inf"))
        out["losses"].append(np.asarray(res["losses"], np.float64))
        out.setdefault("kl", []).append(float(res["kl"]))
    out["theta_final"] = R.flat_params(pol).double().numpy().copy()
    return out


# ------------------------------------------------------------------------------------------------------------------------
# Round 5: the remaining trace replays under the same yardstick
# ------------------------------------------------------------------------------------------------------------------------
def theta_floor(lr, nsteps):
    """Max-norm floor for parameter gates after `nsteps` Adam steps at wide input layers (tests/test_gpu_wide_dims.py::_theta_floor:
    half an ulp of an O(1) parameter + 0.3 % of the distance Adam can move an element in that many steps; the L2 gate keeps the
    plain floor)."""
    return 2e-7 + 3e-3 * lr * nsteps


def gate_array(hip, f32, f64, what, rel_floor=1e-6, c=3.0, noise_directions=False):
    """|hip - f64| <= c |f32 - f64| + rel_floor * max|f64| in max-norm and in L2 (per sqrt(n)); noise_directions:
    adam_noise_directions() first (parameters after Adam steps at wide input layers)."""
    hip, f32, f64 = (np.asarray(t, np.float64).reshape(-1) for t in (hip, f32, f64))
    assert hip.shape == f32.shape == f64.shape, (what, hip.shape, f32.shape, f64.shape)
    assert np.isfinite(hip).all(), f"{what}: non-finite HIP values"
    if noise_directions:
        # round 6: first the gate WITHOUT the exemption and at the narrow kernels' floor (1e-6); the exemption and the
        # wider floor are taken only where that fails, with the measured numbers in the test log
        sc0 = max(float(np.abs(f64).max()), 1e-30)
        dh0, d320 = np.abs(hip - f64), np.abs(f32 - f64)
        n0 = np.sqrt(hip.size)
        strict = max(dh0.max() / (c * d320.max() + 1e-6 * sc0), np.linalg.norm(dh0) / (c * np.linalg.norm(d320) + 1e-6 * sc0 * n0))
        if strict <= 1.0:
            return float(dh0.max() / sc0), float(d320.max() / sc0)
        mask, ok = adam_noise_directions(dh0, d320)
        worst = float((dh0[mask] / d320[mask]).max()) if mask.any() else 0.0
        print(f"[envelope] {what}: exemption IN USE (rel_floor {rel_floor:g}): ratio without it at 1e-6 {strict:.2f}; {int(mask.sum())} of "
              f"{hip.size} elements singled out by the reference's own distance, worst |hip-f64| / |f32-f64| on them {worst:.2f} (gate 10)")
        assert ok, f"{what}: an Adam noise direction of the reference is more than 10 x further from float64 on the HIP path"
        hip, f32, f64 = hip[~mask], f32[~mask], f64[~mask]
    sc = max(float(np.abs(f64).max()), 1e-30)
    dh, d32 = np.abs(hip - f64), np.abs(f32 - f64)
    assert dh.max() <= c * d32.max() + rel_floor * sc, \
        f"{what}: max |hip - f64| {dh.max():.3e} > {c} x |f32 - f64| {d32.max():.3e} + {rel_floor:g} x {sc:.3e}"
    n = np.sqrt(hip.size)
    assert np.linalg.norm(dh) <= c * np.linalg.norm(d32) + rel_floor * sc * n, \
        f"{what}: L2 |hip - f64| {np.linalg.norm(dh):.3e} > {c} x |f32 - f64| {np.linalg.norm(d32):.3e} + floor"
    return float(dh.max() / sc), float(d32.max() / sc)


def gate_scalars(rows, what, rel_floor=1e-6, c=3.0, scale=None):
    """rows: [(name, hip, f32, f64)] of scalars of ONE kind of quantity (e.g. the curvature x^T H x of every epoch of a trace).
    A single scalar's |f32 - f64| is one draw of the reference's rounding noise -- it can be 0 by chance --, so the yardstick is
    the largest RELATIVE deviation the reference shows over the group: |hip - f64| / |f64| <= c * max_i(|f32_i - f64_i| / |f64_i|)
    + rel_floor for every member.  `scale`: measure the deviations against this magnitude instead of |f64| -- for a quantity
    that is a cancelling sum (a surrogate loss -mean(ratio * adv) over standardised advantages is ~0 +- rounding: relative to
    itself every evaluation is 100 % off).  Returns (worst hip, yardstick) deviations."""
    rel = lambda a, b: abs(a - b) / (max(abs(b), 1e-30) if scale is None else scale)
    yard = max(rel(f32, f64) for _, _, f32, f64 in rows)
    worst = 0.0
    for name, hip, f32, f64 in rows:
        d = rel(hip, f64)
        assert np.isfinite(hip) and d <= c * yard + rel_floor, \
            f"{what} / {name}: |hip - f64| / |f64| = {d:.3e} > {c} x {yard:.3e} (largest deviation of the reference's fp32 in the group) + {rel_floor:g}; hip {hip!r} f32 {f32!r} f64 {f64!r}"
        worst = max(worst, d)
    return worst, yard


def _policy_in(z, prefix, dtype, only=None):
    names = [k[len(prefix):] for k in z.files if k.startswith(prefix)]
    D, A = z[prefix + "actor.mean.0.weight"].shape[1], z[prefix + "actor.log_std"].shape[0]
    pol = R.OraclePolicy(D, A)
    pol.load_state_dict({k: torch.from_numpy(z[prefix + k].copy()) for k in names})
    return pol.to(dtype)


def _trace_epoch_data(z, e, dtype):
    """The get() dict of epoch e as the reference saw it: recorded rollout tensors, the oracle's own (bit-pinned) GAE and
    standardisation of them (tests/test_oracle_golden.py), cast to `dtype`.  Returns (data, shuffles)."""
    raw = lambda k: z[f"e{e}_raw_{k}"]
    N, T = raw("reward").shape
    adv_r, adv_c, tgt_r, tgt_c = R.gae_dense(raw("reward"), raw("cost"), raw("value_r"), raw("value_c"), z[f"e{e}_seg_end"],
                                              z[f"e{e}_boot_r"], z[f"e{e}_boot_c"], float(z["meta_cfg_gamma"]), 0.95, 0.95)
    sr, sc = R.adv_standardize(torch.from_numpy(adv_r.reshape(-1)), torch.from_numpy(adv_c.reshape(-1)))
    flat = lambda k: torch.from_numpy(raw(k).reshape(N * T, *raw(k).shape[2:]))
    data = {"obs": flat("obs"), "act": flat("act"), "log_prob": flat("log_prob"), "target_value_r": torch.from_numpy(tgt_r.reshape(-1)),
            "target_value_c": torch.from_numpy(tgt_c.reshape(-1)), "adv_r": sr, "adv_c": sc}
    n_perm = len([k for k in z.files if k.startswith(f"e{e}_perm")])
    return {k: v.to(dtype) for k, v in data.items()}, [z[f"e{e}_perm{i}"] for i in range(n_perm)]


def replay_second_order_trace(z, algo, dtype=torch.float64):
    """The trust-region family (cpo, pcpo, natural_pg, trpo, rcpo, trpo_lag) over the epochs of the reference's main() trace in
    `dtype`, the way the HIP tests replay it: the ACTOR starts every epoch from the reference's recorded parameters (its step has
    no optimiser state: the epochs are independent), the two CRITICS and their Adam state run free from the initial state through
    all epochs (cpo.py:534-571 keeps the optimisers), clipped jointly with the stale actor gradient the step left behind.
    Returns one dict per epoch: the logged scalars, the actor after the step, the critic-fit losses, the critics' parameters
    before the epoch; plus the critics at the end."""
    epochs, tkl = int(z["meta_epochs"]), float(z["meta_cfg_target_kl"])
    iters = int(z["meta_cfg_learning_iters"])
    pol = _policy_in(z, "init_sd_", dtype)
    fit = R.CriticFitter(pol)
    actor_keys = list(pol.actor.state_dict())
    crit = lambda: torch.cat([p.detach().reshape(-1) for p in list(pol.reward_critic.parameters()) + list(pol.cost_critic.parameters())]).double().numpy().copy()
    lagrange = None
    if algo in ("rcpo", "trpo_lag"):
        lagrange = R.OracleLagrange(float(z["meta_arg_cost_limit"]), float(z["meta_arg_lagrangian_multiplier_init"]),
                                    float(z["meta_arg_lagrangian_multiplier_lr"]))
    out = []
    for e in range(epochs):
        pol.actor.load_state_dict({k: torch.from_numpy(z[f"e{e}_sd_before_actor.{k}"].copy()).to(dtype) for k in actor_keys})
        data, perms = _trace_epoch_data(z, e, dtype)
        rec = {"critics_before": crit()}
        if algo == "cpo":
            ep_costs = float(z[f"e{e}_get_stats_Metrics_EpCost"]) - float(z["meta_arg_cost_limit"])
            o = R.cpo_policy_update(pol, data, ep_costs, target_kl=tkl)
            stale, step_norm, loss_actor = o["b"], float(o["step_direction"].norm()), o["loss_r_before"] + o["loss_c_before"]
        elif algo == "pcpo":
            ep_costs = float(z[f"e{e}_get_stats_Metrics_EpCost"]) - float(z["meta_arg_cost_limit"])
            o = R.pcpo_policy_update(pol, data, ep_costs, target_kl=tkl)
            stale, step_norm, loss_actor = o["b"], float(o["step_direction"].norm()), o["loss_r_before"] + o["loss_c_before"]
        else:
            adv = data["adv_r"]
            if lagrange is not None:
                lagrange.update_lagrange_multiplier(float(z[f"e{e}_get_stats_Metrics_EpCost"]))
                adv = R.adv_mix(data["adv_r"], data["adv_c"], lagrange.lagrangian_multiplier)
            o = R.trust_region_policy_update(pol, data, adv, target_kl=tkl, line_search=algo in ("trpo", "trpo_lag"))
            stale, step_norm, loss_actor = -o["g"], float(o["step_direction"].norm()), o["loss_actor"]
        rec.update(xHx=float(o["xHx"]), alpha=float(o["alpha"]), gradient_norm=float(o["g"].norm()), H_inv_g=float(o["x"].norm()),
                   final_step_norm=step_norm, kl=float(o["kl"]), loss_actor=float(loss_actor), accept=o["accept"], case=o.get("case"),
                   actor_after=R.actor_flat_params(pol.actor).double().numpy().copy())
        i = 0
        for _, prm in pol.actor.named_parameters():            # the stale actor gradient takes part in the critics' joint clip
            prm.grad = stale[i:i + prm.numel()].view(prm.shape).clone()
            i += prm.numel()
        bs, M, losses = int(z[f"e{e}_batch_size"]), data["obs"].shape[0], []
        for it in range(iters):
            perm = torch.from_numpy(np.asarray(perms[it]).astype(np.int64))
            for s in range(0, M, bs):
                idx = perm[s:s + bs]
                losses.append(fit.minibatch_step(data["obs"][idx], data["target_value_r"][idx], data["target_value_c"][idx]))
        rec["critic_losses"] = np.asarray(losses, np.float64)
        out.append(rec)
    return out, crit()


def replay_kl_penalty_trace(z, algo, dtype=torch.float64):
    """focops.main() / cup.main() of the reference replayed through the oracle in `dtype` (free-running from the initial
    weights: every network has Adam state): per epoch the per-minibatch losses, the stop iterations, the KL and the flat
    parameters before the epoch; plus the final parameters.  Mirrors tests/test_oracle_golden.py::test_kl_penalty_family_main_trace."""
    epochs, iters = int(z["meta_epochs"]), int(z["meta_cfg_learning_iters"])
    pol = _policy_in(z, "init_sd_", dtype)
    upd = R.KLPenaltyUpdater(pol, epochs=epochs)
    lag = R.OracleLagrange(float(z["meta_arg_cost_limit"]), float(z["meta_arg_lagrangian_multiplier_init"]),
                           float(z["meta_arg_lagrangian_multiplier_lr"]), lagrangian_upper_bound=2.0 if algo == "focops" else 0.2)
    out = []
    for e in range(epochs):
        rec = {"theta_before": R.flat_params(pol).double().numpy().copy()}
        data, perms = _trace_epoch_data(z, e, dtype)
        lag.update_lagrange_multiplier(float(z[f"e{e}_get_stats_Metrics_EpCost"]))
        kw = dict(learning_iters=iters, batch_size=int(z[f"e{e}_batch_size"]), target_kl=float(z["meta_cfg_target_kl"]))
        perms = perms + [perms[-1]] * (2 * iters)
        if algo == "focops":
            o = R.focops_update(pol, upd, data, lag.lagrangian_multiplier, perms, **kw)
        else:
            o = R.cup_update(pol, upd, data, lag.lagrangian_multiplier, perms, float(z["meta_cfg_gamma"]), **kw)
            rec["second_stage_stop_iter"] = o["second_stage_stop_iter"]
        rec.update(losses=np.asarray(o["losses"], np.float64), stop_iter=o["stop_iter"], kl=float(o["kl"]))
        out.append(rec)
    return out, R.flat_params(pol).double().numpy().copy()
