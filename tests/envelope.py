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
                      lr_factor=1.0, threads=4, **loss_kw):
    """Runs `nsteps` consecutive minibatch steps of the oracle (R.PPOLagUpdater.minibatch_step, i.e.
    ppo_lag.py:306-329) in `dtype` from `state_dict` over consecutive chunks of `perm`.
    Returns (losses [nsteps,3] float64, {k: flat theta after k steps, float64})."""
    torch.set_num_threads(threads)
    obs, act, logp, tgt_r, tgt_c, adv = [t.to(dtype) for t in problem]
    D, A = obs.shape[1], act.shape[1]
    pol = R.OraclePolicy(D, A)
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


def theta_envelope(hip, f32, f64, c=3.0, floor_abs=2e-7):
    """L2 and max-abs distance of the parameter vector from the fp64 trajectory against c x the fp32 reference's own
    distance (+ floor_abs per element: half an fp32 ulp of an O(1) parameter).  Returns the worse of the two ratios."""
    hip, f32, f64 = (np.asarray(x, np.float64).reshape(-1) for x in (hip, f32, f64))
    n = f64.size
    dh2, d322 = np.linalg.norm(hip - f64), np.linalg.norm(f32 - f64)
    dhm, d32m = np.abs(hip - f64).max(), np.abs(f32 - f64).max()
    r2 = dh2 / (c * d322 + floor_abs * np.sqrt(n))
    rm = dhm / (c * d32m + floor_abs)
    return max(r2, rm), {"l2_hip": dh2, "l2_f32": d322, "max_hip": dhm, "max_f32": d32m}


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
    out["theta_final"] = R.flat_params(pol).double().numpy().copy()
    return out
