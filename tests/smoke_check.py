"""Smoke / parity check of one tiny collect -> GAE -> PPO-Lagrangian update on cuda:0 against the CPU oracle.

TEST INFRASTRUCTURE (this file is under tests/): called by __graft_entry__.smoke() and tests/test_gpu_parity.py, the
only places allowed to use oracle/ as a checker.  The product package (safe-policy-optimization_amd/safepo) never imports
the oracle."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from safepo import _abi  # noqa: E402
from safepo.common.engine import PPOLagEngine  # noqa: E402
from safepo.common.model import ActorVCritic  # noqa: E402


def smoke_check(verbose: bool = False, num_envs: int = 8, steps: int = 32, seed: int = 0) -> None:
    """Tiny collect -> GAE -> update on cuda:0, compared with oracle/restatement.py."""
    from oracle import restatement as R
    from oracle.synth_env import SynthEnv

    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    D, A = 60, 8
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 0.02, "batch_size": 64, "learning_iters": 2,
           "max_grad_norm": 40.0}
    policy = ActorVCritic(D, A).to(dev)
    ref = R.OraclePolicy(D, A)
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in policy.state_dict().items()})
    eng = PPOLagEngine(policy, num_envs, steps, cfg, dev)
    env = SynthEnv(num_envs, D, A, seed=seed, p_term=0.05, trunc_len=10)
    obs_h, _ = env.reset()
    obs = torch.as_tensor(obs_h, device=dev)
    gen = torch.Generator().manual_seed(seed + 1)
    rec = {k: [] for k in ("obs", "eps", "reward", "cost", "seg", "boot_r", "boot_c", "act", "logp", "v_r", "v_c")}
    for t in range(steps):
        eps = torch.randn((num_envs, A), generator=gen)
        act = eng.collect_step(t, obs, eps.to(dev))
        with torch.no_grad():
            a_ref, lp_ref, vr_ref, vc_ref = ref.step_with_eps(torch.as_tensor(obs_h), eps)
        nobs, rew, cost, term, trunc, info = env.step(act.cpu().numpy())
        fo = None
        vfr = vfc = np.zeros(num_envs, np.float32)
        if "final_observation" in info:
            fo_h = np.stack([a if a is not None else np.zeros(D, np.float32) for a in info["final_observation"]])
            fo = torch.as_tensor(fo_h, dtype=torch.float32, device=dev)
            with torch.no_grad():
                vfr, vfc = ref.reward_critic(torch.as_tensor(fo_h)).numpy(), ref.cost_critic(torch.as_tensor(fo_h)).numpy()
        with torch.no_grad():
            vnr, vnc = ref.reward_critic(torch.as_tensor(nobs)).numpy(), ref.cost_critic(torch.as_tensor(nobs)).numpy()
        seg, br, bc = R.boundary_step(term, trunc, t == steps - 1, vnr, vnc, vfr, vfc)
        eng.post_step(t, torch.as_tensor(nobs, device=dev), torch.as_tensor(rew, device=dev),
                      torch.as_tensor(cost, device=dev), torch.as_tensor(term, dtype=torch.float32, device=dev),
                      torch.as_tensor(trunc, dtype=torch.float32, device=dev), fo)
        for k, v in (("obs", obs_h), ("eps", eps.numpy()), ("reward", rew), ("cost", cost), ("seg", seg),
                     ("boot_r", br), ("boot_c", bc), ("act", a_ref.numpy()), ("logp", lp_ref.numpy()),
                     ("v_r", vr_ref.numpy()), ("v_c", vc_ref.numpy())):
            rec[k].append(np.asarray(v))
        obs_h, obs = nobs, torch.as_tensor(nobs, device=dev)
    st = {k: np.stack(v, 1) for k, v in rec.items()}          # [N, T, ...]
    b = eng.buffer
    assert np.array_equal(b.seg_end.cpu().numpy(), st["seg"].astype(np.uint8)), "segment mask differs"
    np.testing.assert_allclose(b.data["value_r"].cpu().numpy(), st["v_r"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.data["log_prob"].cpu().numpy(), st["logp"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(b.boot_r.cpu().numpy(), st["boot_r"], rtol=1e-4, atol=1e-5)
    # GAE against the oracle on the DEVICE-produced inputs (bit pattern)
    inp = {k: b.data[k].cpu().numpy() for k in ("reward", "cost", "value_r", "value_c")}
    o = R.gae_dense(inp["reward"], inp["cost"], inp["value_r"], inp["value_c"], b.seg_end.cpu().numpy(),
                    b.boot_r.cpu().numpy(), b.boot_c.cpu().numpy(), 0.99, 0.95, 0.95)
    lam = 0.37
    out = eng.update(lam, perm_fn=lambda it: torch.arange(eng.M - 1, -1, -1, device=dev, dtype=torch.int32))
    # (update() ran the GAE kernel first; raw advantages were standardised in place -> recompute reference)
    sr, sc = R.adv_standardize(torch.from_numpy(o[0].reshape(-1)), torch.from_numpy(o[1].reshape(-1)))
    np.testing.assert_allclose(b.data["adv_r"].cpu().numpy().reshape(-1), sr.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(b.data["adv_c"].cpu().numpy().reshape(-1), sc.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_array_equal(b.data["target_value_r"].cpu().numpy(), o[2])
    np.testing.assert_array_equal(b.data["target_value_c"].cpu().numpy(), o[3])       # the cost side of the same scan
    np.testing.assert_allclose(b.boot_c.cpu().numpy(), st["boot_c"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.data["value_c"].cpu().numpy(), st["v_c"], rtol=1e-4, atol=1e-5)
    assert b.last_scan_folded, "the engine path must run the folded form of the scan (spo_boundary_step_fold)"
    # oracle update on the same data / same shuffles
    M = eng.M
    data = {"obs": torch.from_numpy(st["obs"].reshape(M, D)), "act": b.data["act"].cpu().reshape(M, A),
            "log_prob": b.data["log_prob"].cpu().reshape(M), "target_value_r": torch.from_numpy(o[2].reshape(M)),
            "target_value_c": torch.from_numpy(o[3].reshape(M)), "adv_r": sr, "adv_c": sc}
    upd = R.PPOLagUpdater(ref, epochs=1)
    perms = [np.arange(M - 1, -1, -1)] * 2
    ro = R.ppo_lag_update(ref, upd, data, lam, perms, learning_iters=2, batch_size=64, target_kl=0.02)
    got = torch.cat(out["losses"], 0).cpu().numpy()
    np.testing.assert_allclose(got, ro["losses"][:len(got)], rtol=2e-4, atol=2e-6)
    th, tr = policy.theta.cpu().numpy().astype(np.float64), R.flat_params(ref).numpy().astype(np.float64)
    bad = np.abs(th - tr) > (2e-6 + 1e-3 * np.abs(tr))       # Adam amplifies noise-level gradients: allow 0.1 % outliers
    assert bad.mean() <= 1e-3 and np.abs(th - tr).max() <= 2 * 3e-4 * len(got), (bad.sum(), np.abs(th - tr).max())
    assert abs(out["kl"] - ro["kl"]) <= 1e-4 * max(1.0, abs(ro["kl"])) + 1e-7
    if verbose:
        print(f"smoke: seg/boot/GAE/update parity ok; kl={out['kl']:.3e} (oracle {ro['kl']:.3e}), "
              f"loss_pi={out['loss_pi']:.4f}")
