"""GPU parity tests of the round-4 fallback path: ANY (obs_dim, act_dim, hidden_sizes) for every single-agent script.

The reference's `ActorVCritic(obs_dim, act_dim, hidden_sizes)` takes any dims (safepo/common/model.py:131) and its default sweep
(safepo/single_agent/benchmark.py:5-44) pairs cpo / pcpo / rcpo / trpo_lag / focops / cup / ppo_lag / cppo_pid with 72-88-dim
Car / Doggo / Racecar observations and with HumanoidVelocity's 376 observations / 17 actions.  Shapes outside the LDS-resident
kernels' envelope are routed to the wide-network kernels instead of being refused; these tests pin that path to the CPU oracle
(oracle/restatement.py: the reference's own torch calls) at the north_star tolerance (1e-5 relative for first steps; the
fp64-yardstick gate for multi-step quantities)."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import restatement as R  # noqa: E402  (checker only)
from test_gpu_parity import _synthetic_update_problem, _wide_pair  # noqa: E402


def _theta_floor(lr, nsteps):
    """Max-norm floor of the parameter gates in this file: half an ulp of an O(1) parameter + 0.3 % of the distance Adam can move
    an element in `nsteps` steps.  The L2 gate (typical elements) keeps the plain floor; the max over ~25 000 elements finds the
    few whose gradient is ~1e-3 of the typical size (1 in 10^3 of a Gaussian): rounding noise of relative size 1e-6 on the
    gradient scale is 1e-3 of THEIR gradient, and Adam's normalised step moves them lr x that per step on either side.  With a
    376-wide first layer an MFMA accumulator chains 94 sequential products where the reference's blocked sgemm sums 16-wide
    partials, so the HIP maximum sits ~5x above the fp32 oracle's (5.2e-6 against 1.0e-6 after 6 steps at lr 3e-4) while its L2
    distance stays inside 3x.  (Rounds 1-4 exempted 0.1 % of the parameters up to 2 x lr x nsteps: 600x this.)"""
    return 2e-7 + 3e-3 * lr * nsteps

def _floor_users(hip, f32, f64):
    """WHICH elements need the widened max-norm floor -- those further from float64 than 3 x the fp32 oracle's maximum
    + the plain 2e-7.  Returns (indices, fraction of the vector)."""
    hip, f32, f64 = (np.asarray(t, np.float64).reshape(-1) for t in (hip, f32, f64))
    idx = np.nonzero(np.abs(hip - f64) > 3.0 * np.abs(f32 - f64).max() + 2e-7)[0]
    return idx, idx.size / hip.size


# (obs_dim, act_dim, hidden_sizes): Car-class observations, HumanoidVelocity, a wide action vector on a narrow net
SHAPES = [(72, 2, [64, 64]), (376, 17, [64, 64]), (60, 33, [32, 48])]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_routing_by_dims_instead_of_errors(dev):
    """kernels_supported() looks at (obs_dim, act_dim) too: what the persistent kernels cannot hold is ROUTED to the wide
    engines -- the former SpoError cases of test_limits_and_edge_shapes now compute."""
    from safepo.common.engine import PPOLagEngine, WidePPOLagEngine
    from safepo.common.model import ActorVCritic
    from safepo.single_agent import cpo
    cases = [((60, 8, [64, 64]), True, True), ((72, 2, [64, 64]), True, False), ((128, 16, [64, 64]), True, False),
             ((129, 4, [64, 64]), False, False), ((376, 17, [64, 64]), False, False), ((10, 17, [64, 64]), False, False),
             ((60, 8, [128, 128]), False, False)]
    for (D, A, hs), ppo_ok, cpo_ok in cases:
        pol = ActorVCritic(D, A, hidden_sizes=hs)
        assert pol.kernels_supported("ppo") is ppo_ok and pol.kernels_supported("cpo") is cpo_ok, (D, A, hs)
    cfg = dict(cpo.default_cfg)
    for D, A in ((100, 4), (376, 17)):
        pol = ActorVCritic(D, A).to(dev)
        eng = cpo.make_engine(pol, 2, 8, cfg, dev)
        assert type(eng) is cpo.WideCPOEngine
        hv = eng.fvp(torch.ones(eng.Pa, device=dev))          # the round-3 build raised "cpo: obs_dim outside [1,64]" here
        assert torch.isfinite(hv).all()
    assert type(cpo.make_engine(ActorVCritic(60, 8).to(dev), 2, 8, cfg, dev)) is cpo.CPOEngine
    a, lp, vr, vc = ActorVCritic(129, 4).to(dev).step(torch.zeros(2, 129, device=dev))      # was: SpoError "obs_dim"
    assert a.shape == (2, 4) and torch.isfinite(lp).all()
    a, lp, vr, vc = ActorVCritic(10, 17).to(dev).step(torch.zeros(2, 10, device=dev))       # was: SpoError "act_dim"
    assert a.shape == (2, 17) and torch.isfinite(lp).all()
    pcfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    with pytest.raises(NotImplementedError):
        PPOLagEngine(ActorVCritic(376, 17).to(dev), 1, 64, pcfg, dev)
    with pytest.raises(ValueError):
        WidePPOLagEngine(ActorVCritic(60, 8).to(dev), 1, 64, pcfg, dev)
    from safepo import _abi
    with pytest.raises(_abi.SpoError, match="act_dim"):
        WidePPOLagEngine(ActorVCritic(10, 65).to(dev), 1, 64, pcfg, dev)


@pytest.mark.parametrize("D,A,hidden", [(376, 17, [64, 64]), (200, 20, [64, 64]), (60, 33, [32, 48]), (129, 4, [64, 64])])
def test_wide_dims_policy_step_vs_oracle(dev, D, A, hidden):
    """ActorVCritic.step (model.py:149-170) for dims beyond the persistent kernels' envelope, against the oracle."""
    n = 131
    pol, ref = _wide_pair(D, A, hidden, dev, seed=D + A)
    g = torch.Generator().manual_seed(n)
    obs, eps = torch.randn(n, D, generator=g), torch.randn(n, A, generator=g)
    act, logp, v_r, v_c = pol.step(obs.to(dev), eps=eps.to(dev))
    with torch.no_grad():
        a_ref, lp_ref, vr_ref, vc_ref = ref.step_with_eps(obs, eps)
    tol = dict(rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(act.cpu().numpy(), a_ref.numpy(), **tol)
    np.testing.assert_allclose(logp.cpu().numpy(), lp_ref.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(v_r.cpu().numpy(), vr_ref.numpy(), **tol)
    np.testing.assert_allclose(v_c.cpu().numpy(), vc_ref.numpy(), **tol)


def _gate(hip, f32, f64, floor, what=""):
    """fp64 yardstick: the HIP value may be at most 3x as far from the float64 evaluation as the reference's own float32
    arithmetic is, plus a floor (sums with cancellation: a relative tolerance on the result would test the luck of the draw)."""
    d_hip, d_32 = abs(float(hip) - float(f64)), abs(float(f32) - float(f64))
    assert d_hip <= 3.0 * d_32 + floor, (what, float(hip), float(f32), float(f64), d_hip, d_32, floor)


def _cpo_problem(D, A, hidden, M, dev, seed, chunk):
    from safepo.single_agent import cpo
    pol, ref = _wide_pair(D, A, hidden, dev, seed=seed)
    cfg = dict(cpo.default_cfg)
    cfg["hidden_sizes"] = hidden
    eng = cpo.make_engine(pol, 1, M, cfg, dev)
    assert type(eng) is cpo.WideCPOEngine
    eng.CHUNK = chunk
    obs, act, logp, tgt_r, tgt_c, adv = _synthetic_update_problem(M, D, A, seed=seed + 1)
    adv_c = adv.flip(0) * 0.5 + 0.1
    b = eng.buffer
    b.data["obs"].copy_(obs.view(1, M, D)); b.data["act"].copy_(act.view(1, M, A)); b.data["log_prob"].copy_(logp.view(1, M))
    b.data["adv_r"].copy_(adv.view(1, M)); b.data["adv_c"].copy_(adv_c.view(1, M))
    b.data["target_value_r"].copy_(tgt_r.view(1, M)); b.data["target_value_c"].copy_(tgt_c.view(1, M))
    data = {"obs": obs, "act": act, "log_prob": logp, "adv_r": adv, "adv_c": adv_c, "target_value_r": tgt_r, "target_value_c": tgt_c}
    return pol, ref, eng, data


@pytest.mark.parametrize("D,A,hidden", SHAPES + [(60, 8, [128, 128])])
def test_wide_cpo_primitives_vs_oracle(dev, D, A, hidden):
    """The three full-batch primitives of the second-order scripts on the wide kernels, in row chunks (3 chunks + a ragged
    tail): both surrogate gradients (cpo.py:356-381) against autograd, the Fisher-vector product against the reference's
    DOUBLE BACKWARD (cpo.py:132-157) in float32 and float64, the line-search sums (cpo.py:473-491) at the old and at moved
    parameters."""
    M = 3000 + 37
    pol, ref, eng, data = _cpo_problem(D, A, hidden, M, dev, seed=7 + D, chunk=1024)
    b = eng.buffer
    ref64 = copy.deepcopy(ref).double()
    data64 = {k: v.double() for k, v in data.items()}
    # log-probabilities of act_dim terms of size ~1.4 carry ~1e-5 of absolute rounding at 33 dims -- in the reference's fp32 as
    # much as here -- so the surrogate MEANS (sums of +-O(1) terms) are gated against float64, not by a relative tolerance
    floor = 1e-6 * float(data["adv_r"].abs().mean())
    for which, key, sign in (("r", "adv_r", -1.0), ("c", "adv_c", 1.0)):
        ref.actor.zero_grad()
        loss = R.cpo_surrogate(ref, data, which)
        loss.backward()
        g_ref = R.actor_flat_grads(ref.actor).numpy()
        g, mean = eng.surrogate_grad(b.data[key], sign)
        np.testing.assert_allclose(g.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-5 * np.abs(g_ref).max())
        _gate(sign * mean, loss.detach(), R.cpo_surrogate(ref64, data64, which).detach(), floor, f"surrogate {which}")
    v = torch.randn(eng.Pa, generator=torch.Generator().manual_seed(3))
    hv32 = R.cpo_fvp(v, ref, data["obs"]).double().numpy()
    hv64 = R.cpo_fvp(v.double(), ref64, data64["obs"]).numpy()
    hv = eng.fvp(v.to(dev)).double().cpu().numpy()
    scale = np.abs(hv64).max()
    d_hip, d_32 = np.abs(hv - hv64).max(), np.abs(hv32 - hv64).max()
    assert d_hip <= 3.0 * d_32 + 1e-6 * scale, (d_hip, d_32, scale)      # fp64 yardstick: as close as the reference's own fp32
    np.testing.assert_allclose(hv, hv32, rtol=1e-4, atol=1e-5 * scale)
    # line search: unchanged parameters -> KL == 0 and the two surrogates; moved parameters -> the oracle's values
    eng.snapshot_old_distribution()
    l_r, l_c, kl = eng.linesearch_eval()
    assert kl == pytest.approx(0.0, abs=1e-9)
    with torch.no_grad():
        _gate(l_r, R.cpo_surrogate(ref, data, "r"), R.cpo_surrogate(ref64, data64, "r"), floor, "line search r at theta_old")
        _gate(l_c, R.cpo_surrogate(ref, data, "c"), R.cpo_surrogate(ref64, data64, "c"), floor, "line search c at theta_old")

        def moved(rf, dt):
            old = rf.actor(dt["obs"])
            old_mean, old_std = old.mean.clone(), old.stddev.clone()
            R.actor_set_flat_params(rf.actor, R.actor_flat_params(rf.actor) + delta.to(old_mean.dtype))
            kl_ = torch.distributions.kl_divergence(torch.distributions.Normal(old_mean, old_std), rf.actor(dt["obs"])).mean()
            return float(kl_), float(R.cpo_surrogate(rf, dt, "r")), float(R.cpo_surrogate(rf, dt, "c"))
        delta = 0.02 * torch.randn(eng.Pa, generator=torch.Generator().manual_seed(5))
        eng.theta_actor.add_(delta.to(dev))
        kl32, r32, c32 = moved(ref, data)
        kl64, r64, c64 = moved(ref64, data64)
    l_r, l_c, kl = eng.linesearch_eval()
    assert kl == pytest.approx(kl32, rel=1e-4)
    _gate(kl, kl32, kl64, 1e-6 * kl64, "KL at moved parameters")
    _gate(l_r, r32, r64, floor, "line search r at moved parameters")
    _gate(l_c, c32, c64, floor, "line search c at moved parameters")


@pytest.mark.parametrize("D,A,hidden,ep_costs", [(72, 2, [64, 64], -1.0), (376, 17, [64, 64], 0.3), (376, 17, [64, 64], -1.0)])
def test_wide_cpo_actor_step_drift_envelope(dev, D, A, hidden, ep_costs):
    """CPO's whole trust-region step (cpo.py:350-532) through WideCPOEngine under the fp64-yardstick gate of
    test_cpo_actor_step_drift_envelope: discrete decisions equal, curvature / step length / parameters at most 3x as far from
    the float64 step as the float32 oracle is."""
    M = 4096
    pol, ref32, eng, data32 = _cpo_problem(D, A, hidden, M, dev, seed=14, chunk=1500)
    ref64 = copy.deepcopy(ref32).double()
    data64 = {k: v.double() for k, v in data32.items()}
    tk = eng.cfg["target_kl"]
    o32 = R.cpo_policy_update(ref32, data32, ep_costs, target_kl=tk)
    o64 = R.cpo_policy_update(ref64, data64, ep_costs, target_kl=tk)
    th32 = R.actor_flat_params(ref32.actor).double().numpy()
    th64 = R.actor_flat_params(ref64.actor).double().numpy()
    out = eng.policy_update(ep_costs)
    th_hip = eng.theta_actor.double().cpu().numpy()
    assert out["case"] == o32["case"] == o64["case"]
    assert out["acceptance_step"] == o32["accept"] == o64["accept"]
    for name, hip, v32, v64 in (("xHx", out["xHx"], float(o32["xHx"]), float(o64["xHx"])),
                                ("alpha", out["alpha"], float(o32["alpha"]), float(o64["alpha"]))):
        assert abs(hip - v64) <= 3.0 * abs(v32 - v64) + 2e-6 * abs(v64), (name, hip, v32, v64)
    d_hip, d_32 = np.abs(th_hip - th64), np.abs(th32 - th64)
    scale = np.abs(th64).max()
    assert np.linalg.norm(d_hip) <= 3.0 * np.linalg.norm(d_32) + 1e-7 * scale * np.sqrt(th64.size), (np.linalg.norm(d_hip), np.linalg.norm(d_32))
    assert d_hip.max() <= 3.0 * d_32.max() + 1e-6 * scale, (d_hip.max(), d_32.max())
    # the stale actor gradient the critic fit's joint clip will see (cpo.py:557) is the cost gradient b
    np.testing.assert_allclose(eng.flat_grad[eng.ls_off:].cpu().numpy(), out["b"].cpu().numpy())


@pytest.mark.parametrize("D,A,hidden,persistent", [(72, 2, [64, 64], True), (376, 17, [64, 64], False), (60, 33, [32, 48], False),
                                                   (376, 17, [64, 64], "wide")])
def test_wide_cpo_critic_fit_vs_oracle(dev, D, A, hidden, persistent, monkeypatch):
    """Critic fit (cpo.py:534-571) of WideCPOEngine: on the persistent two-critic kernel when the critics fit it (obs 72: the
    KIN = 128 instantiation with act_dim irrelevant); on the persistent FEATURE-SPLIT kernel for hidden [64, 64] critics up to
    512 observations (376: round 5, csrc/update_ks.hip with two networks, a 128-row minibatch as two 64-column chunks);
    minibatch by minibatch on the wide kernels otherwise ("wide": SPO_WIDE_KS=0 keeps that path at 376) -- with the stale actor
    gradient taking part in, and being rescaled by, the joint clip."""
    if persistent == "wide":
        monkeypatch.setenv("SPO_WIDE_KS", "0")
        persistent = False
    _critic_fit_check(dev, D, A, hidden, persistent, monkeypatch, M=1024, iters=2, batch=128)


@pytest.mark.parametrize("D,A,M,batch", [(376, 17, 1024 + 37, 128), (200, 4, 900, 100), (130, 2, 640 + 5, 64), (512, 32, 512, 128),
                                         (129, 1, 300, 128)])
def test_feature_split_critic_fit_shapes_vs_oracle(dev, D, A, M, batch, monkeypatch):
    """The feature-split critic fit at other slice counts (2 ... 8), minibatches that are not two full chunks (100 = 64 + 36 rows,
    64 = one chunk) and a ragged last minibatch (also one that leaves the second chunk EMPTY: 37 of 128 rows)."""
    _critic_fit_check(dev, D, A, [64, 64], False, monkeypatch, M=M, iters=2, batch=batch, expect_ks=True)


def _critic_fit_check(dev, D, A, hidden, persistent, monkeypatch, M, iters, batch, expect_ks=None):
    monkeypatch.setenv("SPO_CPO_SPLIT", "0")
    pol, ref, eng, data = _cpo_problem(D, A, hidden, M, dev, seed=5, chunk=4096)
    eng.cfg.update(learning_iters=iters, batch_size=batch)
    assert eng._critics_on_persistent_kernel is persistent
    if expect_ks is not None:
        assert eng._feature_split_critic_fit_ok(eng._cfg_struct()) is expect_ks
    n_act = sum(p.numel() for p in ref.actor.parameters())
    stale = torch.full((n_act,), 50.0 / np.sqrt(n_act))             # norm 50 > max_grad_norm 40: the clip is active
    eng._set_stale_actor_grad(stale.to(dev))
    g = torch.Generator().manual_seed(5)
    perms = [torch.randperm(M, generator=g).to(torch.int32) for _ in range(iters)]
    ref64 = copy.deepcopy(ref).double()
    fit = eng.critic_fit(perm_fn=lambda it: perms[it].to(dev))

    def oracle(rf, dtype):
        for p in rf.actor.parameters():
            p.grad = torch.full_like(p, 50.0 / np.sqrt(n_act))
        fitter = R.CriticFitter(rf)
        o_, tr_, tc_ = (data[k].to(dtype) for k in ("obs", "target_value_r", "target_value_c"))
        out = []
        for it in range(iters):
            pm = perms[it].long()
            for k in range((M + batch - 1) // batch):
                idx = pm[k * batch:(k + 1) * batch]
                out.append(fitter.minibatch_step(o_[idx], tr_[idx], tc_[idx]))
        return np.asarray(out, np.float64), torch.cat([p.detach().reshape(-1) for p in rf.parameters()]).double().numpy()
    want, want_th = oracle(ref, torch.float32)
    l64, th64 = oracle(ref64, torch.float64)
    got = torch.cat(fit["losses"], 0).cpu().numpy()
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-6)          # first step: 1e-5
    # the fit under the drift envelope (round 5: instead of rtol 2e-4 on the losses / 2e-3 on the critics)
    import envelope as E
    n_crit = want_th.size - n_act
    E.assert_loss_envelope(got, want, l64, "wide critic fit: losses", window=len(want))
    E.assert_theta_envelope(pol.theta.cpu().numpy()[:n_crit], want_th[:n_crit], th64[:n_crit], "wide critic fit: critics",
                            floor_abs_max=_theta_floor(1e-3, iters * ((M + batch - 1) // batch)))
    want_th = want_th.astype(np.float32)
    # the actor's parameters are untouched by the critic fit; its stale gradient shrank exactly like the oracle's .grad
    np.testing.assert_array_equal(pol.theta.cpu().numpy()[n_crit:], want_th[n_crit:])
    want_stale = torch.cat([p.grad.reshape(-1) for p in ref.actor.parameters()])
    stale64 = torch.cat([p.grad.reshape(-1) for p in ref64.actor.parameters()])
    if not persistent:
        E.gate_array(eng.flat_grad[eng.ls_off:].cpu().numpy(), want_stale.numpy(), stale64.numpy(), "stale actor gradient after the fit",
                     rel_floor=1e-6)
    E.gate_scalars([("stale_sq", float(eng.stale_sq.item()), float(want_stale.double().dot(want_stale.double())), float(stale64.dot(stale64)))],
                   "norm^2 of the stale actor gradient", rel_floor=2e-6)


@pytest.mark.parametrize("D,A,hidden,actor_only", [(376, 17, [64, 64], False), (376, 17, [64, 64], True), (60, 33, [32, 48], False),
                                                   (60, 8, [128, 128], True), (130, 8, [64, 64], False), (512, 32, [64, 64], True),
                                                   (200, 20, [64, 64], False), (129, 1, [64, 64], True)])
def test_wide_kl_penalty_minibatch_steps_vs_oracle(dev, D, A, hidden, actor_only):
    """FOCOPS's minibatch step (focops.py:312-347) and CUP's actor-only second stage (cup.py:370-386) for shapes the
    three-workgroup persistent kernel does not hold: the test_kl_penalty_minibatch_steps_vs_oracle protocol (indicator active on
    part of the batch, the actor's optimiser clock ahead of the critics', partial last batch).  hidden [64, 64] with obs_dim <= 512
    / act_dim <= 32 runs on the persistent feature-split kernel (spo_update_iter_ex_ks, round 5: 2 ... 8 feature slices here, the
    actor alone in CUP's stage), other widths on the launch-per-layer wide kernels."""
    from safepo import _abi
    from safepo.common.engine import WidePPOLagEngine
    M = 192 + 22
    pol, ref = _wide_pair(D, A, hidden, dev, seed=M + D)
    obs, act, logp, tgt_r, tgt_c, adv = _synthetic_update_problem(M, D, A, seed=M)
    with torch.no_grad():
        dist = ref.actor(obs)
        old_mean = dist.mean + 0.05 * torch.randn(M, A)
        old_std = dist.stddev[0] * torch.exp(0.05 * torch.randn(A))
        kl0 = torch.distributions.kl_divergence(dist, torch.distributions.Normal(old_mean, old_std.expand(M, A))).sum(-1)
    kl_bound = float(kl0.quantile(0.55)) if not actor_only else float("inf")
    pg_coef = 1 / 1.5 if not actor_only else -0.37
    cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
    assert eng._feature_split_kernel_ok(eng._cfg_struct()) is (hidden == [64, 64])
    b = eng.buffer
    b.data["obs"].copy_(obs.view(1, M, D)); b.data["act"].copy_(act.view(1, M, A)); b.data["log_prob"].copy_(logp.view(1, M))
    b.data["target_value_r"].copy_(tgt_r.view(1, M)); b.data["target_value_c"].copy_(tgt_c.view(1, M))
    eng.mean_old.copy_(old_mean); eng.std_old.copy_(old_std)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(3))
    eng.adam_step_actor_extra = 5
    theta0 = pol.theta.clone()
    ref64 = copy.deepcopy(ref).double()

    def oracle(rp, dtype):
        upd = R.KLPenaltyUpdater(rp)
        for _ in range(5):                          # the actor's Adam clock runs 5 steps ahead (zero gradients: moments stay 0)
            upd.opt_a.zero_grad()
            for prm in rp.actor.parameters():
                prm.grad = torch.zeros_like(prm)
            upd.opt_a.step()
        o_, a_, lp_, tr_, tc_, ad_, om_ = (t.to(dtype) for t in (obs, act, logp, tgt_r, tgt_c, adv, old_mean))
        os_full = old_std.expand(M, A).to(dtype)
        out, n_masked, margin = [], 0, float("inf")
        for s0 in range(0, M, 64):
            idx = perm[s0:s0 + 64]
            with torch.no_grad():
                kl_i = torch.distributions.kl_divergence(rp.actor(o_[idx]), torch.distributions.Normal(om_[idx], os_full[idx])).sum(-1)
                n_masked += int((kl_i > kl_bound).sum())
                margin = min(margin, float((kl_i - kl_bound).abs().min()))
            if actor_only:
                l = upd.cup_second_stage_step(o_[idx], a_[idx], lp_[idx], ad_[idx], om_[idx], os_full[idx],
                                              0.37 / ((1 - 0.99 * 0.95) / (1 - 0.99)), 0.99)
                out.append([np.nan, np.nan, l])
            else:
                out.append(list(upd.focops_step(o_[idx], a_[idx], lp_[idx], tr_[idx], tc_[idx], ad_[idx], om_[idx], os_full[idx], kl_bound)))
        return np.asarray(out, np.float64), R.flat_params(rp).double().numpy(), n_masked, margin
    ref_losses, th32, n_masked, margin = oracle(ref, torch.float32)
    l64, th64, n_masked64, _ = oracle(ref64, torch.float64)
    if not actor_only:
        assert 0 < n_masked < M, n_masked
        assert n_masked == n_masked64 and margin > 1e-6, (n_masked, n_masked64, margin)        # no sample sits ON the bound
    losses = eng.learning_iter_ex(perm.to(torch.int32).to(dev), adv.to(dev).contiguous(), _abi.ACTOR_LOSS_KL_PENALTY, kl_bound,
                                  pg_coef, actor_only)
    got = losses.cpu().numpy()
    np.testing.assert_allclose(got[0], ref_losses[0], rtol=1e-5, atol=2e-6, equal_nan=True)      # first step: 1e-5
    n_steps = (M + 63) // 64
    import envelope as E
    cols = [2] if actor_only else [0, 1, 2]
    E.assert_loss_envelope(got[:, cols], ref_losses[:, cols], l64[:, cols], "wide KL-penalty pass: losses", window=n_steps, floor_rel=3e-6)
    E.assert_theta_envelope(pol.theta.cpu().numpy(), th32, th64, "wide KL-penalty pass: theta", floor_abs_max=_theta_floor(3e-4, n_steps))
    if actor_only:
        off = pol.log_std_offset
        assert torch.equal(pol.theta[:off], theta0[:off])
        assert eng.adam_step == 0 and eng.adam_step_actor_extra == 5 + n_steps


@pytest.mark.parametrize("D,A,batch,steps", [(376, 17, 64, 6), (200, 20, 100, 3)])
def test_wide_dims_ppo_minibatch_steps_vs_oracle(dev, D, A, batch, steps):
    """ppo_lag.py:306-329 at dims beyond the persistent kernels: per-step losses at 1e-5, parameters after the steps."""
    from safepo.common.engine import WidePPOLagEngine
    from test_gpu_parity import _fill_update_problem
    hidden = [64, 64]
    pol, ref = _wide_pair(D, A, hidden, dev, seed=3)
    ref0 = {k: v.clone() for k, v in ref.state_dict().items()}
    M = batch * steps
    cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 0.02, "batch_size": batch, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
    problem = _synthetic_update_problem(M, D, A, seed=17)
    _fill_update_problem(eng, problem)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(2))
    losses = eng.learning_iter(perm.to(torch.int32).to(dev)).cpu().numpy()
    upd = R.PPOLagUpdater(ref, epochs=1, max_grad_norm=40.0)
    obs, act, logp, tgt_r, tgt_c, adv = problem
    want = [upd.minibatch_step(*(t[perm[k * batch:(k + 1) * batch]] for t in (obs, act, logp, tgt_r, tgt_c, adv))) for k in range(steps)]
    np.testing.assert_allclose(losses, np.asarray(want), rtol=1e-5, atol=2e-6)
    import envelope as E
    _, t64 = E.oracle_trajectory(ref0, problem, perm, batch, steps, torch.float64, [steps], hidden_sizes=hidden)
    E.assert_theta_envelope(pol.theta.cpu().numpy(), R.flat_params(ref).numpy(), t64[steps], f"wide dims {D}x{A}: theta after {steps} steps",
                            floor_abs_max=_theta_floor(3e-4, steps))


@pytest.mark.parametrize("D,A", [(376, 17), (130, 8), (512, 32), (129, 1), (200, 20)])
def test_feature_split_gradient_launch_vs_fp64_autograd(dev, D, A):
    """spo_ppo_lag_grad_ks (round 6: the data-parallel step at the feature-split kernel's dims): the raw gradient and the data losses
    of ONE minibatch (ppo_lag.py:306-324 without the L2 terms, which spo_wide_clip_adam adds) against float64 autograd on the
    oracle policy -- full, ragged and one-row minibatches, launched back to back on one stream (only the block's first launch sets
    the exchange slots to the sentinel; the later ones rely on the consumers' resets), with a persistent launch of the same
    kernel family in between; theta is not written.  Then the launch-per-layer gradient of the same rows for scale."""
    from safepo import _abi
    from safepo.common.engine import WidePPOLagEngine
    from test_gpu_parity import _fill_update_problem
    hidden, M = [64, 64], 64 * 5 + 1
    pol, ref = _wide_pair(D, A, hidden, dev, seed=11)
    cfg_d = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 0.02, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = WidePPOLagEngine(pol, 1, M, cfg_d, dev)
    problem = _synthetic_update_problem(M, D, A, seed=23)
    _fill_update_problem(eng, problem)
    lib, d, b = eng.lib, eng.buffer.data, eng.buffer
    cfg = eng._cfg_struct()
    ref64 = copy.deepcopy(ref).double()
    theta0 = pol.theta.clone()
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(5))
    cases = [perm[:64], perm[64:64 + 37], perm[128:129], perm[129:129 + 64], perm[200:232]]
    outs = []
    for k, idx in enumerate(cases):
        idx_dev = idx.to(torch.int32).to(dev)
        g = torch.full_like(eng.flat_grad, float("nan"))
        l3 = torch.full((3,), float("nan"), device=dev)
        _abi.check(lib.spo_ppo_lag_grad_ks(_abi.ptr(pol.theta), _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]),
                                           _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(b.adv_mix),
                                           _abi.ptr(idx_dev), idx_dev.numel(), cfg, _abi.ptr(g), _abi.ptr(l3), _abi.ptr(eng.sync_ws),
                                           _abi.stream_ptr()), "spo_ppo_lag_grad_ks")
        outs.append((g, l3))
        if k == 2:      # a persistent launch of the update kernel on the same scratch block between two gradient launches
            snap = [t.clone() for t in (pol.theta, eng.adam_m, eng.adam_v)]
            eng.learning_iter(perm.to(torch.int32).to(dev))
            for t, s_ in zip((pol.theta, eng.adam_m, eng.adam_v), snap):
                t.copy_(s_)
            eng.adam_step = 0
    eng.check_sync_error()
    assert torch.equal(pol.theta, theta0)
    worst = 0.0
    for idx, (g, l3) in zip(cases, outs):
        rows = [t[idx].double() for t in problem]
        ref64.zero_grad()
        total, loss_pi, loss_r, loss_c = R.ppo_lag_losses(ref64, *rows, use_critic_norm=False)
        total.backward()
        want = R.flat_grads(ref64).numpy()
        got = g.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), f"{int(np.isnan(got).sum())} gradient elements never written (rows {idx.numel()})"
        np.testing.assert_allclose(l3.cpu().numpy(), [loss_r.item(), loss_c.item(), loss_pi.item()], rtol=1e-5, atol=2e-6)
        err = np.abs(got - want).max() / np.abs(want).max()
        worst = max(worst, err)
        assert err < 2e-6, (idx.numel(), err)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=3e-6 * np.abs(want).max())
    # the launch-per-layer gradient of the first minibatch (the path this launch replaces under data parallelism)
    obs, act, logp_old, tgt_r, tgt_c, adv = eng._gather(cases[0].to(dev))
    w, n = eng.wide, 64
    (v_r, ws_r), (v_c, ws_c), (mu, ws_a) = w.forward_multi("rca", obs, slot=1)
    d_vr = torch.empty(n, device=dev); d_vc = torch.empty(n, device=dev); d_mu = torch.empty((n, A), device=dev)
    g2, l3b = torch.zeros_like(eng.flat_grad), torch.zeros(3, device=dev)
    _abi.check(lib.spo_wide_ppo_loss(_abi.ptr(v_r), _abi.ptr(v_c), _abi.ptr(mu), _abi.ptr(pol.theta[w.off_ls:]), _abi.ptr(act),
                                     _abi.ptr(logp_old), _abi.ptr(adv), _abi.ptr(tgt_r), _abi.ptr(tgt_c), n, A, float(cfg.clip),
                                     _abi.ptr(d_vr), _abi.ptr(d_vc), _abi.ptr(d_mu), _abi.ptr(g2[w.off_ls:]), _abi.ptr(l3b),
                                     _abi.ptr(eng.loss_partials), eng.loss_partials.numel(), _abi.stream_ptr()), "spo_wide_ppo_loss")
    w.backward_multi("rca", obs, [ws_r, ws_c, ws_a], [d_vr, d_vc, d_mu], g2)
    rel = float((outs[0][0] - g2).abs().max() / g2.abs().max())
    print(f"grad_ks {D}x{A}: worst max-norm error vs fp64 autograd {worst:.2e}; vs the launch-per-layer gradient {rel:.2e}")
    assert rel < 4e-6


@pytest.mark.parametrize("D,A,hidden", [(60, 8, [128, 128]), (33, 3, [256, 96]), (17, 2, [32]), (61, 5, [100, 50, 30]), (376, 17, [128, 128]),
                                        (60, 8, [64, 64, 64, 64]), (5, 64, [48]), (60, 8, [256, 256])])
def test_row_group_gradient_launch_vs_fp64_autograd(dev, D, A, hidden):
    """csrc/mlp_rows.hip (round 6): gather + forward + loss + backward of the three networks of a minibatch for any hidden_sizes in
    one launch split over 16-row groups, then the fixed-order sum of the groups' partial gradients -- against float64 autograd of
    ppo_lag.py:306-324 (without the L2 terms: spo_wide_clip_adam adds them) on the oracle policy.  Full, ragged, one-row, multi-group
    and maximum (256-row) minibatches; widths that are not multiples of 16 or 4 (scalar weight loads), 1 to 4 hidden layers, a
    64-wide action vector; the index window read through a device cursor as a replayed step does; the critics-only form of the
    critic fit (cpo.py:541-556); theta not written."""
    from safepo import _abi
    from safepo.common.engine import WidePPOLagEngine
    from safepo.common.wide import PermWindow
    from test_gpu_parity import _fill_update_problem
    M = 700
    pol, ref = _wide_pair(D, A, hidden, dev, seed=13)
    cfg_d = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 0.02, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = WidePPOLagEngine(pol, 1, M, cfg_d, dev)
    w = eng.wide
    assert w.rows_grad_ok(64) and w.rows_grad_ok(256) and not w.rows_grad_ok(257)
    problem = _synthetic_update_problem(M, D, A, seed=29)
    _fill_update_problem(eng, problem)
    d, b = eng.buffer.data, eng.buffer
    ref64 = copy.deepcopy(ref).double()
    theta0 = pol.theta.clone()
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(6))
    arrays = (d["obs"].view(M, D), d["act"].view(M, A), d["log_prob"].view(M), d["target_value_r"].view(M), d["target_value_c"].view(M),
              b.adv_mix.view(M))
    worst = 0.0
    for lo, n, windowed in ((0, 64, False), (64, 37, False), (101, 1, False), (102, 100, True), (202, 256, False), (458, 16, True), (474, 17, False)):
        idx = perm[lo:lo + n]
        g = torch.full_like(eng.flat_grad, float("nan"))
        l3 = torch.full((3,), float("nan"), device=dev)
        if windowed:
            win = PermWindow(M, n, dev)
            win.load(perm.to(dev))
            win.cursor.fill_(lo)
            w.grad_rows(win, *arrays, 0.2, g, l3)
        else:
            w.grad_rows(idx.to(dev), *arrays, 0.2, g, l3)
        rows = [t[idx].double() for t in problem]
        ref64.zero_grad()
        total, loss_pi, loss_r, loss_c = R.ppo_lag_losses(ref64, *rows, use_critic_norm=False)
        total.backward()
        want = R.flat_grads(ref64).numpy()
        got = g.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), f"{int(np.isnan(got).sum())} gradient elements never written (rows {n})"
        np.testing.assert_allclose(l3.cpu().numpy(), [loss_r.item(), loss_c.item(), loss_pi.item()], rtol=1e-5, atol=2e-6)
        err = np.abs(got - want).max() / np.abs(want).max()
        worst = max(worst, err)
        # (the log-probability is a sum of act_dim fp32 terms of size ~1 under an exp: the ratio's rounding grows with act_dim)
        gate = 3e-6 * max(1.0, A / 8.0)
        assert err < gate, (n, err)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=gate * np.abs(want).max())
        if n == 64:                                      # critics only: the same numbers for the two critics, nothing beyond them
            g2 = torch.full_like(eng.flat_grad, float("nan"))
            l2 = torch.full((2,), float("nan"), device=dev)
            w.grad_rows(idx.to(dev), *arrays, 0.2, g2, l2, critics_only=True)
            assert torch.equal(g2[:2 * w.Pc], g[:2 * w.Pc]) and torch.isnan(g2[2 * w.Pc:]).all() and torch.equal(l2, l3[:2])
    assert torch.equal(pol.theta, theta0)
    print(f"row-group gradient {D}x{A} {hidden}: worst max-norm error vs fp64 autograd {worst:.2e}")


def test_feature_split_kernel_full_size_drift_envelope_at_humanoid_dims(dev):
    """The persistent feature-split kernel (csrc/update_ks.hip) at BASELINE config 2's size with HumanoidVelocity's dims:
    4096 envs x 128 steps = 524 288 rows of 376 observations / 17 actions, one learning iteration = 8 192 minibatch steps of 64
    rows in ONE launch of 18 workgroups (ppo_lag.py:297-336).  Same gate as test_full_size_update_parity_drift_envelope has at
    60 / 8: the first 8 steps at 1e-5 against the fp32 oracle; every 64-step window of the per-minibatch losses and the
    parameters after 8 / 64 / 512 / 2 048 / 8 192 steps no further from the oracle's float64 trajectory than 3x the fp32 oracle
    (= the reference's arithmetic) is itself; every shorter launch a bit-exact prefix of the longest (the exchange between the
    workgroups does not depend on timing)."""
    import envelope as E
    from safepo.common.engine import WidePPOLagEngine
    from test_gpu_parity import _fill_update_problem, _hip_prefix_runs
    D, A, hidden, batch = 376, 17, [64, 64], 64
    M = 4096 * 128
    pol, ref = _wide_pair(D, A, hidden, dev, seed=5)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 1e9, "batch_size": batch, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
    assert eng._feature_split_kernel_ok(eng._cfg_struct())
    problem = _synthetic_update_problem(M, D, A, seed=2025)
    _fill_update_problem(eng, problem)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(8))
    ks = (8, 64, 512, 2048, 8192)
    runs = _hip_prefix_runs(eng, pol, pol.theta.clone(), perm.to(torch.int32).to(dev), batch, ks)
    assert eng.adam_step == 8192
    kmax = max(ks)
    l32, t32 = E.oracle_trajectory(sd0, problem, perm, batch, kmax, torch.float32, ks, hidden_sizes=hidden)
    l64, t64 = E.oracle_trajectory(sd0, problem, perm, batch, kmax, torch.float64, ks, hidden_sizes=hidden)
    lh = runs[kmax][1]
    np.testing.assert_allclose(lh[:8], l32[:8], rtol=1e-5, atol=1e-6, err_msg="first 8 steps")
    rep = {"loss": E.assert_loss_envelope(lh, l32, l64, "feature-split full size")}
    users = {}
    for k in ks:
        assert np.array_equal(runs[k][1], lh[:k]), f"the {k}-step launch is not a prefix of the {kmax}-step launch"
        rep[k] = E.assert_theta_envelope(runs[k][0], t32[k], t64[k], f"feature-split full size: theta after {k} steps",
                                         floor_abs_max=_theta_floor(3e-4, k))
        users[k], frac = _floor_users(runs[k][0], t32[k], t64[k])
        #  the widened floor must be serving a handful of Adam-amplified elements, not a systematic offset: at most
        # 0.1 % of the vector may need it at any checkpoint
        assert frac <= 1e-3, f"after {k} steps {users[k].size} elements ({frac:.2%}) need the widened max-norm floor"
    # ... and they must be what the floor's rationale says they are -- elements whose gradient is far below the typical size, where
    # Adam's normalised step turns rounding noise of the gradient into full-size steps of either sign (the shorter launches are
    # prefixes of the longest, so an element that left the narrow gate early stays out: the same index at later checkpoints is
    # expected; a replica of b3 / log_std or the Adam state of a slice edge going wrong would NOT have a small second moment)
    last = users[kmax]
    if last.size:
        v_all = eng.adam_v.double().cpu().numpy()
        ratio_v = float(v_all[last].max() / np.median(v_all[v_all > 0]))
        print(f"second moment of the elements beyond the narrow gate after {kmax} steps / median second moment: {ratio_v:.3e} "
              f"(indices {last.tolist()[:8]})")
        assert ratio_v < 0.25, f"elements {last.tolist()[:8]} need the widened floor although their gradients are not small ({ratio_v:.3e})"
    print("feature-split kernel, 376 / 17, drift envelope (ratio <= 1 passes):", rep)
    print("elements beyond 3 x the fp32 oracle's maximum + 2e-7 per checkpoint:", {k: v.tolist()[:8] for k, v in users.items()})


def test_feature_split_critic_fit_full_size_drift_envelope_at_humanoid_dims(dev, monkeypatch):
    """critic_fit_ks_kernel at BASELINE size with HumanoidVelocity's observations -- 524 288 rows x 376, ONE
    launch of 4 096 minibatch steps of 128 rows (two 64-column chunks each, cpo.py:541-571; the stale actor gradient of norm 50 in
    the joint clip) on 12 workgroups.  Same gate as the three-workgroup kernel's test_cpo_full_size_critic_fit_drift_envelope:
    first 8 steps at 1e-5, per-step losses and the critics after 8 / 64 / 512 / 4 096 steps under the float64 yardstick, every
    shorter launch a bit-exact prefix of the longest."""
    import envelope as E
    from safepo.single_agent import cpo
    from test_gpu_parity import _oracle_critic_trajectory
    monkeypatch.setenv("SPO_CPO_SPLIT", "0")
    torch.set_num_threads(8)
    D, A, hidden, batch = 376, 17, [64, 64], 128
    N, T = 4096, 128
    M = N * T
    ks = (8, 64, 512, 4096)
    pol, ref = _wide_pair(D, A, hidden, dev, seed=6)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    cfg = dict(cpo.default_cfg)
    cfg.update(learning_iters=1, batch_size=batch)
    eng = cpo.make_engine(pol, N, T, cfg, dev)
    assert type(eng) is cpo.WideCPOEngine and eng._feature_split_critic_fit_ok(eng._cfg_struct())
    obs, _a, _l, tgt_r, tgt_c, _adv = _synthetic_update_problem(M, D, A, seed=2027)
    bd = eng.buffer.data
    bd["obs"].copy_(obs.view(N, T, D)); bd["target_value_r"].copy_(tgt_r.view(N, T)); bd["target_value_c"].copy_(tgt_c.view(N, T))
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(9)).to(torch.int32)
    perm_dev = perm.to(dev)
    theta0 = pol.theta.clone()
    n_act = sum(p.numel() for p in ref.actor.parameters())
    n_crit = theta0.numel() - n_act
    stale = torch.full((n_act,), 50.0 / np.sqrt(n_act)).to(dev)
    hip = {}
    for k in ks:
        pol.theta.copy_(theta0); eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_step = 0
        eng._set_stale_actor_grad(stale)
        eng.M = k * batch
        fit = eng.critic_fit(perm_fn=lambda it: perm_dev[:k * batch].contiguous())
        eng.check_sync_error()
        hip[k] = (pol.theta[:n_crit].double().cpu().numpy(), torch.cat(fit["losses"], 0).double().cpu().numpy())
    eng.M = M
    kmax = max(ks)
    l32, t32 = _oracle_critic_trajectory(sd0, obs, tgt_r, tgt_c, perm, batch, kmax, torch.float32, ks, 50.0)
    l64, t64 = _oracle_critic_trajectory(sd0, obs, tgt_r, tgt_c, perm, batch, kmax, torch.float64, ks, 50.0)
    lh = hip[kmax][1]
    np.testing.assert_allclose(lh[:8], l32[:8], rtol=1e-5, atol=1e-6, err_msg="first 8 critic-fit steps")
    rep = {"loss": E.assert_loss_envelope(lh, l32, l64, "feature-split critic fit, full size")}
    for k in ks:
        assert np.array_equal(hip[k][1], lh[:k]), f"the {k}-step launch is not a prefix of the {kmax}-step launch"
        rep[k] = E.assert_theta_envelope(hip[k][0], t32[k], t64[k], f"feature-split critic fit: critics after {k} steps",
                                         floor_abs_max=_theta_floor(1e-3, k))
    print("critic_fit_ks_kernel, 376 / 17, 524 288 rows, drift envelope (ratio <= 1 passes):",
          {k: (round(v[0], 3), f"max |hip-f64| {v[1]['max_hip']:.2e} vs f32 {v[1]['max_f32']:.2e}") if k != "loss" else round(v, 3)
           for k, v in rep.items()})


def _kl_penalty_trajectory(sd, problem, old_mean, old_std, perm, batch, nsteps, dtype, checkpoints, actor_only, kl_bound, pg_coef,
                           hidden, actor_clock_ahead):
    """`nsteps` consecutive FOCOPS steps (focops.py:312-347) or CUP second-stage steps (cup.py:370-386) of the oracle in `dtype`;
    returns (losses [nsteps, 3] with NaN where a network is not fitted, {k: flat theta after k steps}) as float64."""
    obs, act, logp, tgt_r, tgt_c, adv = [t.to(dtype) for t in problem]
    M, D, A = obs.shape[0], obs.shape[1], act.shape[1]
    rp = R.OraclePolicy(D, A, hidden_sizes=tuple(hidden))
    rp.load_state_dict({k: v.detach().cpu().clone() for k, v in sd.items()})
    rp = rp.to(dtype)
    upd = R.KLPenaltyUpdater(rp)
    for _ in range(actor_clock_ahead):              # the actor's Adam clock runs ahead (zero gradients: moments stay 0)
        upd.opt_a.zero_grad()
        for prm in rp.actor.parameters():
            prm.grad = torch.zeros_like(prm)
        upd.opt_a.step()
    om, osd = old_mean.to(dtype), old_std.to(dtype)
    perm = torch.as_tensor(perm, dtype=torch.long)
    out, thetas = np.full((nsteps, 3), np.nan), {}
    for s_ in range(nsteps):
        idx = perm[s_ * batch:(s_ + 1) * batch]
        os_b = osd.expand(idx.numel(), A)
        if actor_only:
            out[s_, 2] = upd.cup_second_stage_step(obs[idx], act[idx], logp[idx], adv[idx], om[idx], os_b,
                                                   -pg_coef / ((1 - 0.99 * 0.95) / (1 - 0.99)), 0.99)
        else:
            out[s_] = upd.focops_step(obs[idx], act[idx], logp[idx], tgt_r[idx], tgt_c[idx], adv[idx], om[idx], os_b, kl_bound)
        if (s_ + 1) in checkpoints:
            thetas[s_ + 1] = R.flat_params(rp).double().numpy().copy()
    return out, thetas


@pytest.mark.parametrize("actor_only", [False, True])
def test_feature_split_kl_penalty_full_size_drift_envelope_at_humanoid_dims(dev, actor_only):
    """klpen_update_ks_kernel at BASELINE size with HumanoidVelocity's dims -- 524 288 rows x 376 / 17, ONE launch
    of 8 192 minibatch steps of 64 rows: FOCOPS' step (focops.py:312-347: both critics + the KL-penalty actor loss, 18 workgroups)
    and CUP's actor-only second stage (cup.py:370-386: 6 workgroups, the actor's own optimiser clock).  First 8 steps at 1e-5,
    per-step losses and the parameters after 8 / 64 / 512 / 2 048 / 8 192 steps under the float64 yardstick, prefixes bit-exact.
    The indicator [KL <= bound] is on for every sample here (bound = inf): over 8 192 steps a sample within rounding of a finite
    bound would flip between the float32 and float64 legs and the yardstick would measure that, not the kernel; the indicator's
    arithmetic on part of a batch is test_wide_kl_penalty_minibatch_steps_vs_oracle's."""
    import envelope as E
    from safepo import _abi
    from safepo.common.engine import WidePPOLagEngine
    from test_gpu_parity import _fill_update_problem
    torch.set_num_threads(8)
    D, A, hidden, batch = 376, 17, [64, 64], 64
    M = 4096 * 128
    pol, ref = _wide_pair(D, A, hidden, dev, seed=7 + int(actor_only))
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    problem = _synthetic_update_problem(M, D, A, seed=2028)
    obs, act, logp, tgt_r, tgt_c, adv = problem
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        dist = ref.actor(obs)
        old_mean = dist.mean + 0.05 * torch.randn(M, A, generator=g)
        old_std = dist.stddev[0] * torch.exp(0.05 * torch.randn(A, generator=g))
    kl_bound = float("inf")
    pg_coef = -0.37 if actor_only else 1 / 1.5
    ahead = 5
    cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 1e9, "batch_size": batch, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
    assert eng._feature_split_kernel_ok(eng._cfg_struct())
    _fill_update_problem(eng, problem)
    eng.mean_old.copy_(old_mean); eng.std_old.copy_(old_std)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(10))
    perm_dev = perm.to(torch.int32).to(dev)
    adv_dev = adv.to(dev).contiguous()
    theta0 = pol.theta.clone()
    ks = (8, 64, 512, 2048, 8192)
    hip = {}
    for k in ks:
        pol.theta.copy_(theta0); eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_step = 0; eng.adam_step_actor_extra = ahead
        eng.M = k * batch
        losses = eng.learning_iter_ex(perm_dev[:k * batch].contiguous(), adv_dev, _abi.ACTOR_LOSS_KL_PENALTY, kl_bound, pg_coef, actor_only)
        eng.check_sync_error()
        hip[k] = (pol.theta.double().cpu().numpy(), losses.double().cpu().numpy())
    eng.M = M
    kmax = max(ks)
    args = (sd0, problem, old_mean, old_std, perm, batch, kmax)
    l32, t32 = _kl_penalty_trajectory(*args, torch.float32, ks, actor_only, kl_bound, pg_coef, hidden, ahead)
    l64, t64 = _kl_penalty_trajectory(*args, torch.float64, ks, actor_only, kl_bound, pg_coef, hidden, ahead)
    cols = [2] if actor_only else [0, 1, 2]
    lh = hip[kmax][1]
    np.testing.assert_allclose(lh[:8, cols], l32[:8, cols], rtol=1e-5, atol=2e-6, err_msg="first 8 steps")
    what = "feature-split KL-penalty kernel, full size" + (" (actor only)" if actor_only else "")
    rep = {"loss": E.assert_loss_envelope(lh[:, cols], l32[:, cols], l64[:, cols], what, floor_rel=3e-6)}
    for k in ks:
        assert np.array_equal(hip[k][1], lh[:k], equal_nan=True), f"the {k}-step launch is not a prefix of the {kmax}-step launch"
        rep[k] = E.assert_theta_envelope(hip[k][0], t32[k], t64[k], f"{what}: theta after {k} steps", floor_abs_max=_theta_floor(3e-4, k))
    if actor_only:
        off = pol.log_std_offset
        assert torch.equal(pol.theta[:off], theta0[:off])            # the critics are not touched by CUP's second stage
    print(f"klpen_update_ks_kernel ({'actor only' if actor_only else 'full step'}), 376 / 17, 524 288 rows, drift envelope (ratio <= 1 passes):",
          {k: (round(v[0], 3), f"max |hip-f64| {v[1]['max_hip']:.2e} vs f32 {v[1]['max_f32']:.2e}") if k != "loss" else round(v, 3)
           for k, v in rep.items()})


def test_feature_split_kernels_three_consecutive_launches_on_one_stream(dev, monkeypatch):
    """the exchange tags (tag_base, never cleared) and slot parities carry over from launch to launch.  Three
    consecutive launches on one stream of the critic fit (three passes of cpo.py:541-571) and of the KL-penalty step (three passes
    of focops.py:312-347) against the oracle running the same three passes."""
    import envelope as E
    from safepo import _abi
    from safepo.common.engine import WidePPOLagEngine
    from test_gpu_parity import _fill_update_problem
    _critic_fit_check(dev, 376, 17, [64, 64], False, monkeypatch, M=128 * 9 + 50, iters=3, batch=128, expect_ks=True)
    D, A, hidden, batch, M, passes = 200, 20, [64, 64], 64, 64 * 11 + 20, 3
    pol, ref = _wide_pair(D, A, hidden, dev, seed=3)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    problem = _synthetic_update_problem(M, D, A, seed=88)
    obs, act, logp, tgt_r, tgt_c, adv = problem
    g = torch.Generator().manual_seed(13)
    with torch.no_grad():
        dist = ref.actor(obs)
        old_mean = dist.mean + 0.05 * torch.randn(M, A, generator=g)
        old_std = dist.stddev[0] * torch.exp(0.05 * torch.randn(A, generator=g))
    cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 1e9, "batch_size": batch, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
    assert eng._feature_split_kernel_ok(eng._cfg_struct())
    _fill_update_problem(eng, problem)
    eng.mean_old.copy_(old_mean); eng.std_old.copy_(old_std)
    perms = [torch.randperm(M, generator=g) for _ in range(passes)]
    nst = (M + batch - 1) // batch
    got = [eng.learning_iter_ex(p_.to(torch.int32).to(dev), adv.to(dev).contiguous(), _abi.ACTOR_LOSS_KL_PENALTY, float("inf"), 1 / 1.5,
                                False).double().cpu().numpy() for p_ in perms]
    eng.check_sync_error()
    # the oracle: the same passes; the ragged last minibatch of a pass is its own (short) minibatch, as in the reference's loop
    def oracle(dtype):
        o, ac, lp, tr, tc, ad = [t.to(dtype) for t in problem]
        rp = R.OraclePolicy(D, A, hidden_sizes=tuple(hidden))
        rp.load_state_dict({k: v.clone() for k, v in sd0.items()})
        rp = rp.to(dtype)
        upd = R.KLPenaltyUpdater(rp)
        om, osd = old_mean.to(dtype), old_std.to(dtype)
        out = []
        for p_ in perms:
            for s0 in range(0, M, batch):
                idx = p_[s0:s0 + batch]
                out.append(upd.focops_step(o[idx], ac[idx], lp[idx], tr[idx], tc[idx], ad[idx], om[idx], osd.expand(idx.numel(), A), float("inf")))
        return np.asarray(out, np.float64), R.flat_params(rp).double().numpy()
    l32, t32 = oracle(torch.float32)
    l64, t64 = oracle(torch.float64)
    lh = np.concatenate(got, 0)
    assert lh.shape == (passes * nst, 3)
    np.testing.assert_allclose(lh[:4], l32[:4], rtol=1e-5, atol=2e-6)
    E.assert_loss_envelope(lh, l32, l64, "three KL-penalty launches: losses", window=nst, floor_rel=3e-6)
    E.assert_theta_envelope(pol.theta.double().cpu().numpy(), t32, t64, "three KL-penalty launches: theta",
                            floor_abs_max=_theta_floor(3e-4, passes * nst))


def test_two_feature_split_engines_update_concurrently_on_two_streams(dev):
    """The feature-split kernels' exchange scratch (partial-sum slots, norm granules, placement census) is per (device, stream),
    like the three-workgroup kernel's: two engines of one process launched at the same time on two streams -- 9 + 9 co-resident
    workgroups exchanging through their own blocks -- must equal the same iterations run one engine after the other, bit for bit."""
    from safepo.common.engine import WidePPOLagEngine
    from test_gpu_parity import _fill_update_problem
    D, A, hidden, M = 130, 8, [64, 64], 64 * 96
    cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}

    def make(seed):
        pol, _ = _wide_pair(D, A, hidden, dev, seed=seed)
        eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
        assert eng._feature_split_kernel_ok(eng._cfg_struct())
        _fill_update_problem(eng, _synthetic_update_problem(M, D, A, seed=seed + 100))
        g = torch.Generator().manual_seed(seed)
        return eng, [torch.randperm(M, generator=g).to(torch.int32).to(dev) for _ in range(3)]
    results = {}
    for mode in ("sequential", "concurrent"):
        engs = [make(11), make(22)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        losses = [[], []]
        if mode == "sequential":
            for k, (eng, perms) in enumerate(engs):
                with torch.cuda.stream(streams[k]):
                    for p_ in perms:
                        losses[k].append(eng.learning_iter(p_).clone())
                torch.cuda.synchronize()
        else:
            for it in range(3):
                for k, (eng, perms) in enumerate(engs):
                    with torch.cuda.stream(streams[k]):
                        losses[k].append(eng.learning_iter(perms[it]).clone())
            torch.cuda.synchronize()
        for eng, _ in engs:
            eng.check_sync_error()
        results[mode] = [(e.policy.theta.clone(), e.adam_m.clone(), torch.stack(l)) for (e, _), l in zip(engs, losses)]
    for k in range(2):
        for x, y in zip(results["sequential"][k], results["concurrent"][k]):
            assert torch.equal(x, y), k
    assert not torch.equal(results["sequential"][0][0], results["sequential"][1][0])
    freed = int(_abi_lib().spo_update_scratch_release(None, 1))
    assert freed >= 2, freed                      # both streams' blocks are handed back (anything else of this process with them)


def _abi_lib():
    from safepo import _abi
    return _abi.load()


@pytest.mark.parametrize("algo", ["ppo_lag", "cppo_pid", "focops", "cup", "cpo", "pcpo", "rcpo", "trpo_lag"])
def test_default_sweep_algorithms_train_at_humanoid_dims(dev, tmp_path, algo):
    """`ActorVCritic(376, 17)` trains under every algorithm of the reference's default sweep (benchmark.py:33-44) on the
    synthetic env: main() end to end -- collect, boundary logic, GAE, update, logger columns."""
    import argparse
    import csv
    import importlib
    mod = importlib.import_module(f"safepo.single_agent.{algo}")
    args = argparse.Namespace(seed=0, use_eval=False, task="SynthSafe-v0", num_envs=16, experiment="t",
                              log_dir=str(tmp_path / algo / "task" / "run"), device="cuda", device_id=0, write_terminal=True,
                              headless=False, total_steps=2 * 16 * 32, steps_per_epoch=16 * 32, randomize=False, cost_limit=25.0,
                              lagrangian_multiplier_init=0.001, lagrangian_multiplier_lr=0.035,
                              cfg_override={"learning_iters": 2, "batch_size": 128},
                              env_kwargs={"trunc_len": 16, "obs_dim": 376, "act_dim": 17})
    out = mod.main(args, {})
    rows = list(csv.DictReader(open(tmp_path / algo / "task" / "run" / "progress.csv")))
    assert len(rows) == 2
    assert type(out["engine"]).__name__ in ("WidePPOLagEngine", "WideCPOEngine")
    assert out["policy"].obs_dim == 376 and out["policy"].act_dim == 17
    for col in ("Loss/Loss_actor", "Loss/Loss_reward_critic", "Loss/Loss_cost_critic", "Train/KL"):
        assert np.isfinite(float(rows[-1][col])), (col, rows[-1][col])


@pytest.mark.parametrize("D,A,hidden", [(60, 8, [128, 128]), (376, 17, [64, 64])])
def test_wide_steps_replayed_from_a_graph_equal_eager_launches(dev, D, A, hidden, monkeypatch):
    """The wide path at the reference's default batch of 64 is launch-bound (~70 launches per minibatch step), so a step is
    captured ONCE as a HIP graph (optimiser clocks AND learning rates on the device: spo_wide_clip_adam_dev) and replayed.  Same
    kernels, same arguments: losses and parameters after two passes equal the eager launches bit for bit -- including the ragged
    last minibatch, which runs eagerly in between, and the learning-rate change between passes (the SAME capture: round 5).
    (SPO_WIDE_KS=0: at hidden [64, 64] the engine would otherwise take the persistent feature-split kernel, not this path.)"""
    monkeypatch.setenv("SPO_WIDE_KS", "0")
    import time
    from safepo.common.engine import WidePPOLagEngine
    from test_gpu_parity import _fill_update_problem
    M, batch = 64 * 24 + 19, 64
    problem = _synthetic_update_problem(M, D, A, seed=23)
    perms = [torch.randperm(M, generator=torch.Generator().manual_seed(s_)).to(torch.int32).to(dev) for s_ in (1, 2)]
    out = {}
    for mode, gmax in (("eager", 0), ("graph", 2048)):
        pol, _ = _wide_pair(D, A, hidden, dev, seed=31)
        cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 0.02, "batch_size": batch, "learning_iters": 1, "max_grad_norm": 40.0}
        eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
        eng.graph_max_batch = gmax
        _fill_update_problem(eng, problem)
        eng.learning_iter(perms[0])                 # (first pass: lazy set-up, graph capture)
        eng.lr_factor = 0.7                         # LinearLR between epochs: the rate is read from device memory, no new capture
        l2 = eng.learning_iter(perms[1])
        eng.lr_factor = 1.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        l3 = eng.learning_iter(perms[0])            # (timed: replays only)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[mode] = (l2.cpu(), l3.cpu(), pol.theta.cpu().clone(), eng.adam_step, dt / (M // batch + 1))
        assert len(eng._step_graphs) == (1 if mode == "graph" else 0)
    assert out["eager"][3] == out["graph"][3] == 3 * (M // batch + 1)
    for i in range(3):
        assert torch.equal(out["eager"][i], out["graph"][i]), i
    print(f"wide minibatch step ({D}, {A}, {hidden}, batch {batch}): eager {out['eager'][4] * 1e6:.0f} us, graph replay {out['graph'][4] * 1e6:.0f} us")


@pytest.mark.parametrize("dims", [[60, 128, 128, 8], [376, 64, 64, 17], [7, 32, 1], [50, 200, 96, 3], [33, 256, 256, 5], [61, 100, 1],
                                  [12, 20, 24, 28, 32, 2]])
@pytest.mark.parametrize("rows", [1, 17, 64, 100, 128])
def test_mlp_forward_backward_at_small_row_counts_vs_fp64(dev, dims, rows):
    """spo_mlp_forward / spo_mlp_backward at <= 128 rows run the whole network in one launch each (csrc/mlp_small.hip; wider
    networks than its LDS images hold keep the GEMM launches -- [33, 256, 256, 5] at 100 / 128 rows).  Output, every activation
    and the flat gradient against torch autograd of the same tanh MLP (model.py:30-48) in float64, gated by the float32
    evaluation's own distance to float64: |HIP - f64| <= 3 |f32 - f64| + 1e-6 of the array's scale.  Ragged widths (100, 61,
    20 ...), one output, 17 outputs, 1 to 5 layers."""
    from safepo import _abi
    lib = _abi.load()
    g = torch.Generator().manual_seed(sum(dims) + rows)
    n = len(dims) - 1
    Ws = [torch.randn(dims[l + 1], dims[l], generator=g, dtype=torch.float64) / dims[l] ** 0.5 for l in range(n)]
    bs = [torch.randn(dims[l + 1], generator=g, dtype=torch.float64) * 0.1 for l in range(n)]
    x = torch.randn(rows, dims[0], generator=g, dtype=torch.float64)
    dout = torch.randn(rows, dims[-1], generator=g, dtype=torch.float64)

    def run(dt):
        W = [w.detach().to(dt).clone().requires_grad_() for w in Ws]
        b = [v.detach().to(dt).clone().requires_grad_() for v in bs]
        h, acts = x.to(dt), []
        for l in range(n):
            h = h @ W[l].T + b[l]
            if l + 1 < n:
                h = torch.tanh(h)
            acts.append(h)
        (h * dout.to(dt)).sum().backward()
        flat = torch.cat([torch.cat([W[l].grad.reshape(-1), b[l].grad]) for l in range(n)])
        return [a_.detach().double() for a_ in acts], flat.double()
    acts64, g64 = run(torch.float64)
    acts32, g32 = run(torch.float32)
    net = _abi.MlpNet(n_layers=n)
    for k, d in enumerate(dims):
        net.dims[k] = d
    theta = torch.cat([torch.cat([Ws[l].reshape(-1), bs[l]]) for l in range(n)]).float().to(dev)
    ws = torch.zeros(int(lib.spo_mlp_workspace_floats(net, rows)), device=dev)
    grad = torch.full_like(theta, float("nan"))
    scratch = torch.zeros(int(lib.spo_mlp_backward_scratch_floats(net, rows)), device=dev)
    xd, dd = x.float().to(dev).contiguous(), dout.float().to(dev).contiguous()
    _abi.check(lib.spo_mlp_forward(_abi.ptr(theta), net, _abi.ptr(xd), rows, _abi.ptr(ws), _abi.stream_ptr()), "fwd")
    _abi.check(lib.spo_mlp_backward(_abi.ptr(theta), net, _abi.ptr(xd), rows, _abi.ptr(ws), _abi.ptr(dd), _abi.ptr(grad), _abi.ptr(scratch),
                                    _abi.stream_ptr()), "bwd")
    off = 0
    for l in range(n):
        got = ws[off:off + rows * dims[l + 1]].view(rows, dims[l + 1]).double().cpu()
        off += rows * dims[l + 1]
        err, yard = (got - acts64[l]).abs(), (acts32[l] - acts64[l]).abs()
        assert bool((err <= 3 * yard + 1e-6 * acts64[l].abs().max()).all()), (l, float(err.max()), float(yard.max()))
    got = grad.double().cpu()
    assert bool(torch.isfinite(got).all())
    err, yard = (got - g64).abs(), (g32 - g64).abs()
    o = 0
    for l in range(n):                                   # per parameter block: its own scale
        for cnt in (dims[l + 1] * dims[l], dims[l + 1]):
            sl = slice(o, o + cnt)
            assert bool((err[sl] <= 3 * yard[sl] + 2e-6 * g64[sl].abs().max()).all()), (l, cnt, float(err[sl].max()), float(yard[sl].max()))
            o += cnt
