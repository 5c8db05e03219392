"""Measured numerical deviation of the minibatch loss/gradient kernels: HIP fp32 path and torch-CPU fp32 (the
reference's arithmetic) are both compared with an fp64 evaluation of the same minibatch.  GPU box only."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))
from oracle import restatement as R
from safepo import _abi
from safepo.common.engine import PPOLagEngine
from safepo.common.model import ActorVCritic
dev = torch.device("cuda:0")
D, A, M = 60, 8, 64
worst = {}
for seed in range(20):
    torch.manual_seed(seed)
    pol = ActorVCritic(D, A).to(dev)
    with torch.no_grad():
        pol.theta.add_(0.05 * torch.randn_like(pol.theta))          # move away from init
    g = torch.Generator().manual_seed(100 + seed)
    obs, act = torch.randn(M, D, generator=g), torch.randn(M, A, generator=g)
    logp = -A * 0.9 - 0.5 * (act ** 2).sum(-1) + 0.2 * torch.randn(M, generator=g)
    tr, tc, adv = torch.randn(M, generator=g), torch.rand(M, generator=g), torch.randn(M, generator=g)
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = PPOLagEngine(pol, 1, M, cfg, dev)
    b = eng.buffer
    b.data["obs"].copy_(obs.view(1, M, D)); b.data["act"].copy_(act.view(1, M, A)); b.data["log_prob"].copy_(logp.view(1, M))
    b.data["target_value_r"].copy_(tr.view(1, M)); b.data["target_value_c"].copy_(tc.view(1, M)); b.adv_mix.copy_(adv.view(1, M))
    idx = torch.arange(M, dtype=torch.int32, device=dev)
    d = b.data
    _abi.check(eng.lib.spo_ppo_lag_grad(_abi.ptr(pol.theta), _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]),
                                        _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(b.adv_mix),
                                        _abi.ptr(idx), M, M, eng._cfg_struct(), _abi.ptr(eng.flat_grad), _abi.ptr(eng.losses3),
                                        _abi.stream_ptr()), "grad")
    g_hip, l_hip = eng.flat_grad.cpu().double().numpy(), eng.losses3.cpu().double().numpy()
    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        ref = R.OraclePolicy(D, A).to(dt)
        ref.load_state_dict({k: v.cpu().to(dt) for k, v in pol.state_dict().items()})
        total, lpi, lr, lc = R.ppo_lag_losses(ref, obs.to(dt), act.to(dt), logp.to(dt), tr.to(dt), tc.to(dt), adv.to(dt))
        ref.zero_grad(); total.backward()
        res[name] = (R.flat_grads(ref).double().numpy(), np.array([lr.item(), lc.item(), lpi.item()]))
    g64, l64 = res["f64"]; g32, l32 = res["f32"]
    sc = np.abs(g64).max()
    for tag, gg, ll in (("hip", g_hip, l_hip), ("torch_f32", g32, l32)):
        worst.setdefault(tag + "_grad_err_over_maxgrad", []).append(np.abs(gg - g64).max() / sc)
        worst.setdefault(tag + "_loss_rel", []).append((np.abs(ll - l64) / np.abs(l64)).max())
    worst.setdefault("hip_vs_torch_f32_grad", []).append(np.abs(g_hip - g32).max() / sc)
    worst.setdefault("hip_vs_torch_f32_loss_rel", []).append((np.abs(l_hip - l32) / np.abs(l32)).max())
for k, v in worst.items():
    print(f"{k:36s} max {max(v):.2e}   median {np.median(v):.2e}")
