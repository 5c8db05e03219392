"""Development aid (GPU box): us per 64-row minibatch step of the feature-split persistent update kernel (csrc/update_ks.hip) for a
few (obs_dim, act_dim), next to the LDS-resident kernel at 60 / 8 and the launch-per-layer wide step (SPO_WIDE_KS=0).
    python tools/ks_bench.py [D,A ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))


def one(D, A, steps=2048, reps=3):
    from safepo.common.engine import PPOLagEngine, WidePPOLagEngine
    from safepo.common.model import ActorVCritic
    dev = torch.device("cuda:0")
    M = steps * 64
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    torch.manual_seed(0)
    pol = ActorVCritic(D, A).to(dev)
    eng = (PPOLagEngine if pol.kernels_supported("ppo") else WidePPOLagEngine)(pol, 1, M, cfg, dev)
    b = eng.buffer
    g = torch.Generator(device=dev).manual_seed(1)
    for k in ("obs", "act", "target_value_r", "target_value_c"):
        b.data[k].normal_(generator=g)
    b.data["log_prob"].copy_(-0.92 * A - 0.5 * (b.data["act"] ** 2).sum(-1))
    b.adv_mix.normal_(generator=g)
    perm = torch.randperm(M, device=dev, generator=g).to(torch.int32)
    mode = os.environ.get("KS_BENCH_MODE", "ppo")           # ppo | focops | cup2 (CUP's actor-only second stage)
    if mode != "ppo":
        from safepo import _abi
        eng.snapshot_old_distribution()
        run = lambda: eng.learning_iter_ex(perm, b.adv_mix.view(-1), _abi.ACTOR_LOSS_KL_PENALTY, 0.02 if mode == "focops" else float("inf"),
                                           1 / 1.5 if mode == "focops" else -0.3, mode == "cup2")
    else:
        run = lambda: eng.learning_iter(perm)
    times = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.time()
        losses = run()
        torch.cuda.synchronize()
        times.append(time.time() - t0)
    eng.check_sync_error()
    prof = None
    if hasattr(eng.lib, "spo_debug_ks_profile") or os.environ.get("SPO_KS_PROF"):
        import ctypes
        try:
            fn = eng.lib.spo_debug_ks_profile
            buf = (ctypes.c_ulonglong * 16)()
            if fn(buf) == 0:
                names = ["settle + partial L1", "partial stores", "partial polls + sums", "tanh, L2, L3", "loss + backward", "staging + barrier",
                         "weight gradients", "L2 terms, norms", "granule exchange", "Adam + barrier"]
                prof = {n: round(buf[i] / steps) for i, n in enumerate(names)}
                prof["poll iterations per step: owner / broadcast"] = (round(buf[10] / steps, 2), round(buf[11] / steps, 2))
        except AttributeError:
            pass
    if prof:
        print("   cycles per step (thread 0 of the last workgroup):", prof, "total", sum(v for v in prof.values() if not isinstance(v, tuple)), "co-resident" if buf[15] else "write-through")
    return {"obs_dim": D, "act_dim": A, "engine": type(eng).__name__, "mode": mode, "feature_split": os.environ.get("SPO_WIDE_KS", "1") != "0",
            "us_per_minibatch_step": round(min(times[1:]) * 1e6 / steps, 2), "loss_finite": bool(torch.isfinite(losses).all())}


if __name__ == "__main__":
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [(60, 8), (60, 20), (200, 20), (376, 17), (512, 32)]
    for D, A in shapes:
        print(json.dumps(one(D, A, steps=2048 if os.environ.get("SPO_WIDE_KS", "1") != "0" or D <= 128 and A <= 16 else 256)), flush=True)
