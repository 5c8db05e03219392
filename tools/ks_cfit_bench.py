"""Development aid (GPU box): us per 128-row minibatch step of WideCPOEngine.critic_fit at HumanoidVelocity's dims -- the
feature-split persistent kernel (csrc/update_ks.hip, two networks) against the launch-per-layer wide path (SPO_WIDE_KS=0).
    python tools/ks_cfit_bench.py [D,A ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))


def one(D, A, M, batch=128, iters=2):
    from safepo.common.model import ActorVCritic
    from safepo.single_agent import cpo
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    pol = ActorVCritic(D, A).to(dev)
    cfg = dict(cpo.default_cfg)
    cfg.update(learning_iters=iters, batch_size=batch)
    eng = cpo.make_engine(pol, 1, M, cfg, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    b = eng.buffer
    for k in ("obs", "target_value_r", "target_value_c"):
        b.data[k].normal_(generator=g)
    eng._set_stale_actor_grad(torch.zeros(eng.flat_grad.numel() - eng.ls_off, device=dev))
    eng.critic_fit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit = eng.critic_fit()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = iters * ((M + batch - 1) // batch)
    return {"obs_dim": D, "act_dim": A, "engine": type(eng).__name__, "rows": M, "batch": batch,
            "feature_split": bool(getattr(eng, "_feature_split_critic_fit_ok", lambda c: False)(eng._cfg_struct())),
            "us_per_minibatch_step": round(dt * 1e6 / steps, 2), "loss_r": fit["loss_r"]}


if __name__ == "__main__":
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [(376, 17)]
    M = int(os.environ.get("KS_CFIT_ROWS", 128 * 2048))
    for D, A in shapes:
        print(json.dumps(one(D, A, M)))
