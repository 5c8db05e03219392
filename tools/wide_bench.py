"""Development aid (GPU box): time the wide-network minibatch step (safepo.common.engine.WidePPOLagEngine.minibatch_step:
three forwards, loss kernel, three backwards, joint clip + Adam) for a few (hidden_sizes, batch) pairs.
    python tools/wide_bench.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))


def one(hidden, batch, steps, D=60, A=8, force_wide=False, **cfg_kw):
    """force_wide: keep the launch-per-layer wide step where the engine would take the persistent feature-split kernel."""
    from safepo.common.engine import WidePPOLagEngine
    prev = os.environ.get("SPO_WIDE_KS")
    if force_wide:
        os.environ["SPO_WIDE_KS"] = "0"
    try:
        return _one(WidePPOLagEngine, hidden, batch, steps, D, A, **cfg_kw)
    finally:
        if force_wide:
            os.environ.pop("SPO_WIDE_KS", None) if prev is None else os.environ.__setitem__("SPO_WIDE_KS", prev)


def _one(WidePPOLagEngine, hidden, batch, steps, D, A, **cfg_kw):
    from safepo.common.model import ActorVCritic
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    pol = ActorVCritic(D, A, hidden_sizes=hidden).to(dev)
    M = batch * steps
    cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 1e9, "batch_size": batch, "learning_iters": 1, "max_grad_norm": 40.0}
    cfg.update(cfg_kw)
    eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    b = eng.buffer
    for k in ("obs", "act", "target_value_r", "target_value_c"):
        b.data[k].normal_(generator=g)
    b.data["log_prob"].copy_(-A * 0.92 - 0.5 * (b.data["act"] ** 2).sum(-1))
    b.adv_mix.normal_(generator=g)
    perm = torch.randperm(M, device=dev, generator=g).to(torch.int32)
    eng.learning_iter(perm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.learning_iter(perm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sizes = [D] + list(hidden)
    mac = sum(a * c for a, c in zip(sizes[:-1], sizes[1:]))
    flops = 3 * 2.0 * batch * (2 * (mac + sizes[-1]) + (mac + sizes[-1] * A))        # fwd + 2x bwd, two critics + actor
    ks = eng._feature_split_kernel_ok(eng._cfg_struct())
    if ks:
        eng.check_sync_error()
    rows = (not ks) and eng.wide.rows_grad_ok(batch)
    fused = rows and os.environ.get("SPO_WIDE_ROWS_FUSED", "1") != "0"
    unroll = max(1, int(os.environ.get("SPO_WIDE_GRAPH_UNROLL", "8")))
    rows_label = (f"row-group gradient kernel: 3 networks x {(batch + 15) // 16} workgroups of 16 rows in one launch (csrc/mlp_rows.hip) + "
                  + ("2 optimiser launches" if fused else "group sum + 3 optimiser launches")
                  + f", {unroll} steps per replayed HIP graph")
    return {"hidden_sizes": hidden, "batch": batch, "us_per_minibatch_step": round(dt * 1e6, 2 if ks else 1), "tflops": round(flops / dt / 1e12, 2),
            "params": int(pol.theta.numel()),
            "kernel": (f"ppo_update_ks_kernel: ONE persistent launch for the {steps} steps, 3 networks x {(D + 63) // 64} feature slices = "
                       f"{3 * ((D + 63) // 64)} workgroups (csrc/update_ks.hip)") if ks else rows_label if rows else
                      "launch-per-layer wide step (csrc/ma_net.hip kernels), full minibatches replayed from one HIP graph"}


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--isaac":
    print(json.dumps(one([1024, 1024, 512], 8192, 8, use_critic_norm=False, use_value_coefficient=True, max_grad_norm=1.0)))
    sys.exit(0)
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--small":
    # the launch-latency regime: the reference's default batch of 64 on a widened net and on HumanoidVelocity's dims
    print(json.dumps([one([128, 128], 64, 256), one([64, 64], 64, 256, D=376, A=17)]))
    sys.exit(0)
if __name__ == "__main__":
    out = [one([1024, 1024, 512], 8192, 8, use_critic_norm=False, use_value_coefficient=True, max_grad_norm=1.0),
           one([256, 256], 2048, 16), one([128, 128], 64, 256)]
    print(json.dumps(out))
