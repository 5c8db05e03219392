"""Development aid (GPU box): the compute side of ONE data-parallel minibatch step at the feature-split kernel's dims (hidden
[64, 64], obs_dim <= 512): spo_ppo_lag_grad_ks + spo_wide_clip_adam (round 6) next to what the wide engine's eager minibatch step
launches otherwise -- the row-group gradient kernel (csrc/mlp_rows.hip), or with SPO_WIDE_ROWS=0 the launch-per-layer step of
rounds 4-5 (gather, 3 forwards, loss, 3 backwards, clip + Adam) -- all launched eagerly as under data parallelism (a host collective sits
between the gradient and the clip; none here: world 1, so the numbers are the launches alone).
    python tools/ks_grad_bench.py [D,A ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))


def one(D, A, steps=512, reps=3):
    from safepo import _abi
    from safepo.common.engine import WidePPOLagEngine
    from safepo.common.model import ActorVCritic
    dev = torch.device("cuda:0")
    M = steps * 64
    cfg_d = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    torch.manual_seed(0)
    pol = ActorVCritic(D, A).to(dev)
    eng = WidePPOLagEngine(pol, 1, M, cfg_d, dev)
    b, d, w, lib = eng.buffer, eng.buffer.data, eng.wide, eng.lib
    g = torch.Generator(device=dev).manual_seed(1)
    for k in ("obs", "act", "target_value_r", "target_value_c"):
        d[k].normal_(generator=g)
    d["log_prob"].copy_(-0.92 * A - 0.5 * (d["act"] ** 2).sum(-1))
    b.adv_mix.normal_(generator=g)
    perm = torch.randperm(M, device=dev, generator=g).to(torch.int32)
    perm64 = perm.long()
    cfg = eng._cfg_struct()
    losses = torch.zeros((steps, 3), device=dev)

    def grad_kernel_pass():
        for k in range(steps):
            idx = perm[k * 64:(k + 1) * 64]
            _abi.check(lib.spo_ppo_lag_grad_ks(_abi.ptr(pol.theta), _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]),
                                               _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(b.adv_mix),
                                               _abi.ptr(idx), 64, cfg, _abi.ptr(eng.flat_grad), _abi.ptr(losses[k]),
                                               _abi.ptr(eng.sync_ws), _abi.stream_ptr()), "grad_ks")
            _abi.check(lib.spo_wide_clip_adam(_abi.ptr(pol.theta), _abi.ptr(eng.flat_grad), _abi.ptr(eng.adam_m), _abi.ptr(eng.adam_v),
                                              w.P, w.off_c, w.off_ls, w.off_ls, cfg, eng.adam_step, _abi.ptr(losses[k]),
                                              _abi.ptr(eng.scal4), _abi.ptr(eng.loss_partials), eng.loss_partials.numel(),
                                              _abi.stream_ptr()), "clip_adam")
            eng.adam_step += 1

    def per_layer_pass():
        for k in range(steps):
            eng.minibatch_step(perm64[k * 64:(k + 1) * 64], losses[k], cfg=cfg)

    out = {"obs_dim": D, "act_dim": A, "steps": steps}
    # what WidePPOLagEngine.minibatch_step launches eagerly in this process: the row-group gradient kernel of round 6
    # (csrc/mlp_rows.hip) + group sum + clip / Adam, or -- SPO_WIDE_ROWS=0 -- the launch-per-network step of rounds 4-5
    other = "row_group_kernel_step" if eng.wide.rows_grad_ok(64) else "launch_per_layer"
    for name, fn in (("grad_kernel_plus_clip_adam", grad_kernel_pass), (other, per_layer_pass)):
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.time()
            fn()
            torch.cuda.synchronize()
            ts.append(time.time() - t0)
        out[name + "_us_per_step"] = round(min(ts[1:]) / steps * 1e6, 2)
    eng.check_sync_error()
    return out


if __name__ == "__main__":
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(376, 17), (130, 8)]
    for D, A in shapes:
        print(json.dumps(one(D, A)), flush=True)
