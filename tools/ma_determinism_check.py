"""Diagnostic: is the config-5-shape MAPPO-L update deterministic?  The same three ppo_update steps from the same state, repeated;
gradients / parameters compared bit for bit across the repeats (a difference = a race in one of the kernels)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(reps=6, steps=3, streams=1):
    import test_gpu_ma_full_size as F
    from oracle import ma_restatement as MR
    import ma_yardstick as Y
    from safepo.multi_agent.mappolag import MAPPO_L_Policy, MAPPO_L_Trainer
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    cfg = F._cfg(dev, train_streams=bool(streams))
    pol = MAPPO_L_Policy(cfg, F._Sp(F.D), F._Sp(F.S), F._Sp(F.A))
    with torch.no_grad():
        for net in pol.networks():
            net.theta.add_(0.05 * torch.randn_like(net.theta))
    nets0 = Y.nets_like(pol, cfg["std_x_coef"], cfg["std_y_coef"])
    s = F._sample(F.ROWS, seed=11)
    with torch.no_grad():
        lp = MR.log_probs(nets0["actor"](s["obs"]), nets0["actor"].std(), s["actions"])
        s["old_logp"] = lp + 0.03 * torch.randn(F.ROWS, F.A, generator=torch.Generator().manual_seed(3))
    sample = (s["share_obs"], s["obs"], None, None, s["actions"], s["value_preds"], s["returns"], None, s["active_masks"],
              s["old_logp"], s["adv"], None, s["factor"], s["cost_preds"], s["cost_returns"], None, s["cost_adv"], s["aver_episode_costs"])
    sample = tuple(t.to(dev) if torch.is_tensor(t) else t for t in sample)
    theta0 = [n.theta.clone() for n in pol.networks()]
    opts = [pol.actor_optimizer, pol.critic_optimizer, pol.cost_optimizer]
    names = ["actor", "critic", "cost_critic"]
    base = None
    for rep in range(reps):
        if rep >= 2:
            # poison the caching allocator: blocks of many sizes filled with NaN (odd repeats) or 1e30 (even), then released --
            # a kernel that reads workspace it did not write now computes on garbage
            junk = [torch.full((n,), float("nan") if rep % 2 else 1e30, device=dev) for n in
                    (1 << 28, 1 << 27, 1 << 26, 1 << 26, 1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 70000, 40940, 4096, 1024, 256)]
            del junk
        tr = MAPPO_L_Trainer(cfg, pol)
        for n, t0 in zip(pol.networks(), theta0):
            n.theta.copy_(t0)
        for o in opts:
            o.m.zero_(); o.v.zero_(); o.t = 0; o.grad.zero_()
        rec = []
        for k in range(steps):
            tr.ppo_update(sample)
            torch.cuda.synchronize()
            rec.append([o.grad.clone() for o in opts] + [n.theta.clone() for n in pol.networks()])
        if base is None:
            base = rec
            continue
        for k in range(steps):
            for i in range(6):
                a, b = base[k][i], rec[k][i]
                if not torch.equal(a, b):
                    d = (a - b).abs()
                    idx = int(d.argmax())
                    print(f"rep {rep} step {k + 1} {'grad' if i < 3 else 'theta'} {names[i % 3]}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e} "
                          f"at flat index {idx} (value {float(a[idx]):.4e}), scale {float(a.abs().max()):.3e}")
    print("done: train_streams", streams, "reps", reps)


if __name__ == "__main__":
    main(streams=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
