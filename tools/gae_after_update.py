"""Development aid (GPU box): per-dispatch GAE durations right after a long persistent update launch (3 busy CUs), the
context bench.py measures them in, printed as a series to see how long the ramp lasts."""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))
from safepo.common.engine import PPOLagEngine
from safepo.common.model import ActorVCritic
dev = torch.device("cuda:0")
N, T, D, A = 4096, 128, 60, 8
cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
pol = ActorVCritic(D, A).to(dev)
eng = PPOLagEngine(pol, N, T, cfg, dev)
b = eng.buffer
for k in ("obs", "act", "reward", "cost", "value_r", "value_c", "target_value_r", "target_value_c"):
    b.data[k].normal_()
b.data["log_prob"].fill_(-8.0)
b.adv_mix.normal_()
b.reward_fold.copy_(b.data["reward"]); b.cost_fold.copy_(b.data["cost"])
b.ptr = T
b._fold_cols = T
b.seg_end[:, -1] = 1
b.compute_gae(None)
perm = torch.randperm(N * T, device=dev).to(torch.int32)
for rep in range(3):
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        eng.learning_iter(perm)
    torch.cuda.synchronize()
    d = np.asarray(b.time_scan_dispatches(400, warm=0)) * 1e6
    print("after update: first 10", np.round(d[:10], 2).tolist())
    for lo in range(0, 400, 50):
        print(f"   dispatches {lo:3d}-{lo+49:3d}: mean {d[lo:lo+50].mean():6.2f} median {np.median(d[lo:lo+50]):6.2f} max {d[lo:lo+50].max():6.2f}")
    d2 = np.asarray(b.time_scan_dispatches(100, warm=0)) * 1e6
    print(f"   immediately again: mean {d2.mean():.2f} median {np.median(d2):.2f}")
