import os, sys, time, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from safepo.common.engine import WidePPOLagEngine
from safepo.common.model import ActorVCritic
dev = torch.device("cuda:0")
steps, batch, D, A, hidden = 1024, 64, 60, 8, [128, 128]
M = steps * batch
cfg = {"hidden_sizes": hidden, "gamma": 0.99, "target_kl": 1e9, "batch_size": batch, "learning_iters": 1, "max_grad_norm": 40.0}
pol = ActorVCritic(D, A, hidden_sizes=hidden).to(dev)
eng = WidePPOLagEngine(pol, 1, M, cfg, dev)
b = eng.buffer
for k in ("obs", "act", "target_value_r", "target_value_c"): b.data[k].normal_()
b.data["log_prob"].copy_(-0.92 * A - 0.5 * (b.data["act"] ** 2).sum(-1)); b.adv_mix.normal_()
perm = torch.randperm(M, device=dev).to(torch.int32)
eng.learning_iter(perm); torch.cuda.synchronize()
(g, win, keep), = eng._step_graphs.values()
win.load(perm.long())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {(t1 - t0) / steps * 1e6:.1f} us per replay; total {(t2 - t0) / steps * 1e6:.1f} us per step")
