"""Debug tool: per-phase shader-cycle breakdown of the persistent PPO-Lag update kernel.
Usage (GPU box): python tools/phase_profile.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))
from safepo import _abi
from safepo.common.engine import PPOLagEngine
from safepo.common.model import ActorVCritic
dev = torch.device("cuda:0")
N, T, D, A = 4096, 128, 60, 8
cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
pol = ActorVCritic(D, A).to(dev)
eng = PPOLagEngine(pol, N, T, cfg, dev)
b = eng.buffer
for k in ("obs", "act", "log_prob", "target_value_r", "target_value_c"):
    b.data[k].normal_()
b.data["log_prob"].fill_(-8.0)
b.adv_mix.normal_()
prof = torch.zeros(30, dtype=torch.int64, device=dev)
lib = _abi.load()
perm = torch.randperm(N * T, device=dev).to(torch.int32)
eng.learning_iter(perm)                       # warm
lib.spo_debug_set_update_profile(prof.data_ptr())
torch.cuda.synchronize()
import time
t0 = time.time(); eng.learning_iter(perm); torch.cuda.synchronize(); dt = time.time() - t0
lib.spo_debug_set_update_profile(None)
names = ["fetch-issue+XT stage", "forward", "loss+backward dH", "stage writes+partials", "barrier1", "dW GEMMs",
         "grads/L2/norm", "barrier + publish + speculative Adam", "granule wait + clip decision (+ redo)", "final barrier"]
steps = N * T // 64
p = prof.cpu().view(3, 10).numpy()
print(f"instrumented launch: {dt*1e6/steps:.2f} us/step")
for net in range(3):
    tot = p[net].sum()
    print(f"net {net}: total {tot/steps:.0f} cycles/step")
    for i, n in enumerate(names):
        print(f"   {n:26s} {p[net][i]/steps:8.0f} cyc  {100*p[net][i]/tot:5.1f}%")
