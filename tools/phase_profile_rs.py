"""Debug tool: interval breakdown of the ROW-SPLIT persistent update kernel (csrc/update_rs.hip): cycles per interval of the step
for column wave 0 and optimiser wave 0 of the last workgroup (the actor's second row group).
Usage (GPU box): python tools/phase_profile_rs.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))
os.environ.setdefault("SPO_UPDATE_FORM", "3")
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
prof = torch.zeros(64, dtype=torch.int64, device=dev)
lib = _abi.load()
perm = torch.randperm(N * T, device=dev).to(torch.int32)
eng.learning_iter(perm)
torch.cuda.synchronize()
t0 = time.time(); eng.learning_iter(perm); torch.cuda.synchronize(); dt0 = time.time() - t0
lib.spo_debug_set_update_profile(prof.data_ptr())
torch.cuda.synchronize()
t0 = time.time(); eng.learning_iter(perm); torch.cuda.synchronize(); dt = time.time() - t0
lib.spo_debug_set_update_profile(None)
eng.check_sync_error()
steps = N * T // 64
p = prof.cpu().view(-1)[:36].view(3, 12).numpy()
print(f"plain launch: {dt0*1e6/steps:.2f} us/step; instrumented launch: {dt*1e6/steps:.2f} us/step")
col = ["settle + x^T image", "wait b1 (W1 in place)", "L1 half + h1^T", "wait b2 (W2)", "h1 halves, L2 half + h2^T", "wait b3 (W3)",
       "h2 halves, L3, loss, dO->dZ2 half", "wait b4", "gather issue, dZ2->dZ1 half", "wait b5", "", ""]
opt = ["wait b3 + loss log + preloads", "wait b4 (images)", "dW2 dW3 + stores", "wait b5 (dZ1^T)", "dW1 + stores",
       "layers 2/3: poll, sums, L2 terms", "layer 1: poll, sums, L2, share out, Adam W1", "wait b1 + norms, coefficient", "(redo,) Adam W2", "wait b2 + Adam W3 b3 log_std",
       "(poll retries, layers 2/3)", "(poll retries, layer 1)"]
for row, names, title in ((0, col, "column wave 0"), (1, opt, "optimiser wave 0")):
    tot = p[row][:10].sum()
    print(f"{title} of the actor's last row group: total {tot/steps:.0f} cycles/step")
    for i, n in enumerate(names):
        if n:
            unit = "" if i >= 10 else f"  {100*p[row][i]/max(tot,1):5.1f}%"
            print(f"   {n:34s} {p[row][i]/steps:9.1f}{' /step' if i >= 10 else ' cyc'}{unit}")
