"""Debug tool: interval breakdown of the main + helper form of the persistent update kernel (actor workgroup, wave 0 of each
role).  Usage (GPU box): python tools/phase_profile_h.py"""
import os, sys, time, torch
os.environ["SPO_UPDATE_FORM"] = "2"       # this tool reads the MAIN + HELPER kernel's counters (the row-split kernel: tools/phase_profile_rs.py)
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
eng.learning_iter(perm)
lib.spo_debug_set_update_profile(prof.data_ptr())
torch.cuda.synchronize()
t0 = time.time(); eng.learning_iter(perm); torch.cuda.synchronize(); dt = time.time() - t0
lib.spo_debug_set_update_profile(None)
steps = N * T // 64
p = prof.cpu().view(3, 10).numpy()
print(f"instrumented launch: {dt*1e6/steps:.2f} us/step")
main = ["settle + x^T", "L1", "wait Q2", "L2", "wait Xd", "L3 loss bwd stage", "wait B_stage", "dW1 -> G1", "wait P1", "dW2 dW3 -> G"]
helper = ["wait P1", "Adam W1 (speculative)", "wait P3", "norms of W2 W3 + tags", "wait Q2", "Adam W3 b3 log_std", "wait Xd + B_stage", "", "poll norms, coefficient", "Adam W2 (speculative)"]
spec = os.environ.get("SPO_UPDATE_SPEC", "1") != "0"
if spec:
    helper = ["wait P1", "W1: L2 norm Adam (spec)", "wait P3", "W2: L2 norm Adam (spec)", "wait Q2", "W3 b3 log_std (spec), norm out", "wait B_stage", "",
              "poll norms, verdict", "wait Xd"]
sub = ["prefetch issue + L3", "loss", "dO -> dZ2", "dZ2 -> dZ1", "stage 4 images + dO", "loss / dls sums"]
for row, names in ((0, main), (2, sub), (1, helper)):
    tot = p[row].sum()
    print({0: "main", 1: "helper", 2: "main, inside 'L3 loss bwd stage':"}[row] + f" wave 0 of the actor: total {tot/steps:.0f} cycles/step")
    for i, n in enumerate(names):
        if n:
            print(f"   {n:22s} {p[row][i]/steps:8.0f} cyc  {100*p[row][i]/max(tot,1):5.1f}%")
