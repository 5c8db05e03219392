"""Development aid (GPU box): time the persistent PPO-Lag update launch (8192 minibatch steps of 64 rows, obs 60, act 8) for
every library given on the command line (default: the in-tree build), uninstrumented, and check the variants against the
first one (losses / parameters after the launch must agree to rounding).
    python tools/update_ab.py [lib1.so lib2.so ...]"""
import os
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))
    import time
    import torch
    from safepo.common.engine import PPOLagEngine
    from safepo.common.model import ActorVCritic
    dev = torch.device("cuda:0")
    N, T, D, A = 4096, 128, 60, 8
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    torch.manual_seed(0)
    pol = ActorVCritic(D, A).to(dev)
    eng = PPOLagEngine(pol, N, T, cfg, dev)
    b = eng.buffer
    g = torch.Generator(device=dev).manual_seed(1)
    for k in ("obs", "act", "target_value_r", "target_value_c"):
        b.data[k].normal_(generator=g)
    b.data["log_prob"].copy_(-8 * 0.92 - 0.5 * (b.data["act"] ** 2).sum(-1))
    b.adv_mix.normal_(generator=g)
    perm = torch.randperm(N * T, device=dev, generator=g).to(torch.int32)
    theta0 = pol.theta.clone()
    times = []
    for rep in range(4):
        pol.theta.copy_(theta0); eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_step = 0
        torch.cuda.synchronize()
        t0 = time.time()
        losses = eng.learning_iter(perm)
        torch.cuda.synchronize()
        times.append(time.time() - t0)
    eng.check_sync_error()
    steps = N * T // 64
    print(json.dumps({"us_per_step": round(min(times[1:]) * 1e6 / steps, 3), "all": [round(t * 1e6 / steps, 3) for t in times],
                      "loss_sum": losses.double().sum(0).tolist(), "theta_sum": float(pol.theta.double().sum()),
                      "theta_abs": float(pol.theta.double().abs().sum())}))
    sys.exit(0)
libs = sys.argv[1:] or [""]
base = None
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["SPO_LIB_PATH"] = os.path.abspath(lib)
        env["SPO_LIB_OVERRIDE"] = "1"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(lib or "in-tree", "FAILED", r.stderr[-400:])
        continue
    rec = json.loads(line[-1])
    if base is None:
        base = rec
    dl = max(abs(a - b) / (abs(b) + 1e-9) for a, b in zip(rec["loss_sum"], base["loss_sum"]))
    print(f"{os.path.basename(lib) or 'in-tree':40s} {rec['us_per_step']:8.3f} us/step  runs {rec['all']}  "
          f"rel loss-sum diff vs first {dl:.2e}  theta |sum| diff {abs(rec['theta_abs'] - base['theta_abs']):.3e}", flush=True)
