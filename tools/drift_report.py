"""Diagnostic (GPU box): per-step losses and checkpoint parameters of the HIP update kernel against the oracle's float32 and
float64 trajectories on the full-size problem of tests/test_gpu_parity.py::test_full_size_update_parity_drift_envelope.
Writes gpurun_out/drift_report.npz and prints where the HIP trajectory leaves the fp32 reference's own distance."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import envelope as E
from oracle import restatement as R
from test_gpu_parity import _fill_update_problem, _hip_prefix_runs, _synthetic_update_problem


def main():
    from safepo.common.engine import PPOLagEngine
    from safepo.common.model import ActorVCritic
    dev = torch.device("cuda:0")
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    N, T, D, A = 4096, 128, 60, 8
    M = N * T
    torch.manual_seed(11)
    pol = ActorVCritic(D, A).to(dev)
    problem = _synthetic_update_problem(M, D, A, seed=seed)
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = PPOLagEngine(pol, N, T, cfg, dev)
    _fill_update_problem(eng, problem)
    sd0 = {k: v.detach().cpu().clone() for k, v in pol.state_dict().items()}
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(6))
    ks = tuple(k for k in (8, 64, 512, 1024, 2048, 3072, 4096, 5120, 6144, 7168, 7232, 7296, 8192) if k <= K)
    runs = _hip_prefix_runs(eng, pol, pol.theta.clone(), perm.to(torch.int32).to(dev), 64, ks)
    l32, t32 = E.oracle_trajectory(sd0, problem, perm, 64, max(ks), torch.float32, ks)
    l64, t64 = E.oracle_trajectory(sd0, problem, perm, 64, max(ks), torch.float64, ks)
    lh = runs[max(ks)][1]
    out = {"lh": lh, "l32": l32, "l64": l64}
    for k in ks:
        out[f"th_{k}"], out[f"t32_{k}"], out[f"t64_{k}"] = runs[k][0], t32[k], t64[k]
        d_h, d_32 = np.abs(runs[k][0] - t64[k]), np.abs(t32[k] - t64[k])
        print(k, "theta |hip-f64| l2 %.3e max %.3e (argmax %d) | |f32-f64| l2 %.3e max %.3e" %
              (np.linalg.norm(d_h), d_h.max(), int(d_h.argmax()), np.linalg.norm(d_32), d_32.max()))
    dh, d32 = np.abs(lh - l64), np.abs(l32 - l64)
    for a in range(0, max(ks), 256):
        print("steps %5d..%5d  max|hip-f64| %s   max|f32-f64| %s" % (a, a + 256, np.array2string(dh[a:a + 256].max(0), precision=2),
                                                                     np.array2string(d32[a:a + 256].max(0), precision=2)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "drift_report.npz"), **out)


if __name__ == "__main__":
    main()
