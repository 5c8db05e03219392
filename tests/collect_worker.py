"""Worker of tests/test_gpu_parity.py::test_side_by_side_collect_kernels_equal_the_sequential_ones (not a test module).

Runs two epochs of the collect loop (engine.rollout_epoch on the synthetic device env, fused observation normaliser, an update
in between) from fixed seeds and writes what they leave behind to an .npz -- the test runs it once per kernel selection
(SPO_STEP_PAR = 1 / 0 is read once per process) and compares the files bit for bit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(out_path, n, t, d, a):
    from safepo.common.engine import PPOLagEngine
    from safepo.common.env import SynthDeviceEnv
    from safepo.common.model import ActorVCritic
    dev = torch.device("cuda:0")
    N, T, D, A = int(n), int(t), int(d), int(a)
    torch.manual_seed(3)
    pol = ActorVCritic(D, A).to(dev)
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": float("inf"), "batch_size": 64, "learning_iters": 1, "max_grad_norm": 40.0}
    eng = PPOLagEngine(pol, N, T, cfg, dev)
    env = SynthDeviceEnv(N, D, A, seed=9, p_term=0.01, p_cost=0.1, trunc_len=11, device=dev, normalize_obs=True, obs_scale=1.5, obs_shift=-0.25)
    rms = env.fuse_normalize(True)
    obs, _ = env.reset()
    out = {}
    for e in range(2):
        obs = eng.rollout_epoch(env, obs, rms=rms)
        n_ep = eng.drain_episode_events(None)
        b = eng.buffer
        for k, v in b.data.items():
            out[f"e{e}_{k}"] = v.cpu().numpy()
        out[f"e{e}_seg_end"] = b.seg_end.cpu().numpy()
        out[f"e{e}_boot_r"], out[f"e{e}_boot_c"] = b.boot_r.cpu().numpy(), b.boot_c.cpu().numpy()
        out[f"e{e}_events"] = eng.events[:n_ep].cpu().numpy()
        out[f"e{e}_rms"] = rms.state.cpu().numpy()
        out[f"e{e}_vfinal"] = torch.stack([eng.vfinal_r, eng.vfinal_c]).cpu().numpy()
        eng.update(0.001)
        out[f"e{e}_theta"] = pol.theta.cpu().numpy()
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(*sys.argv[1:6])
