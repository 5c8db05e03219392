"""Worker of test_ma_block_kernel_forms_agree: one MAPPO network forward + backward at a training-size batch with the kernel
forms the environment selects (SPO_MA_FWD_WAVE / SPO_MA_FUSE_HEAD are read once per process), results to an .npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


class _Sp:
    def __init__(self, n):
        self.shape = (n,)


def run(out_path, D, nb, O, actor, B):
    from safepo.common.model import MultiAgentActor, MultiAgentCritic
    from safepo.multi_agent.mappolag import default_cfg
    dev = torch.device("cuda:0")
    torch.manual_seed(D * 1000 + O)
    cfg = dict(default_cfg)
    cfg.update(device="cuda:0", hidden_size=128, layer_N=nb - 1)
    net = MultiAgentActor(cfg, _Sp(D), _Sp(O), dev) if actor else MultiAgentCritic(cfg, _Sp(D), dev)
    with torch.no_grad():
        net.theta.add_(0.1 * torch.randn_like(net.theta))
    x = torch.randn((B, D), device=dev) * 1.5 + 0.2
    dout = torch.randn((B, O), device=dev) / B
    out, saved = net.net_forward(x, keep=True)
    grad = torch.zeros_like(net.theta)
    net.net_backward(saved, dout, grad)
    torch.cuda.synchronize()
    np.savez(out_path, out=out.cpu().numpy(), grad=grad.cpu().numpy(), ws=saved[1].cpu().numpy())


if __name__ == "__main__":
    run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5])), int(sys.argv[6]))
