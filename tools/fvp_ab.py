"""Fisher-vector product of CPO (spo_cpo_fvp: cpo_actor_kernel<64, MODE_FVP> + its fixed-order reduction) at the headline
size, 20 back-to-back launches: us per launch and the fraction of the FP32 matrix peak (A/B and PMC target)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(N=4096, T=128, D=60, A=8):
    from safepo import _abi
    from safepo.common.model import ActorVCritic
    from safepo.single_agent.cpo import CPOEngine, default_cfg
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    pol = ActorVCritic(D, A).to(dev)
    eng = CPOEngine(pol, N, T, dict(default_cfg), dev)
    eng.buffer.data["obs"].copy_(torch.randn(N, T, D, device=dev))
    v = torch.randn(eng.Pa, device=dev)
    out = torch.empty_like(v)

    def launch():
        _abi.check(eng.lib.spo_cpo_fvp(_abi.ptr(eng.policy.theta), _abi.ptr(eng.buffer.data["obs"]), _abi.ptr(v), eng.M, D, A,
                                       _abi.ptr(eng.partial_ws), _abi.ptr(eng.loss_ws), _abi.ptr(out), _abi.stream_ptr()), "spo_cpo_fvp")
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    flops = 4 * 2.0 * (D * 64 + 64 * 64 + 64 * A) * N * T
    print(json.dumps({"fvp_us": round(us, 2), "frac_of_157.3TF": round(flops / (us * 1e-6) / 157.3e12, 4),
                      "out_norm": float(out.norm())}))


if __name__ == "__main__":
    main()
