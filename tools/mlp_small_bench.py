"""Development aid (GPU box): per-launch time of spo_mlp_forward / spo_mlp_backward at small row counts (csrc/mlp_small.hip), each
replayed 50 times from one HIP graph.   python tools/mlp_small_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "safe-policy-optimization_amd"))
from safepo import _abi  # noqa: E402

lib = _abi.load()
dev = torch.device("cuda:0")
REPS = 50


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return round(best, 2)


out = []
for dims, rows in (([60, 128, 128, 8], 64), ([60, 128, 128, 8], 16), ([60, 16, 1], 64), ([60, 16, 1], 16), ([376, 64, 64, 17], 64),
                   ([60, 64, 64, 8], 64), ([60, 128, 128, 8], 128)):
    net = _abi.MlpNet.of(dims)
    P = sum(dims[l + 1] * dims[l] + dims[l + 1] for l in range(len(dims) - 1))
    theta = torch.randn(P, device=dev) * 0.1
    x = torch.randn(rows, dims[0], device=dev)
    ws = torch.zeros(int(lib.spo_mlp_workspace_floats(net, rows)), device=dev)
    grad = torch.zeros_like(theta)
    scratch = torch.zeros(int(lib.spo_mlp_backward_scratch_floats(net, rows)), device=dev)
    dout = torch.randn(rows, dims[-1], device=dev)
    f = timed(lambda: _abi.check(lib.spo_mlp_forward(_abi.ptr(theta), net, _abi.ptr(x), rows, _abi.ptr(ws), _abi.stream_ptr()), "f"))
    b = timed(lambda: _abi.check(lib.spo_mlp_backward(_abi.ptr(theta), net, _abi.ptr(x), rows, _abi.ptr(ws), _abi.ptr(dout), _abi.ptr(grad),
                                                      _abi.ptr(scratch), _abi.stream_ptr()), "b"))
    out.append({"dims": dims, "rows": rows, "forward_us": f, "backward_us": b})
    print(json.dumps(out[-1]), flush=True)
