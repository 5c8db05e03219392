"""Micro-benchmark of the GAE scan kernel: variants x sizes, HIP-event timing over R launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))
from safepo import _abi
from safepo.common.buffer import VectorizedOnPolicyBuffer
from safepo.common.engine import _Space
dev = torch.device("cuda:0")
lib = _abi.load()
T = 128
for N in (4096, 32768, 262144):
    b = VectorizedOnPolicyBuffer(_Space(1), _Space(1), size=T, num_envs=N, device=dev)
    for k in ("reward", "cost", "value_r", "value_c"):
        b.data[k].normal_()
    b.seg_end[:, T - 1] = 1; b.seg_end[:, T // 2 - 1] = 1
    b.boot_r.normal_(); b.boot_c.normal_()
    d = b.data
    def launch():
        lib.spo_gae_fused(_abi.ptr(d["reward"]), _abi.ptr(d["cost"]), _abi.ptr(d["value_r"]), _abi.ptr(d["value_c"]),
                          _abi.ptr(b.seg_end), _abi.ptr(b.boot_r), _abi.ptr(b.boot_c), _abi.ptr(d["adv_r"]), _abi.ptr(d["adv_c"]),
                          _abi.ptr(d["target_value_r"]), _abi.ptr(d["target_value_c"]), _abi.ptr(b._partials), N, T, 0.99, 0.95, 0.95,
                          _abi.stream_ptr())
    for var, name in ((1, "eager"), (2, "predicated"), (2 + 16, "pred-noreduce"), (2 + 32, "pred-noscan"), (2 + 64, "pred-nostore"), (2 + 16 + 32 + 64, "pred-loadonly")):
        lib.spo_debug_gae_variant(var)
        for _ in range(5): launch()
        torch.cuda.synchronize()
        R = 50
        # HIP graph of R launches: GPU-side back-to-back, no host launch cost in the measurement
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            launch(); torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(R): launch()
            graph.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / R
        by = 33.0 * N * T + 8.0 * 2 * N
        print(f"N={N:7d} {name:10s} {us:8.2f} us/launch  {by/us/1e3:8.1f} GB/s algorithmic  ({by/us/1e3/8000*100:.1f}% of 8 TB/s)")
    lib.spo_debug_gae_variant(0)
    del b
