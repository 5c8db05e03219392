"""A/B of the GAE scan layouts on the GPU box: {bootstrap arrays, folded} at the three roofline
sizes.  Prints per-dispatch (HIP events around each launch) and graph-replay averages and the HBM fraction."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))
import numpy as np
import torch

from safepo import _abi
from safepo.common.buffer import VectorizedOnPolicyBuffer
from safepo.common.engine import _Space

dev = torch.device("cuda:0")
lib = _abi.load()
T = 128
out = []
for N in (4096, 32768, 262144):
    buf = VectorizedOnPolicyBuffer(_Space(1), _Space(1), size=T, num_envs=N, device=dev)
    for k in ("reward", "cost", "value_r", "value_c"):
        buf.data[k].normal_()
    buf.seg_end[:, T - 1] = 1
    buf.seg_end[:, T // 2 - 1] = 1
    buf.boot_r.normal_(); buf.boot_c.normal_()
    seg = buf.seg_end.bool()
    g32 = torch.tensor(0.99, dtype=torch.float32, device=dev)
    buf.reward_fold.copy_(torch.where(seg, buf.data["reward"] + g32 * buf.boot_r, buf.data["reward"]))
    buf.cost_fold.copy_(torch.where(seg, buf.data["cost"] + g32 * buf.boot_c, buf.data["cost"]))
    nseg = int(seg.sum().item())
    for vec8 in (0,):
        for folded in (False, True):
            buf.ptr = T
            buf._fold_cols = T if folded else 0
            buf.compute_gae(None)
            assert buf.last_scan_folded == folded
            reps = 100 if N <= 4096 else 20 if N <= 32768 else 10
            disp = np.asarray(buf.time_scan_dispatches(reps))
            graph = buf.time_scan(reps)
            bytes_ = 33.0 * N * T + (0.0 if folded else 8.0 * nseg)
            rec = {"N": N, "vec": 8 if vec8 else 4, "folded": folded, "blocks": lib.spo_gae_num_blocks(N, T),
                   "dispatch_avg_us": round(float(disp.mean()) * 1e6, 3), "dispatch_med_us": round(float(np.median(disp)) * 1e6, 3),
                   "graph_avg_us": round(graph * 1e6, 3), "frac_dispatch": round(bytes_ / disp.mean() / 8e12, 4),
                   "frac_graph": round(bytes_ / graph / 8e12, 4)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    del buf
