"""A/B timing of the GAE scan launch (development aid): python tools/gae_ab.py [num_envs]  (env knobs: SPO_GAE_PLAIN_STORES, SPO_GAE_RC_SPLIT)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "safe-policy-optimization_amd"))
from safepo.common.buffer import VectorizedOnPolicyBuffer
from safepo.common.engine import _Space
dev = torch.device("cuda:0")
N, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 128
b = VectorizedOnPolicyBuffer(_Space(1), _Space(1), size=T, num_envs=N, device=dev)
for k in ("reward", "cost", "value_r", "value_c"):
    b.data[k].normal_()
b.seg_end.zero_(); b.seg_end[:, 63] = 1; b.seg_end[:, 127] = 1
from safepo import _abi
if os.environ.get("SPO_GAE_ABLATE") or os.environ.get("SPO_GAE_VARIANT"):
    # ablate: 1 no stats, 2 no lane scan, 4 no stores; variant: 1 eager bootstrap loads, 2 predicated
    _abi.load().spo_debug_gae_variant(16 * int(os.environ.get("SPO_GAE_ABLATE", "0")) + int(os.environ.get("SPO_GAE_VARIANT", "0")))
b.compute_gae(None)
reps = 200 if N <= 8192 else 20
ts = sorted(b.time_scan(reps) * 1e6 for _ in range(7))
byts = 33.0 * N * T + 16.0 * N
print(N, os.environ.get("SPO_GAE_PLAIN_STORES", "0"), ["%.2f" % t for t in ts], "frac %.3f" % (byts / (ts[3] * 1e-6) / 8e12))
