"""Cost of the in-kernel gradient exchange without a second GPU: R "ranks" driven from ONE process on R streams of one
GPU (regions are plain device pointers here, no IPC), so all R persistent kernels are co-resident and really
exchange through uncached device memory.  Reports us per minibatch step next to the single-rank persistent kernel.
xGMI latency is NOT in this number (add ~1-2 us per flag round trip on a real node)."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-policy-optimization_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(world=2, M=8192, iters=3, D=60, A=8):
    from safepo import _abi
    from safepo.common.engine import PPOLagEngine
    from safepo.common.model import ActorVCritic
    lib = _abi.load()
    dev = torch.device("cuda:0")
    cfg = {"hidden_sizes": [64, 64], "gamma": 0.99, "target_kl": 1e9, "batch_size": 64, "learning_iters": 1,
           "max_grad_norm": 40.0}
    regions = (ctypes.c_void_p * 8)()
    for r in range(world):
        own, h = ctypes.c_void_p(), (ctypes.c_ubyte * 64)()
        _abi.check(lib.spo_p2p_alloc(ctypes.byref(own), h), "alloc")
        regions[r] = own
    # protocol alone: the self-test kernel runs xr_allreduce back to back with no compute in between
    res2 = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(world)]
    sstreams = [torch.cuda.Stream() for _ in range(world)]
    xiters, xstep = 2000, 0
    us_proto = None
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(world):
            with torch.cuda.stream(sstreams[r]):
                _abi.check(lib.spo_p2p_selftest(r, world, regions, xstep, xiters, _abi.ptr(res2[r]), _abi.stream_ptr()), "st")
        torch.cuda.synchronize()
        us_proto = 1e6 * (time.perf_counter() - t0) / xiters
        xstep += xiters
    proto_bad = [t.tolist() for t in res2]
    if os.environ.get("SPO_LOOPBACK_DEBUG"):
        print("selftest results {bad, timeout} per rank:", proto_bad, "us per exchange", round(us_proto, 2), flush=True)
    prof = (ctypes.c_ulonglong * 8)()
    _abi.check(lib.spo_debug_xr_profile(prof, 1), "prof")
    n = max(prof[6], 1)
    print("xr profile per exchange (cycles): stores+misc %.0f | reduce polls %.0f | before final %.0f | final polls %.0f | "
          "rounds reduce %.2f final %.2f" % (prof[0] / n, prof[1] / n, prof[2] / n, prof[3] / n, prof[4] / n, prof[5] / n))
    engines, streams, perms = [], [], []
    for r in range(world):
        torch.manual_seed(7)
        pol = ActorVCritic(D, A).to(dev)
        eng = PPOLagEngine(pol, 1, M, cfg, dev)
        g = torch.Generator().manual_seed(100 + r)
        b = eng.buffer
        b.data["obs"].copy_(torch.randn(1, M, D, generator=g)); b.data["act"].copy_(torch.randn(1, M, A, generator=g))
        b.data["log_prob"].copy_(-8 + 0.1 * torch.randn(1, M, generator=g))
        b.data["target_value_r"].copy_(torch.randn(1, M, generator=g))
        b.data["target_value_c"].copy_(torch.rand(1, M, generator=g)); b.adv_mix.copy_(torch.randn(1, M, generator=g))
        engines.append(eng); streams.append(torch.cuda.Stream())
        perms.append(torch.randperm(M, generator=g).to(torch.int32).to(dev))
    n_mb = (M + 63) // 64
    losses = [torch.empty((n_mb, 3), device=dev) for _ in range(world)]

    def launch_all(step0):
        for r, eng in enumerate(engines):
            d, b = eng.buffer.data, eng.buffer
            with torch.cuda.stream(streams[r]):
                _abi.check(lib.spo_ppo_lag_update_iter_dp(
                    _abi.ptr(eng.policy.theta), _abi.ptr(eng.adam_m), _abi.ptr(eng.adam_v), eng.adam_step,
                    _abi.ptr(d["obs"]), _abi.ptr(d["act"]), _abi.ptr(d["log_prob"]), _abi.ptr(d["target_value_r"]),
                    _abi.ptr(d["target_value_c"]), _abi.ptr(b.adv_mix), _abi.ptr(perms[r]), M, eng._cfg_struct(),
                    _abi.ptr(losses[r]), _abi.ptr(eng.sync_ws), r, world, regions, step0 & 0xFFFFFFFF,
                    _abi.stream_ptr()), "dp")
            eng.adam_step += n_mb
    # every stream's scratch blocks (row-split kernel: exchange slots, granules) are allocated at the stream's first launch, with a
    # device synchronisation: do that now, one rank after the other, not between the ranks' first data-parallel launches
    for r, eng in enumerate(engines):
        with torch.cuda.stream(streams[r]):
            th0, m0, v0, st0 = eng.policy.theta.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.adam_step
            eng.learning_iter(perms[r][:128].contiguous()) if False else None
            eng.M, keep = 128, eng.M
            eng.learning_iter(perms[r][:128].contiguous() % 128)
            eng.M = keep
            eng.policy.theta.copy_(th0); eng.adam_m.copy_(m0); eng.adam_v.copy_(v0); eng.adam_step = st0
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    step0 = xstep
    launch_all(step0); step0 += n_mb
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        launch_all(step0); step0 += n_mb
    torch.cuda.synchronize()
    us_dp = 1e6 * (time.perf_counter() - t0) / (iters * n_mb)
    if os.environ.get("SPO_LOOPBACK_DEBUG"):
        print("us per step", round(us_dp, 2), "error words", [int(e.sync_ws[8].item()) & 0xFFFFFFFF for e in engines], flush=True)
    for eng in engines:
        eng.check_sync_error()
    same = all(torch.equal(engines[0].policy.theta, e.policy.theta) for e in engines[1:])
    # single-rank persistent kernel on the same shapes
    eng = engines[0]
    eng.learning_iter(perms[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.learning_iter(perms[0])
    torch.cuda.synchronize()
    us_1 = 1e6 * (time.perf_counter() - t0) / (iters * n_mb)
    print(json.dumps({"world": world, "M": M, "us_per_step_exchange": round(us_dp, 2),
                      "us_per_step_single": round(us_1, 2), "us_protocol_only": round(us_proto, 2), "selftest": proto_bad, "replicas_identical": same}))
    for r in range(world):
        lib.spo_p2p_free(regions[r])


if __name__ == "__main__":
    worlds = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
    if len(worlds) == 1 and os.environ.get("GPU_MAX_HW_QUEUES"):
        main(world=worlds[0], M=int(os.environ.get("SPO_LOOPBACK_M", "8192")))
    else:
        # one process per world size: the W streams of a run must land on W different hardware queues, and streams left over
        # from an earlier world size in the same process shift that assignment (GPU_MAX_HW_QUEUES=24 gives 8 ranks room)
        import subprocess
        env = dict(os.environ)
        env.setdefault("GPU_MAX_HW_QUEUES", "24")
        for w in worlds:
            subprocess.run([sys.executable, os.path.abspath(__file__), str(w)], env=env)
