"""Development aid (GPU box): stage times of the row-group gradient kernel (csrc/mlp_rows.hip) from its instrumented build
    SPO_VARIANT_SOURCES=mlp_rows.hip python tools/build_variant.py mrprof -DSPO_MR_PROF      (CPU container)
    SPO_LIB_PATH=.../variants/libsafepo_hip_mrprof.so SPO_LIB_OVERRIDE=1 python tools/phase_profile_rows.py [hidden ...]
Wall-clock stamps (100 MHz: 10 ns) of the first lane of the first and of the last workgroup of the LAST launch."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "safe-policy-optimization_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import wide_bench
    hidden = [int(v) for v in sys.argv[1:]] or [128, 128]
    res = wide_bench.one(hidden, 64, 64)
    from safepo import _abi
    lib = _abi.load()
    fn = lib.spo_debug_mr_profile
    buf = (ctypes.c_ulonglong * 64)()
    assert fn(buf) == 0
    n = len(hidden) + 1
    names = ["row indices (cursor -> idx)", "observations + loss inputs -> LDS"] + [f"forward layer {l}" for l in range(n)] + ["loss"] + \
            [f"backward layer {l}" for l in range(n - 1, -1, -1)]
    out = {"us_per_minibatch_step": res["us_per_minibatch_step"]}
    for w, tag in ((0, "first workgroup"), (1, "last workgroup")):
        st = [buf[32 * w + k] for k in range(len(names) + 1)]
        out[tag] = {nm: round((st[k + 1] - st[k]) * 0.01, 2) for k, nm in enumerate(names)}
        out[tag]["total_us"] = round((st[len(names)] - st[0]) * 0.01, 2)
        fine = [buf[32 * w + k] for k in range(32)]
        out[tag]["forward layers, wave 0: [MFMAs issued, stores done, next layer's registers copied] us after the layer's start"] = [
            [round((fine[16 + 4 * l + i] - st[2 + l]) * 0.01, 2) for i in range(3)] for l in range(min(n, 4))]
    for w, tag in ((0, "first workgroup"), (1, "last workgroup")):
        fine = [buf[32 * w + k] for k in range(32)]
        if fine[26] and n >= 2:
            t0 = buf[32 * w + 2 + n + 1 + (n - 1 - 1)]          # start of backward layer 1 = end of the stage before it
            out[tag]["backward layer 1, wave 0: [bias sums done, weight-gradient tiles issued, dZ below stored, registers copied] us after the stage's start"] = [
                round((fine[k] - t0) * 0.01, 2) for k in (26, 27, 28, 29)]
    print(json.dumps(out, indent=1))


main()
