"""Development aid: build a variant of libsafepo_hip.so with extra -D flags for A/B runs on the GPU box.
    python tools/build_variant.py NAME -DSPO_FOO=1 ...   ->  safe-policy-optimization_amd/safepo/_lib/variants/libsafepo_hip_NAME.so
Select it with SPO_LIB_PATH=<that path> (safepo/_abi.py).  Only update.hip is recompiled; the other objects are reused."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

name, flags = sys.argv[1], sys.argv[2:]
G.build()
vdir = os.path.join(G.LIBDIR, "variants")
os.makedirs(vdir, exist_ok=True)
srcs = [s for s in os.environ.get("SPO_VARIANT_SOURCES", "update.hip").split(",") if s]
objs = []
for src in G.HIP_SOURCES:
    if src in srcs:
        obj = os.path.join(vdir, f"{src[:-4]}_{name}.o")
        cmd = [G._hipcc()] + G.HIPCC_FLAGS + G.EXTRA_FLAGS.get(src, []) + flags + ["-c", os.path.join(G.CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        objs.append(obj)
    else:
        objs.append(os.path.join(G.OBJDIR, src.replace(".hip", ".o")))
out = os.path.join(vdir, f"libsafepo_hip_{name}.so")
r = subprocess.run([G._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", out], capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr)
print(out)
