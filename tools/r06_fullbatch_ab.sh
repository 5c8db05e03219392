# GPU box: bounded A/B on the full-batch kernels: MFMA results in VGPRs (-mllvm -amdgpu-mfma-vgpr-form),
# conflict-free weight rows (-DSPO_FULL_PAD=8), both -- KL (actor_full_kernel) and FVP (cpo_actor_kernel) timings per library,
# then the LDS conflict counters of the in-tree and the pad8 builds
cd $GRAFT_REPO_ROOT
V=safe-policy-optimization_amd/safepo/_lib/variants
echo "== KL kernel (tools/kl_ab.py)"; for r in 1 2; do timeout 300 python tools/kl_ab.py "" $V/libsafepo_hip_vgprform.so $V/libsafepo_hip_pad8.so $V/libsafepo_hip_pad8vgpr.so 2>&1 | grep -v "amdgpu.ids\|WARNING"; done
echo "== FVP kernel (tools/fvp_ab.py)"
for r in 1 2; do for l in "" vgprform pad8 pad8vgpr; do
  if [ -z "$l" ]; then echo -n "in-tree "; timeout 200 python tools/fvp_ab.py 2>&1 | grep "^{"; else echo -n "$l "; SPO_LIB_PATH=$GRAFT_REPO_ROOT/$V/libsafepo_hip_$l.so SPO_LIB_OVERRIDE=1 timeout 200 python tools/fvp_ab.py 2>&1 | grep "^{"; fi
done; done
echo "== LDS counters of the KL kernel: in-tree, then pad8"
cd /tmp && export TMPDIR=/tmp
for l in "" pad8; do
  rm -rf /tmp/klc
  if [ -z "$l" ]; then E=""; else E="SPO_LIB_PATH=$GRAFT_REPO_ROOT/$V/libsafepo_hip_$l.so SPO_LIB_OVERRIDE=1"; fi
  env $E timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/klc -- python $GRAFT_REPO_ROOT/tools/kl_ab.py --one > /tmp/klc.log 2>&1
  f=$(find /tmp/klc -name "*counter_collection.csv" | head -1)
  python - "$f" "${l:-in-tree}" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "actor_full_kernel" in r["Kernel_Name"] and ("ELi1E" in r["Kernel_Name"] or "<64, 1>" in r["Kernel_Name"]):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(sys.argv[2], {k: round(v) for k, v in m.items()}, "conflict / active =", round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3))
PY
done
