# GPU box: PMC passes over the full-batch actor kernels (tools/kl_ab.py); one small counter group per pass
set -x
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${SPO_ROUND:-r05}/kl_pmc
mkdir -p $O
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf /tmp/kpmc$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/kpmc$i -- python $GRAFT_REPO_ROOT/tools/kl_ab.py > /tmp/kpmc$i.log 2>&1
  f=$(find /tmp/kpmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|actor_full_kernel" "$f" > $O/pass$i.csv; else echo "pass $i ($grp): no counter file"; tail -3 /tmp/kpmc$i.log; fi
done
python - <<'PY'
import csv, glob, os, collections, json
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ.get("SPO_ROUND", "r05"), "kl_pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O + "/pass*.csv")):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        name = "actor_full_kernel<64,1> (KL)" if "ELi1E" in k or "<64, 1>" in k else "actor_full_kernel other mode"
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for k, d in out.items():
    c = {n: v["mean_per_launch"] for n, v in d.items()}
    kc = c["GRBM_GUI_ACTIVE"] / 8.0
    print(k, "cycles", round(kc), "mfma_busy", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * kc), 3), "valu/simd", round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / 1024),
          "issue", round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3), "wait_inst", round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
          "wait_any", round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3), "lds_conf", round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), 3),
          "mfma", round(c["SQ_INSTS_MFMA"]), "lds_insts", round(c["SQ_INSTS_LDS"]))
PY
