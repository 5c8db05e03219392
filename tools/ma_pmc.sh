# GPU box: PMC passes over the MAPPO-L training kernels (tools/ma_bench.py --episodes 1); one small counter group per pass
# (separate runs, --kernel-trace only); per-kernel means -> gpurun_out/${SPO_ROUND:-r05}/ma_pmc/summary.json
set -x
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${SPO_ROUND:-r05}/ma_pmc
mkdir -p $O
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf /tmp/mpmc$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/mpmc$i -- python $GRAFT_REPO_ROOT/tools/ma_bench.py --episodes 1 > /tmp/mpmc$i.log 2>&1
  f=$(find /tmp/mpmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|fused_block_fwd128w|fused_dx_lnbwd128w|dw_partial_kernel|ma_collect_kernel" "$f" > $O/pass$i.csv; else echo "pass $i ($grp): no counter file"; tail -3 /tmp/mpmc$i.log; fi
done
python - <<'PY'
import csv, glob, os, collections, json
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ.get("SPO_ROUND", "r05"), "ma_pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O + "/pass*.csv")):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        name = next((n for n in ("fused_block_fwd128w", "fused_dx_lnbwd128w", "dw_partial_kernel<true>", "dw_partial_kernel<false>", "ma_collect_kernel") if n in k), None)
        if name and r.get("Grid_Size") not in (None, ""):
            agg[name + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: round(v["mean_per_launch"]) for c, v in d.items()})
PY
