# GPU box: PMC passes over the persistent update launch (UPD_KERNEL: kernel-name substring, default the row-split kernel of round 6;
# PMC_CMD: the profiled command, default tools/update_ab.py --one = 4 launches of 8192 minibatch steps; PMC_OUT: output directory name).
# One small counter group per pass (separate runs, --kernel-trace only); results -> gpurun_out/${SPO_ROUND:-r05}/${PMC_OUT:-update_pmc}/*.csv
set -x
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${SPO_ROUND:-r05}/${PMC_OUT:-update_pmc}
mkdir -p $O
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rm -rf /tmp/upmc$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/upmc$i -- ${PMC_CMD:-python $GRAFT_REPO_ROOT/tools/update_ab.py --one} > /tmp/upmc$i.log 2>&1
  f=$(find /tmp/upmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|${UPD_KERNEL:-ppo_update_rs_kernel}" "$f" | head -400 > $O/pass$i.csv; else echo "pass $i ($grp): no counter file"; tail -3 /tmp/upmc$i.log; fi
done
python - <<'PY'
import csv, glob, os, collections, json
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ.get("SPO_ROUND", "r05"), os.environ.get("PMC_OUT", "update_pmc"))
agg = collections.defaultdict(list)
for f in sorted(glob.glob(O + "/pass*.csv")):
    for r in csv.DictReader(open(f)):
        if os.environ.get("UPD_KERNEL", "ppo_update_rs_kernel") in r.get("Kernel_Name", ""):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in agg.items()}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
