# GPU box: regenerate the round's measurements under gpurun_out/r06 (copied into profiles/r06 afterwards by tools/copy_profiles_r06.sh).
#   gpurun --timeout 2700 -- 'bash tools/refresh_profiles_r06.sh'
set -x
cd $GRAFT_REPO_ROOT
export SPO_ROUND=r06
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof /tmp/profc /tmp/pmcf /tmp/pmcw /tmp/profk
CMD_PPO="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-config5 --no-wide"
CMD_CPO="python bench.py --algo cpo --steps 2 --warmup 1 --no-cpu-baseline"
CMD_KS="python tools/ks_bench.py 376,17"
( cd $GRAFT_REPO_ROOT && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- $CMD_PPO > $O/bench_profiled_line.json 2> /tmp/prof.log )
( cd $GRAFT_REPO_ROOT && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -- $CMD_CPO > $O/bench_cpo_profiled_line.json 2> /tmp/profc.log )
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profk -- $CMD_KS > $O/feature_split_profiled_line.json 2> /tmp/profk.log )
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf -- python $GRAFT_REPO_ROOT/tools/gae_modes.py > /tmp/pmcf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw -- python $GRAFT_REPO_ROOT/tools/gae_modes.py > /tmp/pmcw.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_summary.py /tmp/prof $O/kernel_stats_bench.csv $O/gae_dispatch_durations.json "rocprofv3 --kernel-trace --stats --output-format csv -- $CMD_PPO" | tail -14
python tools/kernel_trace_summary.py /tmp/profc $O/kernel_stats_bench_cpo.csv $O/gae_dispatch_durations_cpo.json "rocprofv3 --kernel-trace --stats --output-format csv -- $CMD_CPO" | tail -3
python tools/kernel_trace_summary.py /tmp/profk $O/kernel_stats_feature_split.csv /tmp/ks_gae.json "rocprofv3 --kernel-trace --stats --output-format csv -- $CMD_KS" | tail -5
F=$(find /tmp/pmcf -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmcw -name "*counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/gae_pmc_summary.py "$F" "$W" $O/gae_pmc.json | tail -12; else tail -5 /tmp/pmcf.log /tmp/pmcw.log; fi
python tools/hbm_copy_ceiling.py > $O/hbm_copy_ceiling.json 2>/dev/null; cat $O/hbm_copy_ceiling.json
python tools/collect_kernels_bench.py > $O/collect_kernels.txt 2>&1; tail -3 $O/collect_kernels.txt
{ echo "== row-split kernel (default, SPO_UPDATE_FORM=3), then the main + helper kernel (2), the four-wave kernel (0), write-through exchange stores (SPO_RS_SAFE=1), 16-row workgroups (SPO_RS_ROWS=16)"
  timeout 200 python tools/update_ab.py 2>&1 | tail -1
  SPO_UPDATE_FORM=2 timeout 200 python tools/update_ab.py 2>&1 | tail -1
  SPO_UPDATE_FORM=0 timeout 200 python tools/update_ab.py 2>&1 | tail -1
  SPO_RS_SAFE=1 timeout 200 python tools/update_ab.py 2>&1 | tail -1
  SPO_RS_ROWS=16 timeout 200 python tools/update_ab.py 2>&1 | tail -1; } > $O/update_ab_rs.txt 2>&1; cat $O/update_ab_rs.txt
timeout 100 python tools/kl_ab.py > $O/kl_ab.txt 2>&1; tail -1 $O/kl_ab.txt
{ for w in 5 0 2; do echo "== workgroup $w (0, 1: reward critic; 2, 3: cost critic; 4, 5: actor)"; SPO_RS_PROF_WG=$w timeout 200 python tools/phase_profile_rs.py 2>&1 | grep -v amdgpu; done; } > $O/update_phase_cycles_rs.txt 2>&1; tail -30 $O/update_phase_cycles_rs.txt
SPO_UPDATE_FORM=2 python tools/phase_profile_h.py > $O/update_phase_cycles_h.txt 2>&1; tail -4 $O/update_phase_cycles_h.txt
{ timeout 120 python tools/ks_bench.py 60,20 130,8 376,17 512,32
  echo "-- FOCOPS step / CUP second stage (KL-penalty loss; actor alone)"
  for m in focops cup2; do KS_BENCH_MODE=$m timeout 120 python tools/ks_bench.py 376,17; done
  echo "-- critic fit of the second-order scripts (128-row minibatches)"
  timeout 120 python tools/ks_cfit_bench.py 376,17 200,8
} 2>&1 | grep -v "amdgpu.ids\|WARNING" > $O/feature_split_bench.txt; cat $O/feature_split_bench.txt
bash tools/r06_loopback.sh > /dev/null 2>&1; cat $O/p2p_loopback.txt
HSA_ENABLE_IPC_MODE_LEGACY=0 SPO_BENCH_ONE_GPU=1 timeout 500 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --config5-threads 1024 > $O/bench_dp2_one_gpu.json 2> $O/bench_dp2_one_gpu.err; tail -c 300 $O/bench_dp2_one_gpu.json
# the wide path's row-group gradient kernel (item 6) and the feature-split gradient launch (item 4b); tests left to tools/r06_final.sh
sed -e '/pytest/d' tools/r06_rows.sh > /tmp/r06_rows_notests.sh; bash /tmp/r06_rows_notests.sh > /dev/null 2>&1; cat $O/wide_step.txt | cut -c1-200
{ timeout 300 python tools/ks_grad_bench.py 376,17 130,8; echo "-- SPO_WIDE_ROWS=0"; SPO_WIDE_ROWS=0 timeout 300 python tools/ks_grad_bench.py 376,17 130,8; } 2>&1 | grep -v "amdgpu.ids\|WARNING" > $O/dp_feature_split_step.txt; cat $O/dp_feature_split_step.txt
bash tools/update_pmc.sh > $O/update_pmc.log 2>&1; cp $O/update_pmc/summary.json $O/update_kernel_pmc.json; tail -5 $O/update_pmc.log
bash tools/kl_pmc.sh > $O/kl_pmc.log 2>&1; cp $O/kl_pmc/summary.json $O/kl_kernel_pmc.json; tail -2 $O/kl_pmc.log
bash tools/fvp_pmc.sh > $O/fvp_pmc.log 2>&1; cp $O/fvp_pmc/summary.json $O/fvp_kernel_pmc.json; tail -1 $O/fvp_pmc.log
bash tools/ma_pmc.sh > $O/ma_pmc.log 2>&1; cp $O/ma_pmc/summary.json $O/ma_train_kernels_pmc.json
ls -la $O
