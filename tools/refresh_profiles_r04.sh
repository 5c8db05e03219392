# GPU box: regenerate the round's measurements under gpurun_out/r04 (copied into profiles/r04 afterwards).
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles_r04.sh'
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof /tmp/profc /tmp/pmcf /tmp/pmcw
CMD_PPO="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-config5 --no-wide"
CMD_CPO="python bench.py --algo cpo --steps 2 --warmup 1 --no-cpu-baseline"
# the bench command under the profiler: kernel trace + stats, and the line that profiled run printed
( cd $GRAFT_REPO_ROOT && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- $CMD_PPO > $O/bench_profiled_line.json 2> /tmp/prof.log )
( cd $GRAFT_REPO_ROOT && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -- $CMD_CPO > $O/bench_cpo_profiled_line.json 2> /tmp/profc.log )
# HBM traffic of the GAE scan: FETCH_SIZE and WRITE_SIZE in separate passes (counters only: no trace domains beside --kernel-trace)
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf -- python $GRAFT_REPO_ROOT/tools/gae_modes.py > /tmp/pmcf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw -- python $GRAFT_REPO_ROOT/tools/gae_modes.py > /tmp/pmcw.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_summary.py /tmp/prof $O/kernel_stats_bench.csv $O/gae_dispatch_durations.json "rocprofv3 --kernel-trace --stats --output-format csv -- $CMD_PPO" | tail -14
python tools/kernel_trace_summary.py /tmp/profc $O/kernel_stats_bench_cpo.csv $O/gae_dispatch_durations_cpo.json "rocprofv3 --kernel-trace --stats --output-format csv -- $CMD_CPO" | tail -3
F=$(find /tmp/pmcf -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmcw -name "*counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/gae_pmc_summary.py "$F" "$W" $O/gae_pmc.json | tail -12; else tail -5 /tmp/pmcf.log /tmp/pmcw.log; fi
# per-kernel numbers of the round
python tools/collect_kernels_bench.py > $O/collect_kernels.txt 2>&1; SPO_STEP_PAR=0 SPO_OBS_STATS_REG=0 python tools/collect_kernels_bench.py >> $O/collect_kernels.txt 2>&1; cat $O/collect_kernels.txt
timeout 200 python tools/update_ab.py > $O/update_ab.txt 2>&1; tail -1 $O/update_ab.txt
timeout 100 python tools/kl_ab.py > $O/kl_ab.txt 2>&1; tail -1 $O/kl_ab.txt
python tools/phase_profile_h.py > $O/update_phase_cycles_h.txt 2>&1; tail -4 $O/update_phase_cycles_h.txt
( timeout 300 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep '^{' | sed 's/^/default   /'; SPO_P2P_ALGO=doubling timeout 300 python tools/p2p_loopback_bench.py 8 2>&1 | grep '^{' | sed 's/^/doubling  /'; SPO_P2P_ALGO=twophase timeout 300 python tools/p2p_loopback_bench.py 2 4 2>&1 | grep '^{' | sed 's/^/twophase  /' ) > $O/p2p_loopback_final.txt; cat $O/p2p_loopback_final.txt
SPO_BENCH_ONE_GPU=1 timeout 500 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --config5-threads 1024 > $O/bench_dp2_one_gpu.json 2> $O/bench_dp2_one_gpu.err; tail -c 300 $O/bench_dp2_one_gpu.json
