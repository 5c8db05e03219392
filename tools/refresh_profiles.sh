# GPU box: regenerate the round's measurements under gpurun_out/r02 (copied into profiles/r02 afterwards).
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $O
timeout 600 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof /tmp/pmcf /tmp/pmcw
# the SAME command under the profiler: kernel trace + stats, and the line that profiled run printed
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config3 > $O/bench_profiled_line.json 2> /tmp/prof.log
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf -- python $GRAFT_REPO_ROOT/tools/gae_modes.py > /tmp/pmcf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw -- python $GRAFT_REPO_ROOT/tools/gae_modes.py > /tmp/pmcw.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_summary.py /tmp/prof $O/kernel_stats_bench.csv $O/gae_dispatch_durations.json "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config3" | tail -25
F=$(find /tmp/pmcf -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmcw -name "*counter_collection.csv" | head -1)
echo "F=$F W=$W"
if [ -n "$F" ] && [ -n "$W" ]; then python tools/gae_pmc_summary.py "$F" "$W" $O/gae_pmc.json | tail -40; else tail -5 /tmp/pmcf.log /tmp/pmcw.log; fi
python tools/gae_modes.py > $O/gae_modes.log 2>&1; tail -6 $O/gae_modes.log
python tools/phase_profile_h.py > $O/update_phase_cycles_h.txt 2>&1; tail -22 $O/update_phase_cycles_h.txt
timeout 200 python tools/update_ab.py > $O/update_ab.txt 2>&1; tail -1 $O/update_ab.txt
timeout 200 python tools/ma_bench.py --episodes 3 2>/dev/null | tail -1 > $O/ma_bench_mappolag_config5.json; cut -c1-300 $O/ma_bench_mappolag_config5.json
cd /tmp && rm -rf /tmp/matrace && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/matrace -- python $GRAFT_REPO_ROOT/tools/ma_bench.py --episodes 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
( echo "# rocprofv3 --kernel-trace --output-format csv -- python tools/ma_bench.py --episodes 2, grouped by kernel and grid size (tools/kernel_trace_by_grid.py)"; python tools/kernel_trace_by_grid.py /tmp/matrace ) > $O/ma_kernel_trace_by_grid.txt; head -8 $O/ma_kernel_trace_by_grid.txt
