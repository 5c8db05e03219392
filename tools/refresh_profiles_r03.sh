# GPU box: regenerate the round's measurements under gpurun_out/r03 (copied into profiles/r03 afterwards).
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof /tmp/profc /tmp/pmcf /tmp/pmcw
# the bench command under the profiler: kernel trace + stats, and the line that profiled run printed
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-config5 > $O/bench_profiled_line.json 2> /tmp/prof.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -- python $GRAFT_REPO_ROOT/bench.py --algo cpo --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cpo_profiled_line.json 2> /tmp/profc.log
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf -- python $GRAFT_REPO_ROOT/tools/gae_modes.py > /tmp/pmcf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw -- python $GRAFT_REPO_ROOT/tools/gae_modes.py > /tmp/pmcw.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_summary.py /tmp/prof $O/kernel_stats_bench.csv $O/gae_dispatch_durations.json "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-config5" | tail -12
python tools/kernel_trace_summary.py /tmp/profc $O/kernel_stats_bench_cpo.csv $O/gae_dispatch_durations_cpo.json "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --algo cpo --steps 2 --warmup 1 --no-cpu-baseline" | tail -3
F=$(find /tmp/pmcf -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmcw -name "*counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/gae_pmc_summary.py "$F" "$W" $O/gae_pmc.json | tail -12; else tail -5 /tmp/pmcf.log /tmp/pmcw.log; fi
python tools/phase_profile_h.py > $O/update_phase_cycles_h.txt 2>&1; tail -8 $O/update_phase_cycles_h.txt
timeout 200 python tools/update_ab.py > $O/update_ab.txt 2>&1; tail -1 $O/update_ab.txt
timeout 100 python tools/kl_ab.py > $O/kl_ab.txt 2>&1; tail -1 $O/kl_ab.txt
timeout 300 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep '^{' > $O/p2p_loopback.txt; cat $O/p2p_loopback.txt
SPO_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_dp2_one_gpu.json 2> $O/bench_dp2_one_gpu.err; tail -c 300 $O/bench_dp2_one_gpu.json
SPO_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --dp-batch global --learning-iters 4 > $O/bench_dp2_one_gpu_global_batch.json 2> $O/bench_dp2_global.err; tail -c 300 $O/bench_dp2_one_gpu_global_batch.json
timeout 900 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
