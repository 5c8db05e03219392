set -x
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err
tail -c 300 gpurun_out/g_bench.json
timeout 200 python bench.py --algo cpo --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/g_bench_cpo.json 2>/dev/null
tail -c 400 gpurun_out/g_bench_cpo.json
timeout 150 python tools/ma_bench.py --episodes 3 2>/dev/null | tail -1 > gpurun_out/g_ma_bench.json
cat gpurun_out/g_ma_bench.json
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof /tmp/pmcf /tmp/pmcw
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/prof.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --learning-iters 1 > /tmp/pmcf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --learning-iters 1 > /tmp/pmcw.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_summary.py /tmp/prof gpurun_out/g_kernel_stats.csv gpurun_out/g_gae.json "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline" | tail -25
F=$(find /tmp/pmcf -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmcw -name "*counter_collection.csv" | head -1)
echo "F=$F W=$W"
if [ -n "$F" ] && [ -n "$W" ]; then python tools/pmc_summary.py "$F" "$W" gpurun_out/g_gae_pmc.json | tail -30; else tail -5 /tmp/pmcf.log /tmp/pmcw.log; fi
