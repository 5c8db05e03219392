# GPU box: round-3 check-in (tests, short bench with the new roofline entries, rocprofv3 kernel stats of the CPO bench)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.log
tail -3 $O/pytest.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/profc
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -- python $GRAFT_REPO_ROOT/bench.py --algo cpo --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cpo_profiled_line.json 2> /tmp/profc.log
cd $GRAFT_REPO_ROOT
python tools/kernel_trace_summary.py /tmp/profc $O/kernel_stats_bench_cpo.csv $O/gae_dispatch_durations_cpo.json "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --algo cpo --steps 2 --warmup 1 --no-cpu-baseline" | tail -5
head -12 $O/kernel_stats_bench_cpo.csv
