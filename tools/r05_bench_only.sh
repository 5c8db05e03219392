cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
date +%s > $O/bench_t0
timeout 700 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
echo "bench wall: $(( $(date +%s) - $(cat $O/bench_t0) )) s"
tail -c 300 $O/bench_default.json
