cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu_final.log; tail -3 $O/pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json
