# GPU box: the round's closing runs on the final build (gpurun_out/r05 -> profiles/r05).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
timeout 900 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -q -s --timeout 1200 2>&1 | grep -v "amdgpu.ids\|^$" | tail -60 > $O/pytest_gpu_final.log; tail -3 $O/pytest_gpu_final.log
