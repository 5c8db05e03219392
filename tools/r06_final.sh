# GPU box: the round's closing runs on the final build (gpurun_out/r06 -> profiles/r06 via tools/copy_profiles_r06.sh).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O
timeout 700 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -q -s --timeout 1200 > /tmp/pytest_full.txt 2>&1
{ echo "# python -m pytest tests -m gpu -q -s  (lines the tests print about what they measured, then the summary)"
  grep -v "amdgpu.ids" /tmp/pytest_full.txt | grep -a "rows at a clip boundary\|drift envelope (ratio\|feature-split kernel, 376\|wide minibatch step (\|exemption IN USE\|elements beyond\|second moment of the elements\|_ks_kernel" | cut -c1-400
  grep -v "amdgpu.ids" /tmp/pytest_full.txt | tail -4; } > $O/pytest_gpu_final.log
tail -3 $O/pytest_gpu_final.log
