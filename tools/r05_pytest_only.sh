cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
python -m pytest tests -m gpu -q -s --timeout 1200 > /tmp/pytest_full.txt 2>&1
{ echo "# python -m pytest tests -m gpu -q -s  (lines the tests print about what they measured, then the summary)"
  grep -v "amdgpu.ids" /tmp/pytest_full.txt | grep -a "rows at a clip boundary\|drift envelope (ratio\|drift envelope ratios\|feature-split kernel, 376\|wide minibatch step (\|yardstick report" | cut -c1-400
  grep -v "amdgpu.ids" /tmp/pytest_full.txt | tail -4; } > $O/pytest_gpu_final.log
tail -3 $O/pytest_gpu_final.log
