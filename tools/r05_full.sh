cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -s ${1:+-k "$1"} > gpurun_out/r05/full.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r05/full.txt | grep -n "passed\|failed\|FAILED\|Error\|rows at a clip" | tail -40
