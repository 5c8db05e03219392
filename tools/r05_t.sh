cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_gpu_parity.py -q --timeout 900 -s -k "$1" > gpurun_out/r05/t.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r05/t.txt | tail -${2:-60}
