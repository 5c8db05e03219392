# GPU box: the data-parallel form at the feature-split dims (round 6 item 4b): parity tests + the launch-side timing
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_wide_dims.py -q -x -s -k "gradient_launch" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -s -k "data_parallel_wide_engine" 2>&1 | tail -8
timeout 300 python tools/ks_grad_bench.py 376,17 130,8 2>&1 | grep -v "amdgpu.ids\|WARNING" | tee gpurun_out/r06/dp_feature_split_step.txt
