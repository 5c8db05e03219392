cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{
timeout 600 python -m pytest tests/test_gpu_wide_dims.py -q --timeout 600 -k "wide_dims_ppo_minibatch_steps or default_sweep_algorithms_train_at_humanoid_dims" 2>&1 | tail -15
timeout 120 python tools/ks_bench.py 60,8 60,20 130,8 200,20 376,17 512,32 2>&1 | grep -v amdgpu.ids
SPO_WIDE_KS=0 timeout 120 python tools/ks_bench.py 376,17 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05/ks.txt 2>&1
cat gpurun_out/r05/ks.txt
