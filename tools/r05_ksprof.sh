cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_wide_dims.py -q --timeout 300 -k "wide_dims_ppo_minibatch_steps" 2>&1 | grep -v "^  \|amdgpu" | tail -${KS_TAIL:-3}
SPO_LIB_PATH=$PWD/safe-policy-optimization_amd/safepo/_lib/variants/libsafepo_hip_ksprof.so SPO_LIB_OVERRIDE=1 SPO_KS_PROF=1 timeout 120 python tools/ks_bench.py 60,20 376,17 2>&1 | grep -v "amdgpu.ids\|WARNING"
timeout 120 python tools/ks_bench.py 60,20 130,8 376,17 512,32 2>&1 | grep -v "amdgpu.ids\|WARNING"
