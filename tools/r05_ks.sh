cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wide_dims.py -q --timeout 600 2>&1 | tail -5
SPO_KS_SAFE=1 timeout 600 python -m pytest tests/test_gpu_wide_dims.py -q --timeout 300 -k "wide_dims_ppo_minibatch_steps or default_sweep_algorithms_train_at_humanoid_dims" 2>&1 | tail -3
SPO_KS_SAFE=1 timeout 120 python tools/ks_bench.py 376,17 2>&1 | grep -v "amdgpu.ids\|WARNING"
