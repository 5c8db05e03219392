cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wide_dims.py tests/test_gpu_parity.py -q --timeout 600 -k "two_feature_split_engines or two_engines_update_concurrently or critic_fit or wide_dims_ppo_minibatch" 2>&1 | grep -v "^  \|amdgpu" | tail -15
