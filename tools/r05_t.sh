cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wide_dims.py -q --timeout 600 -k "kl_penalty or default_sweep" 2>&1 | grep -v "^  \|amdgpu" | tail -30
