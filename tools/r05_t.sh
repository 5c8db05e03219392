cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wide_dims.py -q -x --timeout 600 -k "critic_fit" 2>&1 | grep -v "^  \|amdgpu" | tail -30
