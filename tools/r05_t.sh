cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "ppo_lag_update_vs_reference_main_trace or kl_penalty_family_vs_reference or second_order_family or cpo_fvp_known or policy_step_golden or cpo_update_vs_reference" 2>&1 | grep -v "^  \|amdgpu" | tail -6
