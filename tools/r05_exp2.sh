# GPU box, round 5 experiment 2: dW1 on the helper waves (SPO_H_DW1) A/B + the tests that gate the update kernel.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
{
echo "== update_ab: in-tree (SPO_H_DW1=1) vs variants"
python tools/update_ab.py "" $(ls safe-policy-optimization_amd/safepo/_lib/variants/*.so 2>/dev/null) 2>&1
echo "== phase profile (in-tree)"
python tools/phase_profile_h.py 2>&1 | grep -v amdgpu.ids
echo "== tests"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x --timeout 900 -k "minibatch_grad_and_step or intermittent_clip or full_size_update_parity or long_trajectory or deterministic or ppo_lag_update_vs_reference or limits_and_edge or split_path_equals or pg_unclipped" 2>&1 | tail -8
} > $O/exp2.txt 2>&1
cat $O/exp2.txt
