cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{
python tools/update_ab.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "critic_fit or in_kernel_gradient_exchange or intermittent_clip or minibatch_grad_and_step or pcpo" 2>&1 | tail -5
SPO_CPO_SPLIT_FORM=h timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "critic_fit" 2>&1 | tail -3
} > gpurun_out/r05/check.txt 2>&1
cat gpurun_out/r05/check.txt
