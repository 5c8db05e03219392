# GPU box, round 5 experiment 1: packed helper-side exchange (SPO_P2P_HELPER=2) and the split critic fit on the main + helper kernel.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
{
echo "== correctness"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "critic_fit or in_kernel_gradient_exchange or cpo_update_vs_reference or cpo_data_parallel" 2>&1 | tail -8
echo "== cpo bench A/B (split form h vs 4w)"
for form in h 4w; do
  SPO_CPO_SPLIT_FORM=$form timeout 300 python bench.py --algo cpo --no-cpu-baseline --steps 3 --warmup 1 2>$O/cpo_$form.err | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$form', l['value'], l['ms_per_step'], l['phases'])"
done
echo "== loopback default"
timeout 600 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep -v "^xr profile"
echo "== loopback SPO_P2P_HELPER=2 (doubling forced)"
SPO_P2P_HELPER=2 SPO_P2P_ALGO=doubling timeout 600 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep -v "^xr profile"
echo "== loopback SPO_P2P_HELPER=1 (doubling forced, old words)"
SPO_P2P_HELPER=1 SPO_P2P_ALGO=doubling timeout 600 python tools/p2p_loopback_bench.py 2 2>&1 | grep -v "^xr profile"
} > $O/exp1.txt 2>&1
cat $O/exp1.txt
