cd $GRAFT_REPO_ROOT
SPO_CPO_SPLIT_FORM=h timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "critic_fit" 2>&1 | tail -2
for f in h 4w; do for v in 1 0; do
  SPO_CPO_SPLIT_FORM=$f SPO_CPO_SPLIT_L2=$v timeout 300 python bench.py --algo cpo --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('form=$f L2=$v', l['value'], l['ms_per_step'], l['phases']['update_us_per_minibatch_step'])"
done; done
