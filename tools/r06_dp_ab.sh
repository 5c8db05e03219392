export HSA_ENABLE_IPC_MODE_LEGACY=0 SPO_BENCH_ONE_GPU=1
run() { timeout 900 python bench.py --gpus 2 --steps $3 --warmup 1 --learning-iters $2 --no-cpu-baseline --no-config3 --no-config5 --no-wide 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 iters $2 steps $3', d['value'], [p['update_us_per_minibatch_step'] for p in d['per_rank']], d['update_kernel']['redo_counters'])
"; }
unset SPO_P2P_ALGO; run auto 40 2; run auto 40 3
