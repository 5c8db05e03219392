cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $O
python -m pytest tests -m gpu -q -x -k "exchange or data_parallel or cpo or kl or split or dp or trust or pcpo" 2>&1 | tail -12 > $O/pytest_subset.log
tail -3 $O/pytest_subset.log
timeout 300 python tools/p2p_loopback_bench.py 2 4 8 > $O/p2p_loopback.txt 2>&1; tail -6 $O/p2p_loopback.txt
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03c/bench_short.json").read().strip().splitlines()[-1])
print(d["value"], d["kl_kernel"]["avg_us"], d["kl_kernel"]["frac"], d["config3_cpo"]["value"], d["config3_cpo"]["cpo_fvp"]["avg_us"], d["config3_cpo"]["cpo_fvp"]["frac"])
PY
