cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03g
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_nocpu.json 2> $O/bench_nocpu.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03g/bench_nocpu.json").read().strip().splitlines()[-1])
print(d["value"], d["update_kernel"]["us_per_minibatch_step"], d["kl_kernel"]["avg_us"], d["kl_kernel"]["frac"], d["config3_cpo"]["value"], d["config3_cpo"]["cpo_fvp"]["avg_us"], d["config5_mappolag"])
PY
