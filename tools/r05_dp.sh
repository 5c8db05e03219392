cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
SPO_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --no-config5 > gpurun_out/r05/bench_dp2.json 2> gpurun_out/r05/bench_dp2.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r05/bench_dp2.json").read().strip().splitlines()[-1])
print(json.dumps(l["exchange"], indent=1)); print(l["value"], l["per_rank"])
PY
tail -5 gpurun_out/r05/bench_dp2.err
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "bench_self_launches or in_kernel_gradient_exchange or data_parallel_global_batch or cpo_data_parallel" 2>&1 | tail -4
