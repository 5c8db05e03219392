#!/bin/bash
# round 6: data-parallel forms on ONE GPU (loopback): exactness tests of the row-split form, bench --gpus 2, loopback bench
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
  echo "== pytest rowsplit + bench self-launch"
  timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "in_kernel_gradient_exchange_two_ranks_one_gpu and (rowsplit or pair) or bench_self_launches" 2>&1 | tail -15
  echo "== bench --gpus 2 on one GPU"
  SPO_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-config5 --no-wide > gpurun_out/r06/bench_dp2_one_gpu.json 2> gpurun_out/r06/bench_dp2_one_gpu.err
  tail -c 400 gpurun_out/r06/bench_dp2_one_gpu.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_dp2_one_gpu.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["replicas_identical_after_run"]); print(json.dumps(d["exchange"])[:1500]); print(d["per_rank"])
PY
} > gpurun_out/r06/dp.log 2>&1
tail -60 gpurun_out/r06/dp.log
