# GPU box: the row-group gradient kernel of the wide path (round 6 item 6): parity tests + us per step at hidden [128, 128]
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_wide_dims.py -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wide" 2>&1 | tail -5
{ timeout 300 python tools/wide_bench.py --small
  echo "-- SPO_WIDE_ROWS_FUSED=0 (group sum, norm, coefficient, Adam as four launches)"; SPO_WIDE_ROWS_FUSED=0 timeout 300 python tools/wide_bench.py --small
  echo "-- SPO_WIDE_GRAPH_UNROLL=1 (one step per graph launch, as rounds 4-5)"; SPO_WIDE_GRAPH_UNROLL=1 timeout 300 python tools/wide_bench.py --small
  echo "-- SPO_WIDE_ROWS=0 (the launch-per-network step of rounds 4-5, eight steps per graph launch)"; SPO_WIDE_ROWS=0 timeout 300 python tools/wide_bench.py --small
  echo "-- SPO_WIDE_ROWS=0 SPO_WIDE_GRAPH_UNROLL=1 (rounds 4-5 as measured then)"; SPO_WIDE_ROWS=0 SPO_WIDE_GRAPH_UNROLL=1 timeout 300 python tools/wide_bench.py --small
  echo "-- other shapes: [256, 256] at 60 / 8, [128, 128] at 376 / 17, [64, 64, 64] at 60 / 8; batch 64"
  timeout 300 python - <<'PY'
import json, os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import wide_bench
for hidden, D, A in (([256, 256], 60, 8), ([128, 128], 376, 17), ([64, 64, 64], 60, 8)):
    r = wide_bench.one(hidden, 64, 256, D=D, A=A)
    print(json.dumps({k: r[k] for k in ("hidden_sizes", "us_per_minibatch_step", "params")} | {"obs_dim": D, "act_dim": A}))
PY
} 2>&1 | grep -v "amdgpu.ids\|WARNING\|^+" | tee gpurun_out/r06/wide_step.txt
V=$GRAFT_REPO_ROOT/safe-policy-optimization_amd/safepo/_lib/variants/libsafepo_hip_mrprof.so
SPO_LIB_PATH=$V SPO_LIB_OVERRIDE=1 timeout 200 python tools/phase_profile_rows.py 2>&1 | grep -v "WARNING\|amdgpu.ids" > gpurun_out/r06/wide_rows_phase_cycles.txt
