cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
V=$GRAFT_REPO_ROOT/safe-policy-optimization_amd/safepo/_lib/variants
for lib in "" vb8 vb15; do
  echo "== poll batch variant: ${lib:-in-tree (5)}" | tee -a $O/loopback_vb.txt
  if [ -n "$lib" ]; then export SPO_LIB_PATH=$V/libsafepo_hip_$lib.so SPO_LIB_OVERRIDE=1; else unset SPO_LIB_PATH SPO_LIB_OVERRIDE; fi
  timeout 200 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep '^{' | tee -a $O/loopback_vb.txt
  timeout 200 python bench.py --algo cpo --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cpo', d['value'], d['phases']['update_us_per_minibatch_step'])" | tee -a $O/loopback_vb.txt
done
