cd $GRAFT_REPO_ROOT
V=safe-policy-optimization_amd/safepo/_lib/variants
python tools/kl_ab.py "" $V/libsafepo_hip_vgprform.so
for lib in "" vgprform; do
  if [ -n "$lib" ]; then export SPO_LIB_PATH=$GRAFT_REPO_ROOT/$V/libsafepo_hip_$lib.so SPO_LIB_OVERRIDE=1; else unset SPO_LIB_PATH SPO_LIB_OVERRIDE; fi
  timeout 200 python bench.py --algo cpo --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cpo', '${lib:-in-tree}', d['value'], d['phases'], d['cpo_fvp']['avg_us'])"
done
unset SPO_LIB_PATH SPO_LIB_OVERRIDE
python -m pytest tests -m gpu -q -x -k "policy_step or collect or actor_kl or cpo_surrogate or fvp or bench_self or smoke or fused_normalise or entrypoint" 2>&1 | tail -4
