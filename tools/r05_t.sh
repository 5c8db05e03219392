cd $GRAFT_REPO_ROOT
timeout 300 python tools/update_ab.py safe-policy-optimization_amd/safepo/_lib/libsafepo_hip.so safe-policy-optimization_amd/safepo/_lib/variants/libsafepo_hip_bkbuf.so 2>&1 | tail -3
