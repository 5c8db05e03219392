cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03d
mkdir -p $O
V=safe-policy-optimization_amd/safepo/_lib/variants
timeout 300 python tools/update_ab.py "" $V/libsafepo_hip_pipe1.so $V/libsafepo_hip_pipe2.so > $O/update_ab.txt 2>&1; cat $O/update_ab.txt | tail -8
GPU_MAX_HW_QUEUES=24 timeout 300 python tools/p2p_loopback_bench.py 2 4 8 > $O/p2p_loopback.txt 2>&1; grep -v "^xr profile\|amdgpu.ids" $O/p2p_loopback.txt | tail -6
