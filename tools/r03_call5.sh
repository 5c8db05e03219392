cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03e
mkdir -p $O
V=$GRAFT_REPO_ROOT/safe-policy-optimization_amd/safepo/_lib/variants
for q in "" 24; do
  echo "== packed 16-byte words, GPU_MAX_HW_QUEUES=$q" >> $O/p2p_loopback.txt
  GPU_MAX_HW_QUEUES=$q timeout 200 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep -v "amdgpu.ids\|^xr profile" | tail -4 >> $O/p2p_loopback.txt
done
echo "== 8-byte words (SPO_XR_PACK16=0 build), GPU_MAX_HW_QUEUES=24" >> $O/p2p_loopback.txt
SPO_LIB_PATH=$V/libsafepo_hip_pack8.so SPO_LIB_OVERRIDE=1 GPU_MAX_HW_QUEUES=24 timeout 200 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep -v "amdgpu.ids\|^xr profile" | tail -4 >> $O/p2p_loopback.txt
cat $O/p2p_loopback.txt
