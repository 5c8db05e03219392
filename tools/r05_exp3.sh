cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
{
echo "== update_ab: in-tree vs variants"
python tools/update_ab.py "" $(ls safe-policy-optimization_amd/safepo/_lib/variants/*.so 2>/dev/null) 2>&1
echo "== phase profile (in-tree)"
python tools/phase_profile_h.py 2>&1 | grep -v amdgpu.ids
} > $O/exp3.txt 2>&1
cat $O/exp3.txt
