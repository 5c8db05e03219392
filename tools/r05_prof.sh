cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/phase_profile_h.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/prof_h.txt
cat gpurun_out/r05/prof_h.txt
