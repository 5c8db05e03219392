#!/bin/bash
# round 6: first run of the row-split update kernel on the GPU box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
{
  echo "== pytest (update kernel tests)"
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "minibatch_grad_and_step or intermittent_clip or split_path_equals or long_trajectory_parity or learning_iteration_is_deterministic" 2>&1 | tail -30
  echo "== update_ab"
  timeout 300 python tools/update_ab.py 2>&1 | tail -5
  echo "== update_ab form 2"
  SPO_UPDATE_FORM=2 timeout 300 python tools/update_ab.py 2>&1 | tail -5
  echo "== phase profile rs"
  timeout 300 python tools/phase_profile_rs.py 2>&1 | tail -40
  echo "== safe mode"
  SPO_RS_SAFE=1 timeout 300 python tools/update_ab.py 2>&1 | tail -3
} > gpurun_out/r06/rs_first.log 2>&1
tail -120 gpurun_out/r06/rs_first.log
