#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{
  for w in 0 2 5; do echo "== prof wg $w"; SPO_RS_PROF_WG=$w timeout 300 python tools/phase_profile_rs.py 2>&1 | tail -26; done
  echo "== update_ab"; timeout 300 python tools/update_ab.py 2>&1 | tail -3
} > gpurun_out/r06/rs_prof.log 2>&1
cat gpurun_out/r06/rs_prof.log
