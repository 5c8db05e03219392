#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{
  echo "== four-wave kernel (SPO_P2P_ALGO=pair: doubling at 2 / 4 ranks, two-phase at 8: the default policy of rounds 3-5), M = 8192"
  SPO_P2P_ALGO=pair timeout 600 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep "^{"
  echo "== row-split form (the default since round 6; all-to-all at 2 ranks, reduce-scatter + all-gather at 4 / 8), M = 8192"
  SPO_P2P_ALGO=rowsplit timeout 600 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep "^{"
  echo "== SPO_P2P_ALGO=rowsplit, M = 262144 rows per rank"
  SPO_LOOPBACK_M=262144 SPO_P2P_ALGO=rowsplit timeout 600 python tools/p2p_loopback_bench.py 2 4 2>&1 | grep "^{"
  echo "== four-wave kernel, M = 262144 rows per rank"
  SPO_P2P_ALGO=pair SPO_LOOPBACK_M=262144 timeout 600 python tools/p2p_loopback_bench.py 2 2>&1 | grep "^{"
} > gpurun_out/r06/p2p_loopback.txt 2>&1
cat gpurun_out/r06/p2p_loopback.txt
