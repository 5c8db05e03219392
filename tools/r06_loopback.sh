#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{
  echo "== default policy (doubling at 2 / 4 ranks, two-phase at 8), M = 8192"
  timeout 600 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep "^{"
  echo "== SPO_P2P_ALGO=rowsplit, M = 8192"
  SPO_P2P_ALGO=rowsplit timeout 600 python tools/p2p_loopback_bench.py 2 4 8 2>&1 | grep "^{"
  echo "== SPO_P2P_ALGO=rowsplit, M = 262144 rows per rank"
  SPO_LOOPBACK_M=262144 SPO_P2P_ALGO=rowsplit timeout 600 python tools/p2p_loopback_bench.py 2 4 2>&1 | grep "^{"
  echo "== default, M = 262144 rows per rank"
  SPO_LOOPBACK_M=262144 timeout 600 python tools/p2p_loopback_bench.py 2 2>&1 | grep "^{"
} > gpurun_out/r06/p2p_loopback.txt 2>&1
cat gpurun_out/r06/p2p_loopback.txt
