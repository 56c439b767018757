#!/bin/bash
# experiment: epilogue store flavours of the 256 x 256 tile GEMM (MI355Q_V9_DBG bits 8 sc1 / 16 sc0 sc1 / 32 nt)
mkdir -p gpurun_out/r4
for d in 0 8 16 32 0; do
  echo "== MI355Q_V9_DBG=$d" 
  MI355Q_V9_DBG=$d python tools/dbg/v9_time.py 2>&1 | tail -6
done
