#!/bin/bash
# the tile GEMM alone on the bench operands (tools/dbg/v9_time.py: variant 0 = with exception lists, 8 = without) by route
for env in "MI355Q_V9_FIX=0" "MI355Q_V9_FIX=1" "MI355Q_V9_FIX=1 MI355Q_V9_DBG=8"; do
  echo "== $env"; env $env python tools/dbg/v9_time.py 2>&1 | grep variant
done
echo "== stamps E path"; MI355Q_V9_FIX=1 python tools/dbg/v9_stamps.py 2>&1 | tail -2
