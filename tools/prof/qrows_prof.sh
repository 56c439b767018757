#!/bin/bash
# Kernel-level duration of the fused activation quantiser alone (rocprofv3 kernel stats over tools/timing/time_qrows.py).
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/_pq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_pq -o p -- python3 tools/timing/time_qrows.py > /dev/null 2>&1
F=$(find gpurun_out/_pq -name '*kernel_stats.csv' | head -1)
python3 tools/prof/kstats.py $F | grep -E "align_rows"
rm -rf gpurun_out/_pq
