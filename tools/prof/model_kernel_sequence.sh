#!/bin/bash
# The kernel sequence of one decoder layer in steady state (rocprofv3 kernel trace of tools/prof/prof_model.py): name and duration
# of ~80 consecutive dispatches from the last forward.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/_pm
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_pm -o p -- python3 tools/prof/prof_model.py "$@" > /dev/null 2> gpurun_out/pm.err
F=$(find gpurun_out/_pm -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
seg = rows[n - 420:n - 330]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    nm = r["Kernel_Name"].split("(")[0][-70:]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f}  {nm}')
PY
rm -rf gpurun_out/_pm
