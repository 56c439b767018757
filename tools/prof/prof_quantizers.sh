#!/bin/bash
# Per-dispatch kernel durations of `bench.py --workload quantizers` (rocprofv3 kernel trace), grouped in dispatch order
# by (shape, quantiser).  Run on the GPU box from the repo root:  bash tools/prof/prof_quantizers.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-q}; STEPS=20; WARM=5
rm -rf gpurun_out/_prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_prof_$TAG -o p -- python3 bench.py --workload quantizers --steps $STEPS --warmup $WARM > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_prof.err
F=$(find gpurun_out/_prof_$TAG -name '*kernel_trace.csv' | head -1)
python3 - "$F" $STEPS $WARM <<'PY' | tee gpurun_out/${TAG}_kernel_durations.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
steps, warm = int(sys.argv[2]), int(sys.argv[3])
rows = [r for r in rows if "mi355q" in r["Kernel_Name"]]
shapes = ["act[2048,4096]", "act[2048,11008]", "probs[32,2048,2048]", "w[4096,4096]", "w[11008,4096]"]
fns = ["block_fp_w6", "block_minifloat_w8e4", "block_log_w8"]
per_call = len(rows) // (len(shapes) * len(fns) * (steps + warm))
print("kernels per call:", per_call, "dispatches:", len(rows))
i = 0
for s in shapes:
    for f in fns:
        chunk = rows[i:i + per_call * (steps + warm)]
        i += per_call * (steps + warm)
        timed = chunk[per_call * warm:]
        by = collections.defaultdict(list)
        for r in timed:
            by[r["Kernel_Name"].split("(")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        span = (int(timed[-1]["End_Timestamp"]) - int(timed[0]["Start_Timestamp"])) / 1e3 / steps
        print(f"{s:22s} {f:22s} per-call span {span:7.2f} us | " + " | ".join(f"{k}: {sum(v)/len(v):.2f} us" for k, v in by.items()))
PY
rm -rf gpurun_out/_prof_$TAG
