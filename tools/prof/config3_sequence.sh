#!/bin/bash
# Kernel trace of the full-depth config-3 forward (eager, 4 layers are enough for the per-layer sequence): one decoder layer's dispatches in order with
# durations and gaps, and the per-kernel totals of the whole run.  Run on the GPU box from the repo root:  bash tools/prof/config3_sequence.sh [layers [--no-spread]]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
L=${1:-4}; shift
rm -rf gpurun_out/_c3
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_c3 -o p -- python3 tools/config3_full_depth.py --layers $L --steps 3 --no-graph --no-parity "$@" > gpurun_out/c3_seq.json 2> gpurun_out/c3_seq.err
F=$(find gpurun_out/_c3 -name '*kernel_trace.csv' | head -1)
python3 - "$F" "$L" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
L = int(sys.argv[2])
# the last forward: find the last lm_head GEMM (the longest dispatch) and walk back one forward
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
big = [i for i, r in enumerate(rows) if dur(r) > 1500]
end = big[-1]
start = big[-2] + 1 if len(big) > 1 else 0
fwd = rows[start:end + 1]
t0 = int(fwd[0]["Start_Timestamp"])
print(f"last forward: {len(fwd)} dispatches, {(int(fwd[-1]['End_Timestamp']) - t0) / 1e3:.0f} us, busy {sum(dur(r) for r in fwd):.0f} us")
per = len(fwd) // L
seg = fwd[per * (L - 2) + 4: per * (L - 1) + 8]
prev = None
for r in seg:
    nm = r["Kernel_Name"].split("(")[0][-64:]
    gap = (int(r["Start_Timestamp"]) - prev) / 1e3 if prev else 0.0
    prev = int(r["End_Timestamp"])
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  gap {gap:6.1f}  +{dur(r):7.1f}  {nm}')
tot = collections.Counter(); cnt = collections.Counter()
for r in fwd:
    nm = r["Kernel_Name"].split("(")[0][-64:]
    tot[nm] += dur(r); cnt[nm] += 1
print("--- per kernel, last forward")
for nm, t in tot.most_common(22):
    print(f"{t:9.1f} us  x{cnt[nm]:4d}  {nm}")
PY
rm -rf gpurun_out/_c3
