#!/bin/bash
# one decoder layer's dispatches of config 4 (OPT-1.3B width, W4A4 mixed, unsharded), in order
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/_c4
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_c4 -o p -- python3 tools/config4_sharded.py --layers 4 --steps 3 > gpurun_out/c4_seq.json 2> gpurun_out/c4_seq.err
F=$(find gpurun_out/_c4 -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
big = [i for i, r in enumerate(rows) if dur(r) > 700]
end = big[-1]; start = big[-2] + 1
fwd = rows[start:end + 1]
t0 = int(fwd[0]["Start_Timestamp"])
print(f"last forward: {len(fwd)} dispatches, {(int(fwd[-1]['End_Timestamp']) - t0) / 1e3:.0f} us, busy {sum(dur(r) for r in fwd):.0f} us")
per = len(fwd) // 4
prev = None
for r in fwd[per * 2 + 3: per * 3 + 6]:
    nm = r["Kernel_Name"].split("(")[0][-64:]
    gap = (int(r["Start_Timestamp"]) - prev) / 1e3 if prev else 0.0
    prev = int(r["End_Timestamp"])
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  gap {gap:6.1f}  +{dur(r):7.1f}  {nm}')
tot = collections.Counter(); cnt = collections.Counter()
for r in fwd:
    nm = r["Kernel_Name"].split("(")[0][-64:]
    tot[nm] += dur(r); cnt[nm] += 1
print("--- per kernel, last forward")
for nm, t in tot.most_common(20):
    print(f"{t:9.1f} us  x{cnt[nm]:4d}  {nm}")
PY
rm -rf gpurun_out/_c4
