#!/usr/bin/env python3
"""HBM bytes per launch from rocprofv3 counter passes (tools/prof/pmc_passes.sh: one pass with FETCH_SIZE, one with WRITE_SIZE over
tools/cdriver/step_driver).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at
64 B for wide coalesced reads); WRITE_SIZE is taken as reported.  Units: rocprofv3 reports KiB.
    python tools/prof/pmc_traffic.py gpurun_out profiles/r02_pmc_traffic.json [profiles/r02_pmc_counters.csv]"""
import csv, glob, json, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
acc = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
rows_out = []
for counter in acc:
    for path in glob.glob(f"{root}/pmc_{counter}*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            acc[counter][r["Kernel_Name"]].append(float(r["Counter_Value"]))
            rows_out.append((counter, r["Kernel_Name"], r["Counter_Value"]))
kern = {}
for name in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    if not name.startswith("mi355q::") and "mi355q" not in name:
        continue
    f, w = acc["FETCH_SIZE"].get(name, []), acc["WRITE_SIZE"].get(name, [])
    # steady-state launches: drop the first (cold caches) when there are several
    fm = sum(f[1:]) / len(f[1:]) if len(f) > 1 else (f[0] if f else 0.0)
    wm = sum(w[1:]) / len(w[1:]) if len(w) > 1 else (w[0] if w else 0.0)
    kern[name.split("(")[0].replace("void ", "")] = {"fetch_KiB_raw": round(fm, 1), "write_KiB": round(wm, 1), "launches": max(len(f), len(w)),
                                                    "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- tools/cdriver/step_driver 3 (one counter per pass; the bench step "
                      "through the C ABI, inputs distributed as bench.py's)",
           "units": "rocprofv3 reports KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as is",
           "kernels": kern}, open(out, "w"), indent=1)
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fh:
        fh.write("counter,kernel,value_KiB\n")
        for c, k, v in rows_out:
            fh.write(f'{c},"{k.split("(")[0]}",{v}\n')
print(json.dumps(kern, indent=1))
