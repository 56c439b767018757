#!/usr/bin/env python3
"""SQ counters per kernel from the rocprofv3 passes of tools/prof/pmc_passes.sh (gpurun_out/pmc_SQ_*/..._counter_collection.csv):
averages over the steady-state launches (the first launch of a kernel dropped when there are several).
    python tools/prof/pmc_sq.py gpurun_out profiles/r03_pmc_sq.json"""
import csv, glob, json, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for path in [p for d in ("pmc_SQ_WAVE_CYCLES", "pmc_SQ_INSTS_VALU") for p in glob.glob(f"{root}/{d}/**/*counter_collection.csv", recursive=True)]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "mi355q" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {k: {c: int(sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in cs.items()} for k, cs in acc.items()}
json.dump({"command": "bash tools/prof/pmc_passes.sh (rocprofv3 --pmc <set> --kernel-trace -- tools/cdriver/step_driver 3; averages over the "
                      "steady-state launches)", "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps(kern, indent=1)[:3000])
