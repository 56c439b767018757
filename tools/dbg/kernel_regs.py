#!/usr/bin/env python3
"""register / spill / occupancy table of one .hip file's kernels:  tools/dbg/kernel_regs.py llm-mixed-q_amd/csrc/mi355q_matmul.hip"""
import os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                    "-c", sys.argv[1], "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs Spill": "sspill", "VGPRs Spill": "vspill", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds"}
rows, cur = [], None
for l in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, short in keys.items():
        m = re.search(r"    " + re.escape(k) + r": (\d+)", l)
        if m and cur is not None:
            cur[short] = m.group(1)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{name[:78]:78s} " + " ".join(f"{s} {r.get(s, '?'):>4}" for s in keys.values()))
