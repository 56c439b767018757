#!/bin/bash
# rocprofv3 counter passes over the streaming fake-quantisers at the Llama-7B activation shapes (C driver, one counter per
# pass): bash tools/prof/pmc_quantizers.sh   ->  gpurun_out/r02_pmc_quantizers.txt  (FETCH_SIZE is tallied at 64 B per 128-B
# request on gfx950: x 2, as tools/prof/pmc_traffic.py does; rocprofv3 reports KiB)
export TMPDIR=/tmp
OUT=gpurun_out/r02_pmc_quantizers.txt
: > $OUT
for SET in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcq_$SET
  timeout 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmcq_$SET -o c -- tools/cdriver/step_driver 3 quantizers | tail -2
  F=$(find gpurun_out/pmcq_$SET -name '*counter_collection.csv' | head -1)
  python3 - "$F" $SET >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and "quant_vec_kernel" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-36:], r["Grid_Size"] if "Grid_Size" in r else "")].append(float(r["Counter_Value"]))
for (k, g), v in sorted(acc.items()):
    v = sorted(v)
    print(sys.argv[2], k, "launches", len(v), "KiB per launch (small shape .. large shape):", round(v[0]), "..", round(v[-1]))
PY
  rm -rf gpurun_out/pmcq_$SET
done
cat $OUT
