#!/bin/bash
# rocprofv3 counter passes over the one-pass attention kernel (C driver): bash tools/prof/pmc_attention.sh <heads> <head_dim>
export TMPDIR=/tmp
H=${1:-12}; D=${2:-64}
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  NAME=$(echo "$SET" | cut -d' ' -f1)
  rm -rf gpurun_out/pmca_$NAME
  timeout 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmca_$NAME -o c -- tools/cdriver/step_driver 3 attention $H $D | tail -1
  F=$(find gpurun_out/pmca_$NAME -name '*counter_collection.csv' | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][-50:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "attention_kernel" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  rm -rf gpurun_out/pmca_$NAME
done
