#!/bin/bash
# rocprofv3 counter passes over the C-ABI step driver (one counter set per pass; --pmc passes over the Python bench do
# not finish on this pool).  Run on the GPU box from the repo root:  bash tools/prof/pmc_passes.sh [matmul]
# Output: gpurun_out/pmc_<set>/…_counter_collection.csv ; tools/pmc_summary.py turns FETCH/WRITE into profiles/*.json
export TMPDIR=/tmp
MODE=${1:-}
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  NAME=$(echo "$SET" | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmc_${NAME}${MODE:+_$MODE} -o c -- tools/cdriver/step_driver 3 $MODE | tail -1
done
