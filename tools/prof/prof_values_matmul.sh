#!/bin/bash
# Kernel durations of tools/timing/time_values_matmul.py (the attention products of the three block arithmetics at the Llama-7B
# shapes) by rocprofv3: which launches a product is made of.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/_prof_vm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_prof_vm -o vm -- python3 tools/timing/time_values_matmul.py > gpurun_out/vm_prof.out 2> gpurun_out/vm_prof.err
python3 tools/prof/kstats.py $(find gpurun_out/_prof_vm -name '*kernel_stats.csv' | head -1) | tee gpurun_out/vm_kernel_stats.txt
rm -rf gpurun_out/_prof_vm
