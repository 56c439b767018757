#!/bin/bash
# rocprofv3 kernel stats of the benchmark step.  Run on the GPU box from the repo root:  bash tools/prof/prof_step.sh <tag> [bench args]
# Writes gpurun_out/<tag>_kernel_stats.csv (the summary to copy into profiles/) and prints the per-kernel table.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-prof}; shift
rm -rf gpurun_out/_prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_prof_$TAG -o p -- python3 bench.py --no-cpu-baseline --no-verify --no-ceiling --steps 100 --warmup 10 "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_prof.err
F=$(find gpurun_out/_prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$F" gpurun_out/${TAG}_kernel_stats.csv
python3 tools/prof/kstats.py gpurun_out/${TAG}_kernel_stats.csv | head -12
rm -rf gpurun_out/_prof_$TAG
