cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/_pa
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_pa -o p -- python3 tools/timing/time_attention.py > /dev/null 2>&1
F=$(find gpurun_out/_pa -name '*kernel_stats.csv' | head -1)
python3 tools/prof/kstats.py $F | grep -E "attn|attention|qmatmul|pack_t" 
rm -rf gpurun_out/_pa
