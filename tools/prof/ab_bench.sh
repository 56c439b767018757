for i in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
