for cap in 120 1016 120 1016; do
  MI355Q_X_BUCKET_CAP=$cap python bench.py --steps 200 --warmup 20 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('cap $cap', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
