"""The per-rank GEMM shapes of the row-sharded layer (tools/timing/time_shard_shapes.py) under forced tile heights / K splits: which
(tile rows, splits) the launcher should pick where the grid is under-filled.  TOPS per shape and setting."""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
table = {}
for rows in (0, 128, 256):
    for s in (0, 1, 2, 4, 8):
        env = dict(os.environ, MI355Q_V8_TILE_ROWS=str(rows), MI355Q_V8_SPLITS=str(s))
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "time_shard_shapes.py")], env=env, capture_output=True, text=True, cwd=root)
        for l in r.stdout.splitlines():
            try:
                d = json.loads(l)
            except Exception:
                continue
            table.setdefault(d["shape"] + f" {d['M']}x{d['N']}x{d['K']}", {})[(rows, s)] = d["gemm_us"]
keys = [(rows, s) for rows in (0, 128, 256) for s in (0, 1, 2, 4, 8)]
print("us per GEMM; columns = (tile rows, splits), 0 = the launcher's choice")
print(" " * 40 + " ".join(f"{r}/{s}".rjust(7) for r, s in keys))
for shape, row in table.items():
    print(f"{shape:40s}" + " ".join(f"{row.get(k, float('nan')):7.1f}" for k in keys))
