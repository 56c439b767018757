"""The row quantiser behind LlamaRMSNorm / LayerNorm (pre-op flavour) at the model shapes: 2048 rows of 4096, HIP events, us.
MI355Q_QROWS_GRID=<n> caps its grid (several rows a workgroup, the next row's loads in flight); MI355Q_QROWS_SHORT=0: rows of <= 2048
values on the four-slab build (before round 6) -- run once per setting.  "plain" alternates eight tensors (a repeated tensor is served
from the recorded operand)."""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import torch
from mi355q import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
row = {"MI355Q_QROWS_GRID": os.environ.get("MI355Q_QROWS_GRID", ""), "MI355Q_QROWS_SHORT": os.environ.get("MI355Q_QROWS_SHORT", "")}
for rows, K in ((2048, 4096), (2048, 2048), (2048, 1024), (2048, 768), (4096, 2048)):
    xs = [torch.randn(rows, K, generator=g).to(dev) for _ in range(8)]
    x = xs[0]
    w = (1 + 0.1 * torch.randn(K, generator=g)).to(dev)
    b = (0.1 * torch.randn(K, generator=g)).to(dev)
    for name, pre in (("plain", None), ("rmsnorm", ("rmsnorm", w, 1e-6)), ("layernorm", ("layernorm", w, 1e-5, b))):
        cnt = [0]
        def fn():
            cnt[0] += 1
            return ops.block_fp_quantize_aligned_rows(xs[cnt[0] & 7] if pre is None else x, 6, 8, 127, pre=pre)
        for _ in range(10): fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): fn()
        e.record(); torch.cuda.synchronize()
        row[f"{rows}x{K} {name}"] = round(a.elapsed_time(e) * 10, 1)
print(json.dumps(row))
