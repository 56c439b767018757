"""bf16 flavour of the tile GEMM, one (tile rows, forced split) setting per process."""
import json, os, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device("cuda:0")
M, K, N = (int(v) for v in sys.argv[1:4])
x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
w = torch.randn(N, K, device=dev) * 0.02
xt = ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127)
wt = ops.block_fp_quantize_bf16_tiled(w, 6, 8, 127, reuse=False)
y = torch.empty(M, N, device=dev)
for _ in range(5):
    ops.bf16_gemm_tiled(xt, wt, M, N, K, None, out=y)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30):
    ops.bf16_gemm_tiled(xt, wt, M, N, K, None, out=y)
e.record()
torch.cuda.synchronize()
us = a.elapsed_time(e) / 30 * 1e3
print(json.dumps({"bf16": True, "M": M, "K": K, "N": N, "tile_rows": os.environ.get("MI355Q_V8_TILE_ROWS", "auto"),
                  "splits": os.environ.get("MI355Q_V8_SPLITS", "auto"), "gemm_us": round(us, 1), "TFLOPs": round(2.0 * M * N * K / us / 1e6)}))
