"""One shape, one (tile rows, forced split) setting per process: the tile GEMM's time (env knobs are read once)."""
import json, os, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device("cuda:0")
M, K, N = (int(v) for v in sys.argv[1:4])
g = torch.Generator().manual_seed(M + K + N)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
b = (torch.randn(N, generator=g) * 0.02).to(dev)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
wa = ops.bfp_align_rows(wm, we, 5, 127)
xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
out = torch.empty(M, N, device=dev)
for _ in range(5):
    ops.bfp_gemm_aligned(xa, wa, b, out=out)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    ops.bfp_gemm_aligned(xa, wa, b, out=out)
e.record()
torch.cuda.synchronize()
us = a.elapsed_time(e) / 50 * 1e3
print(json.dumps({"M": M, "K": K, "N": N, "tile_rows": os.environ.get("MI355Q_V8_TILE_ROWS", "auto"),
                  "splits": os.environ.get("MI355Q_V8_SPLITS", "auto"), "gemm_us": round(us, 1), "TOPS": round(2.0 * M * N * K / us / 1e6)}))
