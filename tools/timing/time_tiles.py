"""Main-kernel time of the row-aligned GEMM with 256- and 128-row workgroup tiles (library HIP events). argv: M N K"""
import sys, os
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device('cuda:0')
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 4096, 4096)
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
wa = ops.bfp_align_rows(wm, we, 5, 127)
xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
y = torch.empty(M, N, device=dev)
res = {}
for rnd in range(4):
    for rows in ("256", "128"):
        os.environ["MI355Q_V8_TILE_ROWS"] = rows
        for _ in range(3): ops.bfp_gemm_aligned(xa, wa, out=y)
        torch.cuda.synchronize()
        ops.gemm_timing(True)
        for _ in range(30): ops.bfp_gemm_aligned(xa, wa, out=y)
        torch.cuda.synchronize(); ops.gemm_timing(False)
        res.setdefault(rows, []).append(ops.gemm_timing_read()[1] * 1e3)
for rows, v in res.items():
    v.sort(); print(f"M={M} N={N} K={K} tile rows {rows}: main kernel median {v[len(v)//2]:.2f} us -> {2*M*N*K/v[len(v)//2]/1e6:.0f} TOPS")
