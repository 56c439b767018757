"""bf16 flavour of the tile GEMM (operands with per-block exponents) vs the library's bf16 GEMM, HIP events."""
import sys; sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device('cuda:0')
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
for M, K, N in ((4096, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008), (2048, 8192, 2048), (2048, 3072, 768)):
    x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
    w = torch.randn(N, K, device=dev) * 0.02
    xt = ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127)
    wt = ops.block_fp_quantize_bf16_tiled(w, 6, 8, 127, reuse=False)
    y = torch.empty(M, N, device=dev)
    tq = t(lambda: ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127))
    tg = t(lambda: ops.bf16_gemm_tiled(xt, wt, M, N, K, None, out=y))
    xb = ops.block_fp_quantize_bf16(x, 6, 8, 127, [1, 16], True); wb = w.to(torch.bfloat16)
    tl = t(lambda: torch.mm(xb, wb.t(), out_dtype=torch.float32))
    tq2 = t(lambda: ops.block_fp_quantize_bf16(x, 6, 8, 127, [1, 16], True))
    fl = 2.0 * M * N * K
    print(f"M={M} K={K} N={N}: quantise->tiled {tq:.1f} us, tile GEMM bf16 {tg:.1f} us = {fl/tg/1e6:.0f} TF/s | library: quantise {tq2:.1f} us, mm {tl:.1f} us = {fl/tl/1e6:.0f} TF/s")
