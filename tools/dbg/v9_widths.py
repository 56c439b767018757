"""the tile GEMM alone (events around 300 launches after 600) at several widths, full step too; run with MI355Q_V9_FIX=0 / 1"""
import os, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch, bench
from mi355q import ops
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device('cuda:0')
x, w, b = bench.make_inputs(torch, dev, 0)
y = torch.empty(4096, 4096, device=dev)
for width in (4, 5, 6):
    _, wm, we = ops.block_fp_quantize(w, width, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, width - 1, 127); bq = ops.block_fp_quantize(b, width, 8, 127, [16], False)
    xa = ops.block_fp_quantize_aligned_rows(x, width, 8, 127)
    nx = int(xa.sparse[4:4 + 16].sum()) if xa.sparse.numel() > 32 else -1
    for _ in range(600): ops.bfp_gemm_aligned(xa, wa, bq, out=y)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): ops.bfp_gemm_aligned(xa, wa, bq, out=y)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 300)
    def step():
        xq = ops.block_fp_quantize_aligned_rows(x, width, 8, 127)
        ops.bfp_gemm_aligned(xq, wa, bq, out=y)
    for _ in range(300): step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): step()
    e1.record(); torch.cuda.synchronize()
    print(f"V9_FIX={os.environ.get('MI355Q_V9_FIX', '0')} W{width}A{width}: GEMM {best:6.2f} us = {2*4096**3/best*1e-6:6.0f} TOPS; step {e0.elapsed_time(e1) * 1000 / 300:6.2f} us", flush=True)
