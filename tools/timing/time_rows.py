"""Row-aligned GEMM at 4096^3: main-kernel time (library HIP events) of the default dispatch with / without
exceptions and of variant 8 (no add-back), for each MI355Q_V8_PHASED setting, interleaved round-robin."""
import sys, os
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device('cuda:0')
M = N = K = 4096
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
_, xm, xe = ops.block_fp_quantize(x, 6, 8, 127, [1, 16], True, want_fake=False, want_packed=True)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
xa = ops.bfp_align_rows(xm, xe, 5, 127); wa = ops.bfp_align_rows(wm, we, 5, 127)
xe_, we_ = ops.bfp_align_rows(xm, xe, 5, 127), ops.bfp_align_rows(wm, we, 5, 127)
xe_.sparse.zero_(); we_.sparse.zero_()        # same operands, empty exception lists
print("exceptions", len(ops.row_list_entries(xa.sparse, M)[1]), len(ops.row_list_entries(wa.sparse, N)[1]))
y = torch.empty(M, N, device=dev)
phased = sys.argv[1:] or ["2", "1"]
cases = [(p, c) for p in phased for c in ("full", "empty", "v8")]
acc = {c: [] for c in cases}
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(6):
    for p, c in cases:
        os.environ["MI355Q_V8_PHASED"] = p
        if c == "v8":
            ops.set_gemm_variant(8)
            t = timed(lambda: ops.bfp_gemm_aligned(xa, wa, out=y))
            ops.set_gemm_variant(0)
        else:
            a_, b_ = (xa, wa) if c == "full" else (xe_, we_)
            for _ in range(3): ops.bfp_gemm_aligned(a_, b_, out=y)
            torch.cuda.synchronize()
            ops.gemm_timing(True)
            for _ in range(30): ops.bfp_gemm_aligned(a_, b_, out=y)
            torch.cuda.synchronize(); ops.gemm_timing(False)
            t = ops.gemm_timing_read()[1] * 1e3
        if rnd: acc[(p, c)].append(t)
for k, v in acc.items():
    v.sort(); print(f"phased={k[0]} {k[1]:6s} main kernel median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f} max {v[-1]:7.2f}")
