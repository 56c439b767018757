"""Time the row-aligned GEMM at 4096^3: default dispatch (with exception add-back) vs variant 8 (the product of
the rewritten operands only), interleaved round-robin so clock drift hits both alike."""
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
print("exceptions", len(ops.row_list_entries(xa.sparse, M)[1]), len(ops.row_list_entries(wa.sparse, N)[1]))
y = torch.empty(M, N, device=dev)
specs = [int(v) for v in sys.argv[1:]] or [8, 0]
acc = {s: [] for s in specs}
for rnd in range(6):
    for variant in specs:
        ops.set_gemm_variant(variant)
        for _ in range(3):
            ops.bfp_gemm_aligned(xa, wa, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.bfp_gemm_aligned(xa, wa, out=y)
        e1.record(); torch.cuda.synchronize()
        if rnd:
            acc[variant].append(e0.elapsed_time(e1) / 30 * 1e3)
ops.set_gemm_variant(0)
for variant in specs:
    v = sorted(acc[variant])
    print(f"variant {variant}: per call (all launches) median {v[len(v)//2]:8.2f} us  min {v[0]:8.2f}  max {v[-1]:8.2f}", flush=True)
# machinery without entries: empty lists
xa.sparse.zero_(); wa.sparse.zero_()
for _ in range(10): ops.bfp_gemm_aligned(xa, wa, out=y)
torch.cuda.synchronize()
ops.gemm_timing(True)
for _ in range(50): ops.bfp_gemm_aligned(xa, wa, out=y)
torch.cuda.synchronize(); ops.gemm_timing(False)
print("empty lists: main kernel", ops.gemm_timing_read())
