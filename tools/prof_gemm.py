"""Run the aligned GEMM a few times at 4096^3 W6A6 (for rocprofv3 passes)."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
M = N = K = 4096
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
_, xm, xe = ops.block_fp_quantize(x, 6, 8, 127, [1, 16], True, want_fake=False, want_packed=True)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
xa = ops.bfp_align(xm, xe, 5, 127); wa = ops.bfp_align(wm, we, 5, 127)
y = torch.empty(M, N, device=dev)
ops.set_gemm_variant(variant)
for _ in range(iters):
    ops.bfp_gemm_aligned(xa, wa, out=y)
torch.cuda.synchronize()
