"""A few steady-state Linear steps at 4096^3 (for rocprofv3 passes). argv: variant, iters"""
import sys, os
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
M = N = K = 4096
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
if os.environ.get('EASY'):
    x = ((torch.rand(M, K, generator=g) * 0.5 + 0.5) * torch.sign(torch.randn(M, K, generator=g))).to(dev)
    w = ((torch.rand(N, K, generator=g) * 0.5 + 0.5) * torch.sign(torch.randn(N, K, generator=g)) * 0.02).to(dev)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
wa = ops.bfp_align(wm, we, 5, 127)
y = torch.empty(M, N, device=dev)
ops.set_gemm_variant(variant)
for _ in range(iters):
    xa = ops.block_fp_quantize_aligned(x, 6, 8, 127)
    ops.bfp_gemm_aligned(xa, wa, out=y)
torch.cuda.synchronize()
