import sys, os
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device('cuda:0')
M = N = K = 4096
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
wa = ops.bfp_align(wm, we, 5, 127)
y = torch.empty(M, N, device=dev)
for d in (0, 1, 2, 0, 1, 2):
    os.environ["MI355Q_FIX_DBG"] = str(d)
    for _ in range(5):
        xa = ops.block_fp_quantize_aligned(x, 6, 8, 127); ops.bfp_gemm_aligned(xa, wa, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        xa = ops.block_fp_quantize_aligned(x, 6, 8, 127); ops.bfp_gemm_aligned(xa, wa, out=y)
    e1.record(); torch.cuda.synchronize()
    print(f"fix mode {d} ({['all','x entries only','w entries only'][d]}): step {e0.elapsed_time(e1)/30*1e3:.1f} us")
