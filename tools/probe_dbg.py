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
xa = ops.bfp_align(xm, xe, 5, 127); wa = ops.bfp_align(wm, we, 5, 127)
y = torch.empty(M, N, device=dev)
ops.set_gemm_variant(3)
for rnd in range(2):
    for d in (0, 1, 2, 3, 4, 5, 6):
        os.environ["MI355Q_V3_DBG"] = str(d)
        for _ in range(5): ops.bfp_gemm_aligned(xa, wa, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.bfp_gemm_aligned(xa, wa, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(f"dbg {d} ({['normal','no global loads','no MFMA','no rescale','no barrier','no ds_read','MFMA only'][d]}): {ms*1e3:.1f} us")
