"""Time GEMM kernel variants at 4096^3 with HIP events (torch events on the current stream).
argv: list of 'variant[:envname=val]' e.g. 6 8:MI355Q_V8_CFG=1"""
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
for spec in sys.argv[1:]:
    parts = spec.split(':')
    variant = int(parts[0])
    for kv in parts[1:]:
        k, v = kv.split('=')
        os.environ[k] = v
    ops.set_gemm_variant(variant)
    for _ in range(10):
        ops.bfp_gemm_aligned(xa, wa, out=y)
    torch.cuda.synchronize()
    best = 1e9; tot = 0
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.bfp_gemm_aligned(xa, wa, out=y)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        best = min(best, t); tot += t
    print(f"{spec:32s} avg {tot/5:8.2f} us  best {best:8.2f} us  -> {2*M*N*K/best/1e6:8.1f} TOPS", flush=True)
