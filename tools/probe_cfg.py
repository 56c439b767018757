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
var = int(sys.argv[1]); env = sys.argv[2]; cfgs = [int(c) for c in sys.argv[3].split(",")]
ops.set_gemm_variant(var)
res = {c: [] for c in cfgs}
for rnd in range(4):
    for c in cfgs:
        os.environ[env] = str(c)
        for _ in range(3): ops.bfp_gemm_aligned(xa, wa, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.bfp_gemm_aligned(xa, wa, out=y)
        e1.record(); torch.cuda.synchronize()
        res[c].append(e0.elapsed_time(e1) / 30 * 1e3)
for c in cfgs:
    r = sorted(res[c][1:])
    print(f"{env}={c}: median {r[len(r)//2]:.1f} us  min {r[0]:.1f} us  ({2*M*N*K/r[len(r)//2]/1e6:.0f} TOPS)")
