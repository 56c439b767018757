import sys, time
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from mi355q import ops
from oracle import np_oracle as O
dev = torch.device('cuda:0')
# correctness of v3 alone on fully alignable data (small): use W4 (4 spare bits) gaussian
r = np.random.default_rng(0)
M,N,K = 256, 384, 1024
x = r.normal(size=(M,K)).astype(np.float32); w = (r.normal(size=(N,K))*0.02).astype(np.float32)
cfg = dict(name="block_fp", data_in_width=4, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1,16],
           weight_width=4, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1,16])
_, xm, xe = ops.block_fp_quantize(torch.from_numpy(x).to(dev), 4, 8, 127, [1,16], True, want_fake=False, want_packed=True)
_, wm, we = ops.block_fp_quantize(torch.from_numpy(w).to(dev), 4, 8, 127, [1,16], False, want_fake=False, want_packed=True)
xa = ops.bfp_align(xm, xe, 3, 127); wa = ops.bfp_align(wm, we, 3, 127)
print("flags", xa.rowflag.float().mean().item(), wa.rowflag.float().mean().item(), xa.sparse[0].item(), wa.sparse[0].item())
ref = O.bfp_linear_int(x, w, None, cfg)
for var in (6, 7, 0):
    ops.set_gemm_variant(var)
    yv = ops.bfp_gemm_aligned(xa, wa).cpu().numpy()
    print("variant", var, "err", np.abs(yv-ref).max()/np.abs(ref).max())
# perf at 4096^3 W6
M=N=K=4096
g = torch.Generator().manual_seed(0)
x = (torch.randn(M,K,generator=g)*torch.exp(torch.randn(M,1,generator=g))).to(dev); w = (torch.randn(N,K,generator=g)*0.02).to(dev)
_, xm, xe = ops.block_fp_quantize(x, 6, 8, 127, [1,16], True, want_fake=False, want_packed=True)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1,16], False, want_fake=False, want_packed=True)
xa = ops.bfp_align(xm, xe, 5, 127); wa = ops.bfp_align(wm, we, 5, 127)
print("4096 flags", xa.rowflag.float().mean().item(), wa.rowflag.float().mean().item(), "unaligned row-groups", xa.sparse[0].item(), wa.sparse[0].item())
y = torch.empty(M,N,device=dev)
for var in (6, 7, 6, 7):
    ops.set_gemm_variant(var)
    for _ in range(5): ops.bfp_gemm_aligned(xa, wa, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.bfp_gemm_aligned(xa, wa, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/50
    print(f"variant {var}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.0f} TOPS")
