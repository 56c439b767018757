"""The per-rank pieces of the sharded 4096^3 layer under gather = "quantised" (DESIGN 6), on one GPU: the consumer's quantiser on
one rank's output slice [4096, 4096 / P] (with the relu in front) and the bf16 tile GEMM of the NEXT layer's shard
(4096 x 4096 / P x 4096) with x in P column segments.  us per call."""
import json, sys; sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device('cuda:0')
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
M = K = 4096
g = torch.Generator().manual_seed(0)
y_full = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
for P in (1, 2, 4, 8):
    N = 4096 // P
    w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    wt = ops.block_fp_quantize_bf16_tiled(w, 6, 8, 127, reuse=False)
    sl = y_full[:, :K // P].contiguous()
    tq = t(lambda: ops.block_fp_quantize_bf16_tiled(sl, 6, 8, 127, pre=("relu", None)))
    segs = torch.stack([ops.block_fp_quantize_bf16_tiled(y_full[:, s * K // P:(s + 1) * K // P].contiguous(), 6, 8, 127, reuse=False,
                                                         pre=("relu", None)).reshape(-1) for s in range(P)]).contiguous()
    out = torch.empty(M, N, device=dev)
    tg = t(lambda: ops.bf16_gemm_tiled(segs if P > 1 else segs[0], wt, M, N, K, None, out=out, segments=P))
    print(json.dumps({"P": P, "quantise_own_slice_us": round(tq, 1), "bf16_shard_gemm_us": round(tg, 1),
                      "gathered_MiB_per_rank": round((P - 1) / P * M * K * 2 / 2**20, 1)}), flush=True)
