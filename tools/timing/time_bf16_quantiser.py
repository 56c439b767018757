"""The per-block route's activation quantiser (x fp32 -> tiled bf16) at the model shapes: plain, relu / silu_mul / rmsnorm in front.  HIP events."""
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
    return round(a.elapsed_time(e) / n * 1e3, 1)
for M, K in ((2048, 4096), (2048, 11008), (2048, 8192), (4096, 4096)):
    x, u = torch.randn(M, K, device=dev), torch.randn(M, K, device=dev)
    w = torch.rand(K, device=dev) + 0.5
    row = {"M": M, "K": K}
    row["plain_us"] = t(lambda: ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127))
    row["relu_us"] = t(lambda: ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127, pre=("relu", None)))
    row["silu_mul_us"] = t(lambda: ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127, pre=("silu_mul", u)))
    row["rmsnorm_us"] = t(lambda: ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127, pre=("rmsnorm", w, 1e-6)))
    print(json.dumps(row), flush=True)
