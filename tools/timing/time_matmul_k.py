"""kernel-only timing of the fused product (HIP events around repeated ops.bfp_matmul calls, large batch to hide host)"""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for xs, ys in (((12, 2048, 2048), (12, 2048, 64)), ((12, 2048, 64), (12, 64, 2048)), ((32, 2048, 2048), (32, 2048, 128))):
    x = torch.rand(xs, device=dev); y = torch.randn(ys, device=dev)
    for _ in range(3): ops.bfp_matmul(x, y, 6, 8, 127, 6, 8, 127)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.bfp_matmul(x, y, 6, 8, 127, 6, 8, 127)
    b.record(); torch.cuda.synchronize()
    print(xs, ys, round(a.elapsed_time(b) / 20 * 1e3, 1), "us per call")
