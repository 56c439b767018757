"""Time the fused activation kernels at 4096 x 4096 (HIP events over 50 calls, interleaved)."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = (torch.randn(4096, 4096, generator=g) * torch.exp(torch.randn(4096, 1, generator=g))).to(dev)
fns = {"rows": lambda: ops.block_fp_quantize_aligned_rows(x, 6, 8, 127),
       "bf16": lambda: ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127)}
acc = {k: [] for k in fns}
for rnd in range(5):
    for k, fn in fns.items():
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        acc[k].append(e0.elapsed_time(e1) / 50 * 1e3)
for k, v in acc.items():
    v.sort(); print(f"{k:8s} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}  ({81.0e6 / v[len(v)//2] / 1e6:.2f} TB/s of 81 MB)")
