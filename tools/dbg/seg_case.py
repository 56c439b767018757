import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
from mi355q import ops
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device("cuda:0")
for rows, K, P, relu in ((256, 2048, 4, True), (256, 2048, 4, False), (256, 4096, 4, True), (300, 4096, 8, True), (256, 2048, 1, True)):
    g = torch.Generator().manual_seed(rows + K)
    x = (torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, 1, generator=g))).to(dev)
    seg = x.view(rows, P, K // P).permute(1, 0, 2).contiguous()
    pre = ("relu", None) if relu else None
    a = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127, pre=pre)
    ka = [t.clone() for t in (a.tiled, a.exp, a.rowflag, a.gscale)]
    b = ops.block_fp_quantize_aligned_rows(seg, 6, 8, 127, pre=pre, segments=True)
    kb = [b.tiled, b.exp, b.rowflag, b.gscale]
    c = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127, pre=pre)
    kc = [c.tiled, c.exp, c.rowflag, c.gscale]
    print(rows, K, P, relu, [bool(torch.equal(u, v)) for u, v in zip(ka, kb)], [bool(torch.equal(u, v)) for u, v in zip(ka, kc)],
          (ka[3] != kb[3]).sum().item(), ka[3][:4].tolist(), kb[3][:4].tolist())
