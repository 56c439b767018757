import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
from mi355q import ops
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device("cuda:0")
g = lambda s: torch.Generator().manual_seed(s)
x = (torch.randn(4096, 4096, generator=g(0)) * torch.exp(torch.randn(4096, 1, generator=g(1)))).to(dev)
for _ in range(300):
    ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
torch.cuda.synchronize()
for rep in range(3):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300):
        ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
    e.record()
    torch.cuda.synchronize()
    print("aligned_rows quantiser us per call (back to back, incl. launch gaps)", a.elapsed_time(e) / 300 * 1e3)
