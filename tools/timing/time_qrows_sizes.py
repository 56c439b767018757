"""The fused activation quantiser at K = 4096 over row counts (for rocprofv3 kernel stats: one kernel name per size is
not available, so sizes run in separate invocations: argv[1] = rows)."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
ops.REUSE_QUANTISED_INPUT = False
rows = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda:0')
x = (torch.randn(rows, K) * torch.exp(torch.randn(rows, 1))).to(dev)
for _ in range(10): ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
e.record(); torch.cuda.synchronize()
print(f"rows {rows} K {K}: {a.elapsed_time(e) * 10:.2f} us per call (events)")
