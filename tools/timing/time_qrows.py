"""Time the fused activation quantiser (quantise + pack + row-align + tile) alone, HIP events, at the benchmark size."""
import sys; sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch, bench
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device('cuda:0')
x, w, b = bench.make_inputs(torch, dev, 0)
for _ in range(10): ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
e.record(); torch.cuda.synchronize()
print(f"quantise+align rows: {a.elapsed_time(e) * 10:.2f} us per call")
