"""A few steady-state row-aligned Linear steps at 4096^3 (for rocprofv3 passes). argv: iters"""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
import bench
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda:0')
x, w, b = bench.make_inputs(torch, dev, 0)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
wa = ops.bfp_align_rows(wm, we, 5, 127)
y = torch.empty(4096, 4096, device=dev)
for _ in range(iters):
    xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
    ops.bfp_gemm_aligned(xa, wa, out=y)
torch.cuda.synchronize()
