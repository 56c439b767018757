"""time the tile GEMM alone (events around 200 launches) under the current env; variant 8 = without exception lists"""
import os, sys, time
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch, bench
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False
dev = torch.device('cuda:0')
x, w, b = bench.make_inputs(torch, dev, 0)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
wa = ops.bfp_align_rows(wm, we, 5, 127); bq = ops.block_fp_quantize(b, 6, 8, 127, [16], False)
y = torch.empty(4096, 4096, device=dev)
xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
for variant in (0, 8):
    ops.set_gemm_variant(variant)
    for _ in range(600): ops.bfp_gemm_aligned(xa, wa, bq, out=y)
    torch.cuda.synchronize()
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): ops.bfp_gemm_aligned(xa, wa, bq, out=y)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 300
        print(f"variant {variant}: {us:6.2f} us per launch = {2*4096**3/us*1e-6:6.0f} TOPS (frac {2*4096**3/us*1e-6/5000:.3f})", flush=True)
ops.set_gemm_variant(0)
