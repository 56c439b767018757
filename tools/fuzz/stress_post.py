"""Repeatability stress (race screen) of the row post-pass and the fused matmul: many re-quantisations of the same
input (the exception entries land in a different slot order every time) must give bit-identical results."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device("cuda:0")
r = np.random.default_rng(11)
for (M, N, K) in ((2048, 768, 3072), (600, 200, 1024), (4096, 1024, 4096)):
    x = np.maximum(r.normal(size=(M, K)), 0).astype(np.float32) * np.exp(r.normal(size=(M, 1))).astype(np.float32)
    x[:, ::7] *= np.float32(0.01)
    w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
    w[::9, 512:528] *= 300.0
    xt = torch.from_numpy(x).to(dev)
    _, wm, we = ops.block_fp_quantize(torch.from_numpy(w).to(dev), 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True,
                                      fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    ref, diff = None, 0
    for it in range(150):
        xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, bucket_cap=1016)
        y = ops.bfp_gemm_aligned(xa, wa, None)
        if ref is None:
            ref = y.clone(); over, full = ops.row_list_fill(xa.sparse, M, 1016)
        elif not torch.equal(ref, y):
            diff += 1
    print(f"post-pass {M}x{N}x{K}: overflow {over} fullest bucket {full}: {diff} differing runs of 149")
for (B, M, K, N) in ((12, 2048, 2048, 64), (12, 2048, 64, 2048), (4, 333, 400, 80)):
    x = torch.rand(B, M, K, device=dev); y = torch.randn(B, K, N, device=dev)
    ref, diff = None, 0
    for it in range(150):
        o = ops.bfp_matmul(x, y, 6, 8, 127, 6, 8, 127)
        if ref is None: ref = o.clone()
        elif not torch.equal(ref, o): diff += 1
    print(f"fused matmul {B}x{M}x{K}x{N}: {diff} differing runs of 149")
