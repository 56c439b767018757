"""Race screen: the row-aligned GEMM repeated many times on the same operands must give bit-identical results
(any ordering hole between LDS-DMA landings, barriers and fragment reads shows up as an occasional different tile)."""
import sys, os
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device('cuda:0')
for (M, N, K, reps) in [(4096, 4096, 4096, 300), (2048, 4096, 4096, 300), (2048, 11008, 4096, 100), (2048, 4096, 11008, 100),
                        (300, 520, 1024, 300), (4096, 4096, 128, 300), (777, 1300, 2048, 200)]:
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    _, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
    for rows in ("256", "128"):
        os.environ["MI355Q_V8_TILE_ROWS"] = rows
        ref = ops.bfp_gemm_aligned(xa, wa).clone()
        bad = 0
        y = torch.empty_like(ref)
        for _ in range(reps):
            ops.bfp_gemm_aligned(xa, wa, out=y)
            bad += int(not torch.equal(ref, y))
        ops.set_gemm_variant(2)
        blk = ops.bfp_gemm_aligned(xa, wa)
        ops.set_gemm_variant(0)
        err = ((ref - blk).abs().max() / blk.abs().max()).item()
        print(f"M={M} N={N} K={K} tile rows {rows}: {reps} runs, {bad} differ; vs blockwise kernel rel err {err:.1e}", flush=True)
