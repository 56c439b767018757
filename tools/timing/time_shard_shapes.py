"""The row-sharded layer's per-rank GEMM shapes on one GPU (north_star partition of the 4096^3 benchmark layer and of the
OPT-1.3B / Llama-7B layers over P ranks): steady-state step (fused quantise + tile GEMM, split-K where the grid is
under-filled), HIP events."""
import json, sys; sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device('cuda:0')
def t(fn, n=50, reps=3):
    """us per call: the best of `reps` timed runs of n calls (a box now and then stalls a stream for tens of ms: one such stall in a run
    of 50 short launches would be the whole figure)"""
    for _ in range(5): fn()
    best = float("inf")
    for _ in range(reps):
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / n * 1e3)
    return best
for name, M, N, K in (("bench P=8", 4096, 512, 4096), ("bench P=4", 4096, 1024, 4096), ("bench P=2", 4096, 2048, 4096),
                      ("OPT-1.3B fc1 P=8", 2048, 1024, 2048), ("OPT-1.3B q_proj P=8", 2048, 256, 2048), ("Llama-7B q_proj P=8", 2048, 512, 4096),
                      ("Llama-7B up P=8", 2048, 1376, 4096)):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    _, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    y = torch.empty(M, N, device=dev)
    xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
    tg = t(lambda: ops.bfp_gemm_aligned(xa, wa, None, out=y))
    tq = t(lambda: ops.block_fp_quantize_aligned_rows(x, 6, 8, 127))
    # the same launch 50 times in a HIP graph: launch to launch on the device, without the host's share of an eager call (a launch of
    # 15-25 us is as short as the Python + ctypes path that issues it)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): ops.bfp_gemm_aligned(xa, wa, None, out=y)
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(50): ops.bfp_gemm_aligned(xa, wa, None, out=y)
        tgg = t(gr.replay, n=10) / 50
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "gemm_us": round(tg, 1), "gemm_TOPS": round(2.0 * M * N * K / tg / 1e6),
                      "gemm_us_graph": round(tgg, 1), "gemm_TOPS_graph": round(2.0 * M * N * K / tgg / 1e6), "quantise_us": round(tq, 1)}))
