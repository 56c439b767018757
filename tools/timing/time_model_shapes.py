"""The reference models' Linear shapes at 2048 tokens through the GEMM launch alone (pre-quantised operands, HIP events):
kernel-level time of the row-scale int8 tile GEMM per shape, to see what its fixed costs weigh at real sizes."""
import json, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device("cuda:0")
shapes = [("OPT-125m q/k/v/out", 2048, 768, 768), ("OPT-125m fc1", 2048, 768, 3072), ("OPT-1.3B q/k/v/out", 2048, 2048, 2048),
          ("OPT-1.3B fc1", 2048, 2048, 8192), ("Llama-7B q/k/v/o", 2048, 4096, 4096), ("Llama-7B gate/up", 2048, 4096, 11008),
          ("bench", 4096, 4096, 4096)]
for name, M, K, N in shapes:
    g = torch.Generator().manual_seed(M + K + N)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    b = (torch.randn(N, generator=g) * 0.02).to(dev)
    _, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
    out = torch.empty(M, N, device=dev)

    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / n * 1e3

    tg = timed(lambda: ops.bfp_gemm_aligned(xa, wa, b, out=out))
    tq = timed(lambda: ops.block_fp_quantize_aligned_rows(x, 6, 8, 127))
    print(json.dumps({"layer": name, "M": M, "K": K, "N": N, "gemm_us": round(tg, 1), "gemm_TOPS": round(2.0 * M * N * K / tg / 1e6),
                      "quantise_us": round(tq, 1), "overflow": int(xa.sparse[0])}), flush=True)
