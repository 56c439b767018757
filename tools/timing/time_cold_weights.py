"""Does a layer's GEMM run slower in the model than alone because its weights are COLD?  A timing loop over one GEMM re-reads the same
operands, which then sit in the 256 MB memory-side cache; a 32-layer forward streams 26 GB of weights through it.  Llama-7B's two bf16-route
GEMM shapes at 2048 tokens, HIP events around every launch, us (median of 40):
  warm     the same weight operand every call
  cold     16 different weight operands in turn (0.5 - 1.4 GB: each call's weights come from HBM)
  touched  cold, but a streaming read of the weights (torch sum) runs right in front of the timed launch"""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import torch
from mi355q import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
NW = 16


def med(xs):
    return round(sorted(xs)[len(xs) // 2], 1)


def run(label, M, N, K, bf16):
    x = torch.randn(M, K, generator=g).to(dev)
    ws = [(torch.randn(N, K, generator=g) * 0.02).to(dev) for _ in range(2)]
    if bf16:
        xt = ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127)
        base = [ops.block_fp_quantize_bf16_tiled(w, 6, 8, 127) for w in ws]
        wops = [base[i % 2].clone() for i in range(NW)]
        call = lambda w, out: ops.bf16_gemm_tiled(xt, w, M, N, K, out=out)
        touch = lambda w: w.view(torch.int32).sum()
    else:
        xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
        wops = []
        for i in range(NW):
            _, wm, we = ops.block_fp_quantize(ws[i % 2] * (1 + 0.01 * i), 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
            wops.append(ops.bfp_align_rows(wm, we, 5, 127))
        call = lambda w, out: ops.bfp_gemm_aligned(xa, w, None, out=out)
        touch = lambda w: w.mant.view(torch.int32).sum()
    out = torch.empty(M, N, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    res = {}
    for mode in ("warm", "cold", "touched"):
        ts = []
        for i in range(48):
            w = wops[0] if mode == "warm" else wops[i % NW]
            if mode == "touched":
                touch(w)
            a, e = ev(), ev()
            a.record(); call(w, out); e.record()
            torch.cuda.synchronize()
            if i >= 8:
                ts.append(a.elapsed_time(e) * 1e3)
        res[mode + "_us"] = med(ts)
    print(json.dumps({"gemm": label, "shape": f"{M} x {N} x {K}", "route": "bf16 per-block" if bf16 else "int8 rows", **res}), flush=True)


z = torch.randn(4096, 4096, device=dev)
for _ in range(30): z @ z
run("o_proj", 2048, 4096, 4096, True)
run("down_proj", 2048, 4096, 11008, True)
