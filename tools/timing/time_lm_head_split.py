"""The unquantised lm_head (reference: nn.Linear in fp32, modeling_llama.py:758 / modeling_opt.py:915) as an fp32-equivalent product on
the bf16 MFMA: each fp32 operand split into three bf16 parts (h + m + l carry 24 significand bits), the six part products of weight
>= 2^-16 laid side by side along K -- ONE launch of the tile GEMM's bf16 arithmetic over K' = 6 K, fp32 accumulation, smallest terms
first.  Against torch's fp32 GEMM (the vendor library) at Llama-7B's [2048, 4096] x [32000, 4096]: time, and the error of both against
an fp64 product on sampled rows."""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import torch
from mi355q import ops
dev = torch.device("cuda:0")
M, N, K = 2048, int(os.environ.get("LM_N", 32000)), 4096
g = torch.Generator().manual_seed(3)
x = torch.randn(M, K, generator=g).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)


def split3(a):
    h = a.bfloat16().float(); r = a - h
    m = r.bfloat16().float(); l = (r - m).bfloat16().float()
    return h, m, l


def operand(a, order):
    parts = dict(zip("hml", split3(a)))
    return ops.bf16_tile(torch.cat([parts[c] for c in order], dim=1).contiguous())


# pairs, smallest first: (m,m) (l,h) (h,l) (m,h) (h,m) (h,h)
XO, WO = "mlhmhh", "mhlhmh"
wt = operand(w, WO)
out = torch.empty(M, N, device=dev)


def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(e) / n * 1e3, 1)


def split_gemm():
    xt = operand(x, XO)
    return ops.bf16_gemm_tiled(xt, wt, M, N, 6 * K, out=out)


row = {"shape": f"[{M}, {K}] x [{N}, {K}]^T fp32"}
row["vendor_fp32_us"] = t(lambda: torch.matmul(x, w.t()))
row["split6_bf16_us (operand build + GEMM)"] = t(split_gemm)
xt = operand(x, XO)
row["split6_bf16_gemm_only_us"] = t(lambda: ops.bf16_gemm_tiled(xt, wt, M, N, 6 * K, out=out))
XO3, WO3 = "lhh"[0:0] + "mhh", "hmh"
wt3 = operand(w, WO3); xt3 = operand(x, XO3)
row["split3_bf16_gemm_only_us"] = t(lambda: ops.bf16_gemm_tiled(xt3, wt3, M, N, 3 * K, out=out))
y3 = ops.bf16_gemm_tiled(xt3, wt3, M, N, 3 * K).clone()
y6 = split_gemm().clone(); yv = torch.matmul(x, w.t())
rows = torch.arange(0, M, 64, device=dev)
ref = x[rows].double() @ w.double().t()
den = ref.abs().mean()
for name, y in (("vendor_fp32", yv), ("split6", y6), ("split3", y3)):
    d = (y[rows].double() - ref).abs()
    row[name + "_err_mean_rel"] = float(d.mean() / den); row[name + "_err_max_rel"] = float(d.max() / den)
print(json.dumps(row))
if os.path.isdir(ROOT / "gpurun_out"):
    open(ROOT / "gpurun_out" / "r06_lm_head_split.json", "w").write(json.dumps(row) + "\n")
