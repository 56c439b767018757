"""W4A4 / W5A5 4096^3 step on the MX scaled MFMA (quantiser + product) against the int8 row-scale route, HIP events.
    python tools/timing/time_mx.py [M N K [width]]"""
import json, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 4096)
width = int(sys.argv[4]) if len(sys.argv) > 4 else 4
def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
g = torch.Generator().manual_seed(1)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
z = torch.randn(4096, 4096, device=dev)
for _ in range(40): z @ z                      # (clock ramp)
wq = ops.block_fp_quantize(w, width, 8, 127, [1, 16], False)
wop = ops.block_fp_quantize_mx(w, width, 8, 127, reuse=False)
y = torch.empty(M, N, device=dev)
xop = ops.block_fp_quantize_mx(x, width, 8, 127)
row = {"M": M, "N": N, "K": K, "width": width, "x_flag": int(xop.bad[0]), "w_flag": int(wop.bad[0])}
row["mx_quantise_us"] = round(t(lambda: ops.block_fp_quantize_mx(x, width, 8, 127)), 1)
row["mx_gemm_us"] = round(t(lambda: ops.mx_gemm(xop, wop, wq, None, out=y)), 1)
def step_mx():
    ops.mx_gemm(ops.block_fp_quantize_mx(x, width, 8, 127), wop, wq, None, out=y)
row["mx_step_us"] = round(t(step_mx), 1)
_, wm, we = ops.block_fp_quantize(w, width, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
wa = ops.bfp_align_rows(wm, we, width - 1, 127)
xa = ops.block_fp_quantize_aligned_rows(x, width, 8, 127)
row["int8_quantise_us"] = round(t(lambda: ops.block_fp_quantize_aligned_rows(x, width, 8, 127)), 1)
row["int8_gemm_us"] = round(t(lambda: ops.bfp_gemm_aligned(xa, wa, None, out=y)), 1)
def step_i8():
    ops.bfp_gemm_aligned(ops.block_fp_quantize_aligned_rows(x, width, 8, 127), wa, None, out=y)
row["int8_step_us"] = round(t(step_i8), 1)
row["step_speedup"] = round(row["int8_step_us"] / row["mx_step_us"], 3)
row["mx_gemm_TOPS"] = round(2.0 * M * N * K / row["mx_gemm_us"] / 1e6)
print(json.dumps(row))
