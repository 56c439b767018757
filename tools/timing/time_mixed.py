"""The mixed contraction's two launches at bench.py's `robustness` operands (4096^3, K / 64 outlier channels x 60), each alone and
as the step: class-aware quantiser, mixed tile GEMM; beside them the per-block bf16 route's two launches and the headline's.
    python tools/timing/time_mixed.py            -> JSON line (also appended to gpurun_out/r06_mixed.jsonl)"""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import torch
import bench
from mi355q import ops
import mi355q.quantize as Q
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device("cuda:0")
M = N = K = 4096
g = torch.Generator().manual_seed(11)
x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
idx = torch.randint(0, K, (K // 64,), generator=g)
x[:, idx] *= 60.0
x = x.to(dev)
_, w, b = bench.make_inputs(torch, "cpu", 0)
cfg = dict(bench.CFG)
fp = torch.nn.Linear(K, N)
with torch.no_grad():
    fp.weight.copy_(w); fp.bias.copy_(b)
lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to(dev)
with torch.no_grad():
    for _ in range(3):
        y = lin(x)
assert lin._mixed is not None
m = lin._mixed
cls = m["classes"]


def t(fn, n=100, rounds=3):
    best = 1e9
    for _ in range(10):
        fn()
    for _ in range(rounds):
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / n * 1e3)
    return round(best, 2)


x0, x1 = ops.block_fp_quantize_classes(x, cls, 6, 8, 127)
out = torch.empty(M, N, device=dev)
row = {"shape": "4096^3 W6A6, K / 64 outlier channels x 60", "K0": cls.K0, "K1": cls.K1}
row["class_quantiser_us"] = t(lambda: ops.block_fp_quantize_classes(x, cls, 6, 8, 127))
row["mixed_gemm_us"] = t(lambda: ops.bfp_gemm_mixed(x0, m["wa0"], x1, m["w1"], cls.K1, lin.bias, out=out))
with torch.no_grad():
    row["mixed_step_module_us"] = t(lambda: lin(x))
# the per-block bf16 route on the same operands
cfg2 = dict(cfg, mi355q_mixed=False)
lin2 = Q.get_quantized_cls("linear", cfg2).from_float(fp, cfg2).to(dev)
with torch.no_grad():
    for _ in range(3):
        lin2(x)
    assert lin2._uses_bf16_route()
    xt = ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127)
    row["bf16_quantiser_us"] = t(lambda: ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127))
    wt = lin2._bf16_weight_operand(dev)
    row["bf16_gemm_us"] = t(lambda: ops.bf16_gemm_tiled(xt, wt, M, N, K, lin2.bias, out=out))
    row["bf16_step_module_us"] = t(lambda: lin2(x))
row["mixed_TFLOPs"] = round(2.0 * M * N * K / row["mixed_step_module_us"] / 1e6, 1)
row["bf16_TFLOPs"] = round(2.0 * M * N * K / row["bf16_step_module_us"] / 1e6, 1)
print(json.dumps(row))
if os.path.isdir(ROOT / "gpurun_out"):
    with open(ROOT / "gpurun_out" / "r06_mixed.jsonl", "a") as f:
        f.write(json.dumps(row) + "\n")
