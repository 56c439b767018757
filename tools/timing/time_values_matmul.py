#!/usr/bin/env python3
"""The second attention product (probabilities [32, 2048, 2048] x values [32, 2048, 128], Llama-7B shapes) under
matmul_block_minifloat / matmul_block_log: operands as bf16 + bf16-MFMA product against fp32 fake-quantised tensors + the
library fp32 GEMM (config["mi355q_values_matmul"] = "fp32"), and the block_fp fused kernel for scale.  us per call."""
import json
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
import mi355q.quantize as Q

dev = torch.device("cuda:0")
H, T, hd = 32, 2048, 128
g = torch.Generator().manual_seed(0)
p = torch.softmax(torch.randn(H, T, T, generator=g).to(dev) * 3 + torch.full((T, T), float("-inf"), device=dev).triu(1), dim=-1)
v = torch.randn(H, T, hd, generator=g).to(dev)
qq = torch.randn(H, T, hd, generator=g).to(dev)
kt = torch.randn(H, hd, T, generator=g).to(dev)
cfgs = {"block_minifloat": dict(name="block_minifloat", bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias_width=8,
                                data_in_block_size=[1, 16], weight_width=8, weight_exponent_width=4, weight_exponent_bias_width=8,
                                weight_block_size=[1, 16]),
        "block_log": dict(name="block_log", bypass=False, data_in_width=8, data_in_exponent_bias_width=8, data_in_block_size=[1, 16],
                          weight_width=8, weight_exponent_bias_width=8, weight_block_size=[1, 16]),
        "block_fp": dict(name="block_fp", bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
                         data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
                         weight_block_size=[1, 16])}


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


for name, cfg in cfgs.items():
    f = Q.get_quantized_func("bmm", cfg)
    rec = {"arith": name, "shape": "p[32,2048,2048] x v[32,2048,128] ; q[32,2048,128] x k^T[32,128,2048]"}
    # routes: "fused" the library's own product kernels (round 4: block_minifloat and block_log too); "bf16" / "bf16_split" the
    # quantisers' bf16 output + the vendor's bf16 batched GEMM (round 3); "fp32" fake-quantised fp32 tensors + the vendor's fp32 GEMM
    routes = {"block_fp": ("fused",), "block_minifloat": ("fused", "fp32"), "block_log": ("fused", "fp32")}[name]
    for route in routes:
        c = dict(cfg, mi355q_values_matmul=route)
        rec[f"pv_{route}_us"] = round(t(lambda: f(p, v, c)), 1)
        rec[f"qk_{route}_us"] = round(t(lambda: f(qq, kt, c)), 1)
    if name == "block_minifloat":
        fs = Q.get_quantized_func("softmax_bmm", cfg)
        sc = torch.randn(H, T, T, generator=torch.Generator().manual_seed(1)).to(dev) * 3
        rec["softmax_pv_fused_us"] = round(t(lambda: fs(sc, v, dict(cfg), causal=True)), 1)
    print(json.dumps(rec), flush=True)
