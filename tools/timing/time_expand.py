#!/usr/bin/env python3
"""The expand pass of width-bit weight storage alone (mi355q_bfp_expand through PackedWeights.expand): us per call and GB/s of its
own traffic (packed mantissas + code bytes in; tiled int8 + exponent bytes, or tiled bf16, out) at the Llama-7B weight shapes."""
import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
import mi355q.quantize as Q
dev = torch.device("cuda:0")
base = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
            bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16], mi355q_weight_storage="packed")
for name, (K, N, bf16) in {"q_proj 4096x4096 (int8 rows)": (4096, 4096, False), "gate_proj 11008x4096 (int8 rows)": (4096, 11008, False),
                           "o_proj 4096x4096 (bf16)": (4096, 4096, True), "down_proj 4096x11008 (bf16)": (11008, 4096, True)}.items():
    torch.manual_seed(0)
    cfg = dict(base, mi355q_align="blocks" if bf16 else "rows")
    lin = Q.get_quantized_cls("linear", cfg).from_float(torch.nn.Linear(K, N), cfg).to(dev)
    x = torch.randn(256, K, device=dev)
    with torch.no_grad():
        lin(x); lin(x)
    pw = lin._w_packed
    for _ in range(10):
        pw.expand()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        pw.expand()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 50 * 1e3
    vals = K * N
    traffic = vals * 6 / 8 + vals / 16 + (vals * 2 if not pw.row_scale_flavour else vals + vals / 16)
    print(json.dumps({"weight": name, "flavour": "int8 rows" if pw.row_scale_flavour else "bf16", "us": round(us, 1), "MB": round(traffic / 1e6, 1),
                      "GB/s": round(traffic / us / 1e3, 0)}))
