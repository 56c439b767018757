#!/usr/bin/env python3
"""A 4096 x 4096 W6A6 PTQ Linear, 4096 tokens, and Llama-7B's down_proj after SiLU (bf16 route): resident operand against
width-bit storage (the expand pass into the scratch operand in front of every GEMM).  us per forward, module level."""
import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
import mi355q.quantize as Q
from mi355q import ops
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device("cuda:0")
base = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
            bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16], mi355q_align="rows")
g = lambda s: torch.Generator().manual_seed(s)
for name, (K, N, M, act) in {"4096x4096 plain": (4096, 4096, 4096, False), "Llama down_proj 11008->4096 after SiLU (bf16 route)": (11008, 4096, 2048, True)}.items():
    torch.manual_seed(0)
    fp = torch.nn.Linear(K, N)
    with torch.no_grad():
        fp.weight.mul_(1.5)
    x = torch.randn(M, K, generator=g(0)) * torch.exp(torch.randn(M, 1, generator=g(1)))
    if act:
        x = torch.nn.functional.silu(x) * torch.randn(M, K, generator=g(2))
    x = x.to(dev)
    rec = {"layer": name}
    for tag, cfg in (("int8_resident", dict(base, mi355q_align="auto" if act else "rows")),
                     ("packed", dict(base, mi355q_align="auto" if act else "rows", mi355q_weight_storage="packed"))):
        lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to(dev)
        with torch.no_grad():
            for _ in range(30):
                y = lin(x)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(100):
                y = lin(x)
            e.record()
            torch.cuda.synchronize()
        rec[tag + "_us"] = round(a.elapsed_time(e) / 100 * 1e3, 1)
        rec[tag + "_bits"] = round(lin.weight_storage_bits(), 2)
        if tag == "int8_resident":
            ref = y.clone()
        else:
            rec[tag + "_max_rel_diff"] = float((y - ref).abs().max() / ref.abs().max())
    print(json.dumps(rec))
