"""Pack + attention at head_dim 64 with the Q fragments packed in front (ops.attention_set_qpack(2)) against q quantised inside the
kernels (0): the shapes where the eight-key-wave kernel runs (T > 1024).  HIP events, us.  profiles/r06_attention_kw8_qf.jsonl holds the
run that decided the default (its first line of each pair: the eight-key-wave variant not yet allowed to load fragments)."""
import json, math, sys, os
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device("cuda:0")
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(e) / n * 1e3, 1)
par = (6, 8, 127, 6, 8, 127)
row = {}
for H, T, D in ((12, 2048, 64), (32, 2048, 64), (32, 1536, 64)):
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(1, H, T, D, generator=g).to(dev) for _ in range(3))
    for mode in (0, 2):
        prev = ops.attention_set_qpack(mode)
        row[f"[{H},{T},{D}] qpack={mode}"] = timed(lambda: ops.bfp_attention(q, k, v, par, par, causal=True, token_major=True))
        ops.attention_set_qpack(prev)
print(json.dumps(row))
