"""Where the time of the attention products goes: the full call, and (run again with MI355Q_MATMUL_DBG=1) the call that stops
behind the y pack kernel [+ block_log's statistics pass], for the three block formats.  us per call."""
import sys, json, os
sys.path.insert(0, 'llm-mixed-q_amd')
import torch
from mi355q import ops
dev = torch.device('cuda:0')
H, T, hd = 32, 2048, 128
g = torch.Generator().manual_seed(0)
p = torch.softmax(torch.randn(H, T, T, generator=g).to(dev) * 3 + torch.full((T, T), float('-inf'), device=dev).triu(1), dim=-1)
v = torch.randn(H, T, hd, generator=g).to(dev); q = torch.randn(H, T, hd, generator=g).to(dev); kt = torch.randn(H, hd, T, generator=g).to(dev)


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(e) / n * 1e3, 1)


calls = {"block_fp": lambda x, y: ops.bfp_matmul(x, y, 6, 8, 127, 6, 8, 127),
         "block_minifloat": lambda x, y: ops.values_matmul(x, y, "block_minifloat", (8, 4, 8), (8, 4, 8)),
         "block_log": lambda x, y: ops.values_matmul(x, y, "block_log", (8, 8))}
for name, f in calls.items():
    rec = {"arith": name}
    for tag, x, y in (("pv", p, v), ("qk", q, kt)):
        rec[tag + ("_pack_only_us" if os.environ.get("MI355Q_MATMUL_DBG") == "1" else "_us")] = t(lambda: f(x, y))
    print(json.dumps(rec), flush=True)
