"""The benchmark step (bench.py's layer and inputs) eager vs replayed from a HIP graph holding 10 steps: what the
launch path costs the step when the GPU is the bottleneck.  Run on the GPU box from the repo root."""
import json, sys, time
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import bench
from mi355q.graphs import GraphedForward
from mi355q.quantize import get_quantized_cls
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)

dev = torch.device("cuda:0")
x, w, b = bench.make_inputs(torch, dev, 0)
lin = get_quantized_cls("linear", bench.CFG)(bench.K, bench.N, bias=True, config=dict(bench.CFG)).to(dev)
with torch.no_grad():
    lin.weight.copy_(w); lin.bias.copy_(b)
    ref = lin(x).clone()


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    t_eager = timed(lambda: lin(x), 200) * 1e6


def ten(t):
    y = None
    for _ in range(10):
        y = lin(t)
    return y


fwd = GraphedForward(ten, (x,))
same = bool(torch.equal(fwd(x), ref))
t_graph = timed(lambda: fwd.graph.replay(), 20) / 10 * 1e6
print(json.dumps({"step": "bench.py layer 4096^3 W6A6", "eager_us": round(t_eager, 2), "graph_replay_us_per_step": round(t_graph, 2),
                  "graph_equals_eager": same}))
