"""Is the first GPU case after a multi-threaded CPU workload host-bound?  (bench.py's config 5 follows the CPU baseline.)"""
import sys, time; sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device('cuda:0')
x = torch.randn(2048, 4096, generator=torch.Generator().manual_seed(7)).to(dev) * 4
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(n): fn()
    e.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return round(a.elapsed_time(e) / n * 1e3, 1), round((t1 - t0) / n * 1e6, 1)
f = lambda: ops.block_fp_quantize(x, 6, 8, 127, [1, 16], True)
print("fresh (gpu us, host enqueue us):", t(f), t(f))
a = torch.randn(4096, 4096); b = torch.randn(4096, 4096)
t0 = time.time()
while time.time() - t0 < 3.0: (a @ b).sum()
print("after 3 s of CPU matmul on", torch.get_num_threads(), "threads:", t(f), t(f), t(f))
time.sleep(1.0)
print("one second later:", t(f))
