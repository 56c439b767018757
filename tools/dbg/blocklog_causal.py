"""block_log on the causal attention-probability tensor of BASELINE config 5 ([32, 2048, 2048]): N timed calls, for
rocprofv3 --kernel-trace --stats (per-kernel time of the two passes)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
from mi355q import ops

shp = (32, 2048, 2048)
dev = torch.device("cuda:0")
x = torch.randn(*shp, generator=torch.Generator().manual_seed(7)).to(dev) * 4.0
p = torch.softmax(x + torch.full(shp[-2:], float("-inf"), device=dev).triu(1), dim=-1)
del x
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, fn in (("block_log", lambda: ops.block_log_quantize(p, 8, 8, [1, 16], True)),
                 ("block_fp", lambda: ops.block_fp_quantize(p, 6, 8, 127, [1, 16], True))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(name, "us per call", a.elapsed_time(e) / n * 1e3)
