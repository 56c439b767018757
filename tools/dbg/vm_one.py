"""one attention product, a few calls (for rocprofv3 --kernel-trace --stats):  python3 tools/dbg/vm_one.py block_log qk"""
import sys
sys.path.insert(0, 'llm-mixed-q_amd')
import torch
from mi355q import ops
dev = torch.device('cuda:0')
H, T, hd = 32, 2048, 128
g = torch.Generator().manual_seed(0)
arith, which = sys.argv[1], sys.argv[2]
if which == "pv":
    x = torch.softmax(torch.randn(H, T, T, generator=g).to(dev) * 3 + torch.full((T, T), float('-inf'), device=dev).triu(1), dim=-1)
    y = torch.randn(H, T, hd, generator=g).to(dev)
else:
    x = torch.randn(H, T, hd, generator=g).to(dev); y = torch.randn(H, hd, T, generator=g).to(dev)
f = {"block_fp": lambda x, y: ops.bfp_matmul(x, y, 6, 8, 127, 6, 8, 127),
     "block_minifloat": lambda x, y: ops.values_matmul(x, y, "block_minifloat", (8, 4, 8), (8, 4, 8)),
     "block_log": lambda x, y: ops.values_matmul(x, y, "block_log", (8, 8))}[arith]
for _ in range(10): f(x, y)
torch.cuda.synchronize()
