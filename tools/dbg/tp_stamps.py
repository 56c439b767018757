"""Cycle counts inside the tile product kernel (mi355q_debug_tp_stamps): per workgroup, wave 0 and wave 7 -- cycles in the piece
loop, of which waiting for the own loads (s_waitcnt) and at the barrier.  python tools/dbg/tp_stamps.py [block_fp|block_minifloat|block_log]"""
import sys, ctypes, os
sys.path.insert(0, 'llm-mixed-q_amd')
import numpy as np, torch
from mi355q import ops, _lib
dev = torch.device('cuda:0')
H, T, hd = 32, 2048, 128
g = torch.Generator().manual_seed(0)
p = torch.softmax(torch.randn(H, T, T, generator=g).to(dev) * 3 + torch.full((T, T), float('-inf'), device=dev).triu(1), dim=-1)
v = torch.randn(H, T, hd, generator=g).to(dev); q = torch.randn(H, T, hd, generator=g).to(dev); kt = torch.randn(H, hd, T, generator=g).to(dev)
arith = sys.argv[1] if len(sys.argv) > 1 else "block_fp"
f = {"block_fp": lambda x, y: ops.bfp_matmul(x, y, 6, 8, 127, 6, 8, 127),
     "block_minifloat": lambda x, y: ops.values_matmul(x, y, "block_minifloat", (8, 4, 8), (8, 4, 8)),
     "block_log": lambda x, y: ops.values_matmul(x, y, "block_log", (8, 8))}[arith]
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libmi355q.so"))
nwg = H * (T // 128)
stamps = torch.zeros(nwg * 2 * 4, dtype=torch.int64, device=dev)
for tag, x, y in (("P V", p, v), ("Q K^T", q, kt)):
    for _ in range(5): f(x, y)
    torch.cuda.synchronize()
    lib.mi355q_debug_tp_stamps(ctypes.c_void_p(stamps.data_ptr()))
    f(x, y); torch.cuda.synchronize()
    lib.mi355q_debug_tp_stamps(ctypes.c_void_p(0))
    s = stamps.cpu().numpy().reshape(nwg, 2, 4).astype(np.float64)
    for w, name in ((0, "wave 0"), (1, "wave 7")):
        tot, wait, bar, n = s[:, w, 0], s[:, w, 1], s[:, w, 2], s[:, w, 3]
        print(f"{arith} {tag} {name}: loop {np.median(tot):9.0f} cycles (s_memtime ticks) for {n[0]:.0f} pieces = {np.median(tot / n):7.0f} / piece; "
              f"waiting for loads {np.median(wait / n):7.0f}, at the barrier {np.median(bar / n):7.0f}, the rest {np.median((tot - wait - bar) / n):7.0f}")
