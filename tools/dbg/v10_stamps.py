"""In-kernel phase stamps of the small-tile GEMM (library built with -DV10_STAMPS: `make -C llm-mixed-q_amd/csrc EXTRA=-DV10_STAMPS`):
medians over the workgroups, x 10 ns -> us.   python tools/dbg/v10_stamps.py [M N K [width]]"""
import os, sys, time, ctypes
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from mi355q import ops, _lib
dev = torch.device('cuda:0')
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 512, 4096)
width = int(sys.argv[4]) if len(sys.argv) > 4 else 6
g = torch.Generator().manual_seed(1)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
_, wm, we = ops.block_fp_quantize(w, width, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
wa = ops.bfp_align_rows(wm, we, width - 1, 127)
y = torch.empty(M, N, device=dev)
xa = ops.block_fp_quantize_aligned_rows(x, width, 8, 127)
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libmi355q.so"))
nwg = 2048
stamps = torch.zeros(nwg * 2 * 8, dtype=torch.int64, device=dev)
t_end = time.time() + 0.15
while time.time() < t_end:
    for _ in range(10): ops.bfp_gemm_aligned(xa, wa, None, out=y)
    torch.cuda.synchronize()
lib.mi355q_debug_v10_stamps(ctypes.c_void_p(stamps.data_ptr()))
names = ["issue", "buckets+bookkeeping+gathers", "wait stage 0", "K loop", "split/exceptions", "epilogue"]
for rep in range(3):
    stamps.zero_()
    for _ in range(20): ops.bfp_gemm_aligned(xa, wa, None, out=y)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nwg, 2, 8).astype(np.int64)
    s = s[s[:, 0, 0] != 0]
    for wv in (0, 1):
        t = s[:, wv, :7]
        d = np.diff(t, axis=1) * 0.01
        nent = s[:, wv, 7] >> 32; nlive = (s[:, wv, 7] >> 8) & 0xffff; mode = s[:, wv, 7] & 0xff
        print(f"wave {'0' if wv == 0 else 'last'} ({len(s)} workgroups): " + "  ".join(f"{n} {np.median(d[:, i]):.2f} (max {d[:, i].max():.2f})" for i, n in enumerate(names)) +
              f" | whole {np.median(t[:, 6] - t[:, 0]) * 0.01:.2f}  span {(t[:, 6].max() - t[:, 0].min()) * 0.01:.2f} us | entries med {np.median(nent):.0f} live {np.median(nlive):.0f} max {nent.max()} modes {np.bincount(mode.astype(int))}")
