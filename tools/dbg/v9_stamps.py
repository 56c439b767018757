"""In-kernel phase stamps of the 256 x 256 tile GEMM (MI355Q_V9_STAMPS=1 selects the stamps build): after 150 ms of
un-stamped work (steady clocks) a few stamped launches; medians over the workgroups, wave 0 and wave 7."""
import os, sys, time, ctypes
os.environ["MI355Q_V9_STAMPS"] = "1"
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch, bench
from mi355q import ops, _lib
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False
dev = torch.device('cuda:0')
x, w, b = bench.make_inputs(torch, dev, 0)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
wa = ops.bfp_align_rows(wm, we, 5, 127); bq = ops.block_fp_quantize(b, 6, 8, 127, [16], False)
y = torch.empty(4096, 4096, device=dev)
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libmi355q.so"))
stamps = torch.zeros(256 * 2 * 8, dtype=torch.int64, device=dev)
xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
t_end = time.time() + 0.15
while time.time() < t_end:
    for _ in range(10): ops.bfp_gemm_aligned(xa, wa, bq, out=y)
    torch.cuda.synchronize()
lib.mi355q_debug_v9_stamps(ctypes.c_void_p(stamps.data_ptr()))
for rep in range(3):
    for _ in range(20): ops.bfp_gemm_aligned(xa, wa, bq, out=y)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(256, 2, 8).astype(np.int64)
    for wv, name in ((0, "wave 0"), (1, "wave 7")):
        t = s[:, wv, :6]
        d = np.diff(t, axis=1) * 0.01
        nent = s[:, wv, 7] >> 32; mode = s[:, wv, 7] & 0xff; r1 = ((s[:, wv, 7] >> 8) & 0xfff) * 0.01; r2 = ((s[:, wv, 7] >> 20) & 0xfff) * 0.01; nev = (s[:, wv, 7] >> 8) & 15; evc = (s[:, wv, 7] >> 12) & 0xfffff
        px = s[:, wv, 1]; bk = (px & 0xfff) * 0.01; st0 = ((px >> 12) & 0xfff) * 0.01; vec = ((px >> 24) & 0xfff) * 0.01
        cps = s[:, wv, 6] / 64.0
        clk = s[:, wv, 6] / np.maximum(t[:, 3] - t[:, 2], 1) * 100.0
        print(f"{name}: start->loop {np.median((t[:,2]-t[:,0])*0.01):5.2f}  loop {np.median(d[:,2]):6.2f} (max {d[:,2].max():6.2f})  "
              f"behind the loop {np.median(d[:,3]):5.2f} (max {d[:,3].max():5.2f}) [bookkeeping {np.median(bk):.2f}, keys+masks {np.median(st0):.2f}, service {np.median(vec):.2f} max {vec.max():.2f} = round 1 {np.median(r1):.2f} + round 2 {np.median(r2):.2f} (max {r2.max():.2f}) + fold]  epilogue {np.median(d[:,4]):5.2f} (max {d[:,4].max():5.2f}) us | "
              f"epi pct 10/50/90/99 {np.percentile(d[:,4],10):.1f}/{np.percentile(d[:,4],50):.1f}/{np.percentile(d[:,4],90):.1f}/{np.percentile(d[:,4],99):.1f} | {np.median(clk):5.0f} MHz {np.median(cps):7.1f} clk/K-step | span {(t[:,5].max()-t[:,0].min())*0.01:6.2f} us | entries med {np.median(nent):.0f} max {nent.max()} modes {np.bincount(mode.astype(int))}")
