import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/llm-mixed-q_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import test_gpu_gemm as T
from oracle import np_oracle as O
M, N, K, style = [int(v) if v.isdigit() else v for v in (sys.argv[1:5] if len(sys.argv) > 4 else ["256", "256", "256", "sparse"])]
x, w, b = T._inputs(M, N, K, 3000 + M + N + K, style)
cfg = T._cfg(6, 6)
y = T._run(x, w, b, cfg, aligned="rows", x_cap=0)
ref = O.bfp_linear_int(x, w, b, cfg)
print("counts (overflow words)", T._run.last_counts, "flags", T._run.last_flags)
bad = np.abs(y - ref) > 2e-6 * np.abs(ref).max() * max(1, K // 256)
print("bad", bad.sum(), "rows with bad", np.where(bad.any(1))[0][:40], "n", bad.any(1).sum(), "cols with bad", np.where(bad.any(0))[0][:40], "n", bad.any(0).sum())
from mi355q import ops
dev = torch.device("cuda:0")
xt = torch.from_numpy(x).to(dev)
_, xm, xe = ops.block_fp_quantize(xt, 6, 8, 127, [1, 16], True, want_fake=False, want_packed=True, fast_zero_blocks=True)
xa = ops.bfp_align_rows(xm, xe, 5, 127)
wt = torch.from_numpy(w).to(dev)
_, wm, we = ops.block_fp_quantize(wt, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
wa = ops.bfp_align_rows(wm, we, 5, 127)
torch.cuda.synchronize()
xl = xa.sparse.cpu().numpy(); wl = wa.sparse.cpu().numpy()
print("x list head", xl[:8], "bucket0 count", xl[8], "w bucket0 count", wl[8])
ents = xl[16:16 + 8 * min(int(xl[8]), 120)].reshape(-1, 8)
print("x entries rows", ents[:, 0][:60], "kb", ents[:, 1][:60])
ents = wl[16:16 + 8 * min(int(wl[8]), 120)].reshape(-1, 8)
print("w entries rows", ents[:, 0][:60], "kb", ents[:, 1][:60])
d = y - ref
for r in (32, 33, 38, 44, 47):
    print("row", r, "err[:6]", d[r, :6], "ref[:6]", ref[r, :6], "ratio", (y[r, :4] / ref[r, :4]))
print("err row 38 - err row 33:", (d[38] - d[33])[:6], " err row 44 - err row 33:", (d[44] - d[33])[:6])
