"""Randomised cross-checks of the two newest paths: fused quantise + matmul vs the two-quantisers route, and the
row post-pass (large exception buckets) vs the blockwise-exact kernel (variant 2) and the oracle."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import mi355q.quantize as Q
from mi355q import ops
from oracle import np_oracle as O
dev = torch.device("cuda:0")
r = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def cfg(wx, wy, fused):
    return dict(name="block_fp", bypass=False, data_in_width=wx, data_in_exponent_width=8, data_in_exponent_bias=127,
                data_in_block_size=[1, 16], weight_width=wy, weight_exponent_width=8, weight_exponent_bias=127,
                weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16],
                mi355q_fused_matmul=fused)
bad = 0
for it in range(150):
    B, M = int(r.integers(1, 5)), int(r.integers(1, 200))
    K, N = 16 * int(r.integers(1, 40)), 16 * int(r.integers(1, 20))
    wx, wy = int(r.integers(2, 10)), int(r.integers(2, 10))
    x = (r.normal(size=(B, M, K)) * np.exp(2 * r.normal(size=(B, M, 1)))).astype(np.float32)
    if it % 3 == 0: x = np.maximum(x, 0)
    if it % 5 == 0: x[:, :, : 16 * int(r.integers(0, K // 16 + 1))] = 0
    if it % 7 == 0: x *= np.float32(1e-9)
    y = (r.normal(size=(B, K, N)) * np.exp(r.normal(size=(B, 1, 1)))).astype(np.float32)
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    a = Q.get_quantized_func("bmm", cfg(wx, wy, True))(xt, yt, cfg(wx, wy, True)).cpu().numpy()
    b = Q.get_quantized_func("bmm", cfg(wx, wy, False))(xt, yt, cfg(wx, wy, False)).cpu().numpy()
    # (pass-through elements |x| <= 1e-8 enter the fused product rounded to bf16: <= 2e-11 each, times |y|, times K)
    tol = 3e-6 * (np.abs(b).max() + 1e-30) * max(1, K // 64) + 2e-11 * np.abs(y).max() * K
    if not np.allclose(a, b, rtol=0, atol=tol):
        bad += 1; print("MATMUL MISMATCH", B, M, K, N, wx, wy, np.abs(a - b).max(), tol)
print("matmul fuzz done, mismatches:", bad)
bad2 = 0
for it in range(40):
    M, N = int(r.integers(1, 700)), 16 * int(r.integers(1, 30))
    K = 128 * int(r.integers(1, 17))
    wx, ww = (6, 6) if it % 2 else (4, 6)
    x = np.maximum(r.normal(size=(M, K)), 0).astype(np.float32) * np.exp(r.normal(size=(M, 1))).astype(np.float32)
    x[:, :: int(r.integers(3, 40))] *= np.float32(10.0 ** r.integers(-4, 4))
    w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
    if it % 3 == 0: w[:: int(r.integers(2, 9)), 16 * int(r.integers(0, K // 16)):][:, :16] *= 200.0
    c = cfg(wx, ww, True)
    xt = torch.from_numpy(x).to(dev)
    _, wm, we = ops.block_fp_quantize(torch.from_numpy(w).to(dev), ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True,
                                      fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, ww - 1, 127)
    xa = ops.block_fp_quantize_aligned_rows(xt, wx, 8, 127, bucket_cap=int(r.choice([1016, 504, 200, 40])))
    y1 = ops.bfp_gemm_aligned(xa, wa, None).cpu().numpy()
    over, full = ops.row_list_fill(xa.sparse, M, xa.list_cap)
    ref = O.bfp_linear_int(x, w, None, c)
    tol = 3e-6 * np.abs(ref).max() * max(1, K // 256)
    if not np.allclose(y1, ref, rtol=0, atol=tol):
        bad2 += 1; print("POST MISMATCH", M, N, K, wx, ww, xa.list_cap, over, full, np.abs(y1 - ref).max(), tol)
print("post-pass fuzz done, mismatches:", bad2)
