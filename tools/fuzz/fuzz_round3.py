"""Randomised cross-checks of round 3's new paths against the oracle:
  (1) the streaming quantisers (block_fp / block_minifloat / block_log) on random shapes with random all-zero block patterns,
      called in random order on one stream -- the zero-block fill speculation sees hits, misses and foreign fills;
  (2) the aligned-rows quantiser reading P row segments == the plain call (products compared);
  (3) block_minifloat products on bf16 MFMAs == the fp32 route.
    python tools/fuzz/fuzz_round3.py [seed]"""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import mi355q.quantize as Q
from mi355q import ops
from oracle import np_oracle as O
dev = torch.device("cuda:0")
r = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
KW = {"block_fp": lambda: dict(width=int(r.integers(2, 9)), exponent_width=8, exponent_bias=None, block_size=[1, 16]),
      "block_minifloat": lambda: dict(width=8, exponent_width=int(r.integers(2, 6)), exponent_bias_width=int(r.integers(2, 9)), block_size=[1, 16]),
      "block_log": lambda: dict(width=int(r.integers(3, 9)), exponent_bias_width=int(r.integers(2, 9)), block_size=[1, 16])}
bad = 0


def same_up_to_zero_sign(a, b):
    aw, bw = a.view(np.uint32), b.view(np.uint32)
    return np.array_equal(aw | np.where(a == 0, np.uint32(0x80000000), np.uint32(0)), bw | np.where(b == 0, np.uint32(0x80000000), np.uint32(0)))


for it in range(120):
    name = ("block_fp", "block_minifloat", "block_log")[int(r.integers(0, 3))]
    kw = KW[name]()
    lead, rows, cols = int(r.integers(1, 4)), int(r.integers(1, 90)), 16 * int(r.integers(1, 24))
    x = (r.normal(size=(lead, rows, cols)) * np.exp(r.normal(size=(lead, rows, 1)) * 3) * 10.0 ** r.integers(-12, 4)).astype(np.float32)
    mode = it % 5
    if mode == 0:
        x[:, :, : 16 * int(r.integers(0, cols // 16 + 1))] = 0
    elif mode == 1:
        x[r.random(size=x.shape) < 0.7] = 0
        x.reshape(-1, 16)[r.random(size=x.size // 16) < 0.5] = 0
    elif mode == 2:
        x[:] = 0 if it % 10 == 2 else x
    elif mode == 3:
        x[:, ::2, 16:32] = -0.0
    got = Q.get_quantizer("", dict(name=name))(torch.from_numpy(x).to(dev), **kw, skip_first_dim=True).cpu().numpy()
    want = np.asarray(getattr(O, name + "_quantize")(x, **kw, skip_first_dim=True), dtype=np.float32)
    ok = same_up_to_zero_sign(got, want) if not np.isnan(want).any() else np.array_equal(got, want, equal_nan=True)
    if not ok:
        bad += 1
        d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        print("QUANT MISMATCH", it, name, kw, x.shape, mode, len(d), d[:3].tolist(), [(float(x[tuple(i)]), float(got[tuple(i)]), float(want[tuple(i)])) for i in d[:3]])
print("quantisers done, bad =", bad)

ops.REUSE_QUANTISED_INPUT = False
for it in range(25):
    P = int(r.choice([1, 2, 4, 8]))
    seg = 64 * int(r.integers(1, 9))
    K, rows = P * seg, int(r.integers(1, 400))
    if K > 16384:
        continue
    width = int(r.integers(4, 7))
    x = torch.from_numpy((r.normal(size=(rows, K)) * np.exp(r.normal(size=(rows, 1)))).astype(np.float32)).to(dev)
    if it % 3 == 0:
        x[::3, 32:48] *= 2.0 ** -11
    w = torch.from_numpy((r.normal(size=(128, K)) * 0.1).astype(np.float32)).to(dev)
    _, wm, we = ops.block_fp_quantize(w, width, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, width - 1, 127)
    ya = ops.bfp_gemm_aligned(ops.block_fp_quantize_aligned_rows(x, width, 8, 127), wa).clone()
    s = x.view(rows, P, seg).permute(1, 0, 2).contiguous()
    yb = ops.bfp_gemm_aligned(ops.block_fp_quantize_aligned_rows(s, width, 8, 127, segments=True), wa)
    ref = O.bfp_linear_int(x.cpu().numpy(), w.cpu().numpy(), None, dict(data_in_width=width, data_in_exponent_width=8, data_in_exponent_bias=127,
                           data_in_block_size=[1, 16], weight_width=width, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16]))
    e1 = float((ya - yb).abs().max() / (ya.abs().max() + 1e-30))
    e2 = float(np.abs(yb.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30))
    if e1 > 2e-6 or e2 > 4e-6:
        bad += 1
        print("SEGMENT MISMATCH", it, rows, K, P, width, e1, e2)
print("segments done, bad =", bad)

cfg = dict(name="block_minifloat", bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias_width=8, data_in_block_size=[1, 16],
           weight_width=8, weight_exponent_width=4, weight_exponent_bias_width=8, weight_block_size=[1, 16])
for it in range(30):
    B, M, K, N = int(r.integers(1, 5)), int(r.integers(1, 150)), 16 * int(r.integers(1, 20)), 16 * int(r.integers(1, 12))
    x = (r.normal(size=(B, M, K)) * 10.0 ** r.integers(-3, 4)).astype(np.float32)
    y = (r.normal(size=(B, K, N)) * 10.0 ** r.integers(-2, 3)).astype(np.float32)
    if it % 4 == 0:
        x[:, :, :16] = 0
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    f = Q.get_quantized_func("bmm", cfg)
    a = f(xt, yt, dict(cfg)).cpu().numpy()
    b = f(xt, yt, dict(cfg, mi355q_values_matmul="fp32")).cpu().numpy()
    ref = O.matmul_quantized(x, y, cfg)
    # (elements |x| <= 1e-8, which the reference passes through unquantised, enter the bf16 product rounded to bf16: <= 4e-11 each,
    #  times |y|, times K -- all there is to see when everything else of x quantises to zero, quirk 5)
    sc = 3e-6 * (np.abs(ref).max() + 1e-30) + 4e-11 * (np.abs(y).max() + np.abs(x).max()) * K
    if np.abs(a - b).max() > sc or np.abs(a - ref).max() > sc:
        bad += 1
        print("MINIFLOAT MATMUL MISMATCH", it, (B, M, K, N), np.abs(a - b).max(), np.abs(a - ref).max(), sc)
print("all done, bad =", bad)
sys.exit(1 if bad else 0)
