#!/usr/bin/env python3
"""Randomised cross-check of round 5's kernels against the oracle: the row-scale int8 GEMM at random shapes under the launcher's own
choice (small tiles for small grids), with exception-rich operands; the MX W4A4 product; the bf16 product with the residual add in its
stores against the two steps; the bf16 quantiser with RMSNorm in front against the row quantiser's.   python tools/fuzz/fuzz_round5.py [seeds]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from mi355q import ops
from oracle import np_oracle as O

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
ops.REUSE_QUANTISED_INPUT = False
worst = {"int8": 0.0, "mx": 0.0}
for seed in range(seeds):
    r = np.random.default_rng(5000 + seed)
    M = int(r.integers(1, 1400)); N = int(r.integers(1, 900)); K = 64 * int(r.integers(1, 40))
    wx, ww = int(r.integers(3, 9)), int(r.integers(3, 9))
    x = (r.normal(size=(M, K)) * np.exp(r.normal(size=(M, 1)))).astype(np.float32)
    w = (r.normal(size=(N, K)) * 0.02 * np.exp(0.5 * r.normal(size=(N, 1)))).astype(np.float32)
    b = (r.normal(size=(N,)) * 0.02).astype(np.float32)
    # exception blocks of both operands, some at the same K position, some rows with several
    for _ in range(int(r.integers(0, 6))):
        kb = int(r.integers(0, K // 16)); f = float(2.0 ** r.integers(-9, 9))
        x[:: int(r.integers(3, 40)), kb * 16:(kb + 1) * 16] *= f
        if r.random() < 0.5:
            w[:: int(r.integers(3, 40)), kb * 16:(kb + 1) * 16] *= f
    x[r.random((M, K)) < 0.05] = 0
    cfg = dict(name="block_fp", data_in_width=wx, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
               weight_width=ww, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=ww,
               bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
    xt, wt, bt = (torch.from_numpy(a).to(dev) for a in (x, w, b))
    _, wm, we = ops.block_fp_quantize(wt, ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    bq = ops.block_fp_quantize(bt, ww, 8, 127, [16], False)
    wa = ops.bfp_align_rows(wm, we, ww - 1, 127)
    xa = ops.block_fp_quantize_aligned_rows(xt, wx, 8, 127)
    y = ops.bfp_gemm_aligned(xa, wa, bq).cpu().numpy()
    ref = O.bfp_linear_int(x, w, b, cfg)
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(y - ref).max()) / scale
    worst["int8"] = max(worst["int8"], err)
    tol = 6e-6 * max(1, K // 256)
    assert err <= tol, ("int8", seed, M, N, K, wx, ww, err)
    # ---- the bf16 product with the residual in its stores == the two steps, bit for bit
    if K % 32 == 0:
        xb = ops.block_fp_quantize_bf16_tiled(xt, min(wx, 8), 8, 127, reuse=False)
        wb = ops.block_fp_quantize_bf16_tiled(wt.clone(), min(ww, 8), 8, 127, reuse=False)
        Np = (N + 3) // 4 * 4
        res = torch.randn(M, Np, device=dev)[:, :N] if N % 4 else torch.randn(M, N, device=dev)
        two = res + ops.bf16_gemm_tiled(xb, wb, M, N, K, bq)
        if N % 4 == 0:
            one = ops.bf16_gemm_tiled(xb, wb, M, N, K, bq, residual=res)
            assert torch.equal(one, two), ("residual", seed, M, N, K)
    # ---- RMSNorm in front: the bf16 quantiser sees the row quantiser's normalised values (same mean, same roundings)
    if K <= 16384:
        wn = (1 + 0.1 * torch.randn(K, device=dev)).contiguous()
        h = wn * (xt * torch.rsqrt(xt.pow(2).mean(-1, keepdim=True) + 1e-6))
        fq = torch.empty_like(xt)
        ops.block_fp_quantize_bf16_tiled(xt, 6, 8, 127, out_fake=fq, reuse=False, pre=("rmsnorm", wn, 1e-6))
        want = ops.block_fp_quantize(h, 6, 8, 127, [1, 16], True)
        bad = (fq != want).float().mean().item()
        assert bad < 2e-3, ("rmsnorm", seed, M, K, bad)         # (the last bit of the mean moves a few elements across a rounding boundary)
    # ---- MX W4A4
    if K % 128 == 0:
        cfg4 = dict(cfg, data_in_width=4, weight_width=4, bias_width=4)
        wq4 = ops.block_fp_quantize(wt, 4, 8, 127, [1, 16], False)
        bq4 = ops.block_fp_quantize(bt, 4, 8, 127, [16], False)
        wop = ops.block_fp_quantize_mx(wt, 4, 8, 127, reuse=False)
        xop = ops.block_fp_quantize_mx(xt, 4, 8, 127, reuse=False)
        y4 = ops.mx_gemm(xop, wop, wq4, bq4).cpu().numpy()
        ref4 = O.bfp_linear_int(x, w, b, cfg4)
        e4 = float(np.abs(y4 - ref4).max()) / max(float(np.abs(ref4).max()), 1e-30)
        worst["mx"] = max(worst["mx"], e4)
        assert e4 <= 6e-6 * max(1, K // 256), ("mx", seed, M, N, K, e4, int(xop.bad[0]), int(wop.bad[0]))
    if seed < 3 or seed % 10 == 0:
        print(f"seed {seed:3d} M{M} N{N} K{K} W{wx}A{ww}: int8 {err:.2e}", flush=True)
print(f"{seeds} seeds clean; worst relative error int8 {worst['int8']:.2e}, MX {worst['mx']:.2e}")
