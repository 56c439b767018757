#!/usr/bin/env python3
"""Randomised cross-check of round 4's product kernels: block_fp / block_minifloat / block_log products at random shapes (rows that
do not fill a workgroup, contractions of 1 .. 40 steps with partial last ones, column counts around the chunk / ring sizes), with
all-zero blocks, tiny blocks and rows of very different magnitude, against the oracle.  python tools/fuzz/fuzz_round4.py [seeds]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from mi355q import ops
from oracle import np_oracle as O

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
for seed in range(seeds):
    r = np.random.default_rng(seed)
    B = int(r.integers(1, 5)); M = int(r.integers(1, 300)); K = 16 * int(r.integers(1, 160)); N = 16 * int(r.integers(1, 40))
    if seed % 5 == 0: K = 16 * int(r.integers(1, 13))          # resident depths
    if seed % 7 == 0: N = 16 * int(r.integers(40, 140))        # many chunks
    x = (r.normal(size=(B, M, K)) * np.exp(r.normal(size=(B, M, 1)) * 2)).astype(np.float32)
    zb = r.random(size=(B, M, K // 16)) < 0.2
    x.reshape(B, M, K // 16, 16)[zb] = 0
    tb = r.random(size=(B, M, K // 16)) < 0.05
    x.reshape(B, M, K // 16, 16)[tb] *= 1e-7
    if seed % 3 == 0: x = np.abs(x)
    y = (r.normal(size=(B, K, N)) * np.exp(r.normal(size=(B, 1, N)))).astype(np.float32)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    wx, wy = int(r.integers(3, 9)), int(r.integers(3, 9))
    cases = [("block_fp", dict(name="block_fp", data_in_width=wx, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
                               weight_width=wy, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16]),
              lambda: ops.bfp_matmul(xt, yt, wx, 8, 127, wy, 8, 127))]
    ew = int(r.integers(2, 5)); ebw = int(r.integers(3, 9)); wm = ew + 1 + int(r.integers(0, 5))
    cases.append(("block_minifloat", dict(name="block_minifloat", data_in_width=wm, data_in_exponent_width=ew, data_in_exponent_bias_width=ebw,
                                          data_in_block_size=[1, 16], weight_width=wm, weight_exponent_width=ew, weight_exponent_bias_width=ebw,
                                          weight_block_size=[1, 16]),
                  lambda: ops.values_matmul(xt, yt, "block_minifloat", (wm, ew, ebw), (wm, ew, ebw))))
    wl, ebl = int(r.integers(3, 9)), int(r.integers(2, 9))
    cases.append(("block_log", dict(name="block_log", data_in_width=wl, data_in_exponent_bias_width=ebl, data_in_block_size=[1, 16],
                                    weight_width=wl, weight_exponent_bias_width=ebl, weight_block_size=[1, 16]),
                  lambda: ops.values_matmul(xt, yt, "block_log", (wl, ebl))))
    for name, cfg, f in cases:
        got = f().cpu().numpy()
        ref = O.matmul_quantized(x, y, dict(cfg, bypass=False))
        scale = max(float(np.abs(ref).max()), 1e-30)
        err = float(np.abs(got - ref).max()) / scale
        worst = max(worst, err)
        tol = 2e-6 * max(1, K // 64)
        flag = "" if err <= tol else "   <-- FAIL"
        if flag or seed < 3:
            print(f"seed {seed:3d} {name:16s} B{B} M{M} K{K} N{N}: rel err {err:.2e}{flag}", flush=True)
        assert err <= tol, (seed, name, B, M, K, N, err)
print(f"{seeds} seeds x 3 formats clean, worst relative error {worst:.2e}")
