"""Decode an MX operand (csrc/mi355q_quant.hip, MxOut) on the host and compare it with the oracle's fake-quantised tensor;
then the product against the oracle for scale patterns that tell fragments apart.  python tools/dbg/mx_decode.py"""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from mi355q import ops
from oracle import np_oracle as O
dev = torch.device("cuda:0")


def decode(op):
    rows, K = op.rows, op.K
    kp = K // 128
    c16 = op.c16.cpu().numpy(); c8 = op.c8.cpu().numpy(); sc = op.sc.cpu().numpy()
    out = np.zeros((rows, K), np.float64)
    lut = np.zeros(64)
    for v in range(64):
        s, e, m = (v >> 5) & 1, (v >> 3) & 3, v & 7
        mag = m / 8.0 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)
        lut[v] = -mag if s else mag
    for r in range(rows):
        for g in range(K // 32):
            ks, g4 = g // 4, g % 4
            o16 = ((r >> 4) * kp + ks) * 1024 + g4 * 256 + (r & 15) * 16
            o8 = ((r >> 5) * kp + ks) * 1024 + ((r >> 4) & 1) * 512 + g4 * 128 + (r & 15) * 8
            by = np.concatenate([c16[o16:o16 + 16], c8[o8:o8 + 8]]).astype(np.uint64)
            bits = 0
            for i, b in enumerate(by):
                bits |= int(b) << (8 * i)
            S = int(sc[((r >> 6) * kp + ks) * 256 + g4 * 64 + (r & 15) * 4 + ((r >> 4) & 3)])
            for t in range(32):
                out[r, g * 32 + t] = lut[(bits >> (6 * t)) & 63] * 2.0 ** (S - 127)
    return out


r = np.random.default_rng(0)
for (rows, K, width, style) in ((64, 256, 4, "randn"), (48, 128, 5, "scaled"), (80, 384, 4, "scaled")):
    x = r.normal(size=(rows, K)).astype(np.float32)
    if style == "scaled":
        x *= np.repeat(2.0 ** r.integers(-3, 3, size=(rows, K // 32)), 32, axis=1).astype(np.float32)
    op = ops.block_fp_quantize_mx(torch.from_numpy(x).to(dev), width, 8, 127, reuse=False)
    torch.cuda.synchronize()
    got = decode(op)
    ref = O.block_fp_quantize(x, width, 8, 127, [1, 16], True).astype(np.float64)
    d = np.abs(got - ref)
    print(f"quantiser rows {rows} K {K} W{width} {style}: bad {int(op.bad[0])}, max |decode - oracle| {d.max():.3g} (at {np.unravel_index(d.argmax(), d.shape)}), mismatches {int((d > 0).sum())} of {d.size}")
    if d.max() > 0:
        rr, cc = np.unravel_index(d.argmax(), d.shape)
        b0 = cc // 16 * 16
        print("  got", got[rr, b0:b0 + 16]); print("  ref", ref[rr, b0:b0 + 16])
