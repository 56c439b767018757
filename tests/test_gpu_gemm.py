"""GPU parity of the block-fp int8-MFMA GEMM (through the C ABI) against the oracle's exact
integer contraction and against an fp32 GEMM on the fake-quantised operands (the reference's
F.linear semantics).  Tolerance: BASELINE.json north_star, <= 1e-3 on the dequantised result
(relative to the output scale); the integer block dots themselves are exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG6 = dict(name="block_fp", data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127,
            bias_block_size=[16])


def _cfg(wx, ww):
    c = dict(CFG6)
    c["data_in_width"], c["weight_width"] = wx, ww
    return c


def _inputs(M, N, K, seed, style):
    r = np.random.default_rng(seed)
    x = r.normal(size=(M, K)).astype(np.float32)
    w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
    b = (r.normal(size=(N,)) * 0.02).astype(np.float32)
    if style == "rowscale":
        x *= np.exp(r.normal(size=(M, 1))).astype(np.float32)
    elif style == "outlier":                 # neighbouring K-blocks with far-apart exponents
        x[:, ::53] *= 200.0
        w[::7, 5::41] *= 64.0
    elif style == "sparse":
        x[r.random((M, K)) < 0.6] = 0
        x[:, 32:80] = 0
    return x, w, b


def _run(x, w, b, cfg, aligned=False, x_cap=0):
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    xt, wt = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    _, xm, xe = ops.block_fp_quantize(xt, cfg["data_in_width"], 8, 127, [1, 16], True,
                                      want_fake=False, want_packed=True, fast_zero_blocks=True)
    _, wm, we = ops.block_fp_quantize(wt, cfg["weight_width"], 8, 127, [1, 16], False,
                                      want_fake=False, want_packed=True, fast_zero_blocks=True)
    bq = None
    if b is not None:
        bq = ops.block_fp_quantize(torch.from_numpy(b).to(dev), cfg["bias_width"], 8, 127, [16], False)
    if aligned:
        assert aligned == "rows"
        align = ops.bfp_align_rows
        xa = (align(xm, xe, cfg["data_in_width"] - 1, 127, bucket_cap=x_cap) if x_cap
              else align(xm, xe, cfg["data_in_width"] - 1, 127))
        wa = align(wm, we, cfg["weight_width"] - 1, 127)
        y = ops.bfp_gemm_aligned(xa, wa, bq)
        _run.last_flags = (float(xa.rowflag.float().mean()), float(wa.rowflag.float().mean()))
        _run.last_counts = (int(xa.sparse[0]), int(wa.sparse[0]))
    else:
        y = ops.bfp_gemm(xm, xe, wm, we, bq, cfg["data_in_width"] - 1, 127, cfg["weight_width"] - 1, 127)
    torch.cuda.synchronize()
    return y.cpu().numpy()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (100, 72, 48), (1, 16, 16), (300, 130, 1024)])
@pytest.mark.parametrize("style", ["randn", "rowscale", "outlier", "sparse"])
@pytest.mark.parametrize("wx,ww", [(6, 6), (4, 4), (8, 8), (6, 4)])
def test_gemm_vs_oracle(M, N, K, style, wx, ww):
    from oracle import np_oracle as O
    x, w, b = _inputs(M, N, K, 1000 + M + N + K, style)
    cfg = _cfg(wx, ww)
    y = _run(x, w, b, cfg)
    ref = O.bfp_linear_int(x, w, b, cfg)
    scale = np.abs(ref).max() + 1e-30
    # fp32 accumulation over K/16 block products: a few ulp of the output scale
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * scale * max(1, K // 256))


def test_gemm_no_bias_and_column_slice():
    """ldy > N: a rank writes its column slice of a wider output (row-sharded linear)"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    x, w, _ = _inputs(64, 96, 128, 77, "rowscale")
    cfg = _cfg(6, 6)
    dev = torch.device("cuda:0")
    _, xm, xe = ops.block_fp_quantize(torch.from_numpy(x).to(dev), 6, 8, 127, [1, 16], True, want_fake=False, want_packed=True)
    full = torch.full((64, 96), float("nan"), device=dev)
    for lo, hi in ((0, 48), (48, 96)):
        _, wm, we = ops.block_fp_quantize(torch.from_numpy(w[lo:hi].copy()).to(dev), 6, 8, 127, [1, 16], False,
                                          want_fake=False, want_packed=True)
        ops.bfp_gemm(xm, xe, wm, we, None, 5, 127, 5, 127, out=full[:, lo:hi])
    ref = O.bfp_linear_int(x, w, None, cfg)
    np.testing.assert_allclose(full.cpu().numpy(), ref, rtol=0, atol=2e-6 * np.abs(ref).max())


def test_gemm_matches_reference_semantics_fp32_linear():
    """F.linear on the fake-quantised operands (what the reference computes), tolerance 1e-3"""
    import torch
    from mi355q import ops
    x, w, b = _inputs(512, 512, 2048, 9, "rowscale")
    cfg = _cfg(6, 6)
    y = _run(x, w, b, cfg)
    dev = torch.device("cuda:0")
    xq = ops.block_fp_quantize(torch.from_numpy(x).to(dev), 6, 8, 127, [1, 16], True)
    wq = ops.block_fp_quantize(torch.from_numpy(w).to(dev), 6, 8, 127, [1, 16], False)
    bq = ops.block_fp_quantize(torch.from_numpy(b).to(dev), 6, 8, 127, [16], False)
    ref = (xq.double() @ wq.double().T + bq.double()).float().cpu().numpy()
    np.testing.assert_allclose(y, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


# ---------------------------------------------------------------------------------------------------
# ROW-aligned operands (one exponent per row; int8 GEMM with row / column scales; bucketed exception lists)
# ---------------------------------------------------------------------------------------------------
def _row_exceptions(al, rows, nkb):
    from mi355q import ops
    over, ent = ops.row_list_entries(al.sparse, rows, al.list_cap)
    mant = np.zeros((rows, nkb, 16), np.int64)
    exp = np.zeros((rows, nkb), np.int64)
    mask = np.zeros((rows, nkb), bool)
    for e in ent:
        assert not mask[e[0], e[1]], "a block is listed once"
        assert e[0] // 256 == e[0] // 256
        mask[e[0], e[1]] = True
        exp[e[0], e[1]] = e[2]
        mant[e[0], e[1]] = e[4:8].astype(np.int32).view(np.int8)
    return over, mant, exp, mask


def _untile(tiled, rows, K):
    """tiled aligned mantissas (1-KiB pieces of 16 rows x 64 B laid out [block 0..3][row 0..15][16 B]) -> [rows, K] int8"""
    t = tiled.cpu().numpy().view(np.int8)
    kp = K // 64
    rp = (t.size // (kp * 1024)) * 16
    return np.ascontiguousarray(t[: (rp // 16) * kp * 1024].reshape(rp // 16, kp, 4, 16, 16).transpose(0, 3, 1, 2, 4)
                                ).reshape(rp, K)[:rows]


@pytest.mark.parametrize("width", [6, 4, 8])
def test_row_align_is_value_preserving(width):
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    rows, K = 300, 1024
    x, _, _ = _inputs(rows, 8, K, 5 + width, "rowscale")
    x[::4, 256:272] *= 300.0          # one block far above its neighbours in every 4th row (64 per bucket of 256 rows)
    x[7, :] = 0.0                      # an all-zero row
    x[9, 16:] = 0.0                    # a row with one non-zero block
    _, xm, xe = ops.block_fp_quantize(torch.from_numpy(x).to(dev), width, 8, 127, [1, 16], True, want_fake=False,
                                      want_packed=True)
    al = ops.bfp_align_rows(xm, xe, width - 1, 127)
    torch.cuda.synchronize()
    over, em, ee, emask = _row_exceptions(al, rows, K // 16)
    f = al.rowflag.cpu().numpy()
    m2 = _untile(al.tiled, rows, K).reshape(rows, K // 16, 16)
    e2 = al.exp.cpu().numpy().reshape(rows, K // 16).astype(np.float64)
    v1 = xm.cpu().numpy().reshape(rows, K // 16, 16).astype(np.float64) * np.exp2(xe.cpu().numpy().reshape(rows, K // 16, 1).astype(np.float64))
    v2 = m2.astype(np.float64) * np.exp2(e2[..., None]) + em.astype(np.float64) * np.exp2(ee[..., None].astype(np.float64))
    assert np.array_equal(v1, v2), "row-aligned operand + its exception blocks denote the input exactly"
    assert np.all(m2[emask] == 0)
    assert over == (f == 0).sum()
    fl = f == 1
    assert np.all(e2[fl] == e2[fl][:, :1]), "flagged rows carry one exponent"
    rs = al.gscale.cpu().numpy()[:rows]
    assert np.array_equal(rs[fl], np.exp2(e2[fl][:, 0] - (127 + width - 1))) and np.all(rs[~fl] == 0)
    if width < 8:
        assert over == 0 and fl.all() and emask[::4, 16].all() and emask.sum() < 75 + 30
    else:                               # no head-room at W8: far too many exceptions, most rows stay unaligned
        assert over > 0


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 130, 1024), (100, 72, 384), (33, 16, 128), (520, 260, 512),
                                   (64, 64, 192)])
@pytest.mark.parametrize("style", ["randn", "rowscale", "outlier", "sparse"])
@pytest.mark.parametrize("wx,ww", [(6, 6), (4, 4), (8, 8), (6, 4)])
@pytest.mark.parametrize("tile_rows,x_cap", [(256, 0), (128, 0), (256, 1016), (128, 40)])
def test_row_aligned_gemm_vs_oracle(M, N, K, style, wx, ww, tile_rows, x_cap, monkeypatch):
    """row-scale int8 GEMM + exception add-back in its epilogue; overflowing buckets (outlier data, W8) and
    K % 128 != 0 take the blockwise kernel"""
    from oracle import np_oracle as O
    monkeypatch.setenv("MI355Q_V8_TILE_ROWS", str(tile_rows))      # both workgroup-tile flavours of the row-scale kernel
    x, w, b = _inputs(M, N, K, 3000 + M + N + K, style)
    cfg = _cfg(wx, ww)
    y = _run(x, w, b, cfg, aligned="rows", x_cap=x_cap)       # x_cap != 0: x's exceptions through the row post-pass
    ref = O.bfp_linear_int(x, w, b, cfg)
    scale = np.abs(ref).max() + 1e-30
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * scale * max(1, K // 256))


@pytest.mark.parametrize("variant,tile_rows,x_cap", [(0, 256, 0), (0, 128, 0), (2, 0, 0), (0, 256, 1016), (0, 128, 504),
                                                     (2, 0, 1016)])
def test_row_aligned_gemm_exceptions_of_both_operands_share_blocks(variant, tile_rows, x_cap, monkeypatch):
    from mi355q import ops
    monkeypatch.setenv("MI355Q_V8_TILE_ROWS", str(tile_rows))
    from oracle import np_oracle as O
    x, w, b = _inputs(520, 300, 768, 17, "rowscale")
    x[::5, 256:272] *= 300.0
    w[::4, 256:272] *= 300.0
    x[::7, 512:528] *= 1e-3
    w[1::6, 512:528] *= 1e-3
    cfg = _cfg(6, 6)
    prev = ops.set_gemm_variant(variant)
    try:
        y = _run(x, w, b, cfg, aligned="rows", x_cap=x_cap)
    finally:
        ops.set_gemm_variant(prev)
    assert _run.last_counts == (0, 0) and _run.last_flags == (1.0, 1.0)        # no bucket overflowed
    ref = O.bfp_linear_int(x, w, b, cfg)
    np.testing.assert_allclose(y, ref, rtol=0, atol=6e-6 * np.abs(ref).max())


def test_row_post_pass_takes_what_the_tile_cannot():
    """post-activation input (half zeros, block maxima spread over many exponents): ~2 exception blocks per row, far
    beyond what a GEMM tile adds from LDS; with 1016-entry buckets nothing overflows and the row post-pass adds
    them -- exact against the oracle, bit-identical between runs and re-quantisations"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    r = np.random.default_rng(5)
    M, N, K = 600, 200, 3072
    x = np.maximum(r.normal(size=(M, K)), 0).astype(np.float32) * np.exp(r.normal(size=(M, 1))).astype(np.float32)
    x[:, ::7] *= np.float32(0.01)
    w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
    w[::9, 512:528] *= 300.0                    # weight exceptions, some in blocks where x has one too
    b = r.normal(size=N).astype(np.float32)
    cfg = _cfg(6, 6)
    y = _run(x, w, b, cfg, aligned="rows", x_cap=1016)
    assert _run.last_counts == (0, 0) and _run.last_flags == (1.0, 1.0)
    ref = O.bfp_linear_int(x, w, b, cfg)
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * np.abs(ref).max() * (K // 256))
    dev = torch.device("cuda:0")
    xt = torch.from_numpy(x).to(dev)
    _, wm, we = ops.block_fp_quantize(torch.from_numpy(w).to(dev), 6, 8, 127, [1, 16], False, want_fake=False,
                                      want_packed=True, fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    bq = ops.block_fp_quantize(torch.from_numpy(b).to(dev), 6, 8, 127, [16], False)
    outs = []
    for _ in range(3):
        xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, bucket_cap=1016)
        outs.append(ops.bfp_gemm_aligned(xa, wa, bq).clone())
    torch.cuda.synchronize()
    over, fullest = ops.row_list_fill(xa.sparse, M, 1016)
    assert over == 0 and fullest > 200, fullest
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref, rtol=0, atol=2e-6 * np.abs(ref).max() * (K // 256))
    # the same input with the default buckets overflows them: blockwise fallback, same values within rounding
    y2 = _run(x, w, b, cfg, aligned="rows")
    assert _run.last_counts[0] > 0
    np.testing.assert_allclose(y2, ref, rtol=0, atol=2e-6 * np.abs(ref).max() * (K // 256))


@pytest.mark.parametrize("M,N,K", [(300, 130, 1024), (64, 48, 128), (520, 260, 512)])
@pytest.mark.parametrize("wx", [6, 4])
def test_unaligned_activations_take_the_blockwise_kernel(M, N, K, wx):
    """inputs no row window fits (SiLU-gated products: block exponents spread over ~20 values in every row): the fused
    quantiser can leave the row format unaligned (bucket_cap = ROW_NO_ALIGN) and the GEMM runs blockwise-exact against
    row-aligned weights, whose own exception blocks are added per tile"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    r = np.random.default_rng(M + K)
    g, u = r.normal(size=(M, K)), r.normal(size=(M, K))
    x = ((g / (1 + np.exp(-g))) * u * np.exp(3 * r.normal(size=(M, K // 16, 1)).repeat(16, 2).reshape(M, K))).astype(np.float32)
    w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
    w[::7, 256 % K: 256 % K + 16] *= 300.0
    b = r.normal(size=N).astype(np.float32)
    cfg = _cfg(wx, 6)
    dev = torch.device("cuda:0")
    _, wm, we = ops.block_fp_quantize(torch.from_numpy(w).to(dev), 6, 8, 127, [1, 16], False, want_fake=False,
                                      want_packed=True, fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    bq = ops.block_fp_quantize(torch.from_numpy(b).to(dev), 6, 8, 127, [16], False)
    xa = ops.block_fp_quantize_aligned_rows(torch.from_numpy(x).to(dev), wx, 8, 127, bucket_cap=ops.ROW_NO_ALIGN)
    assert xa.unaligned and not bool(xa.rowflag.any()) and xa.sparse is None
    y = ops.bfp_gemm_aligned(xa, wa, bq).cpu().numpy()
    ref = O.bfp_linear_int(x, w, b, cfg)
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * np.abs(ref).max() * max(1, K // 256))
    # the same input through row alignment overflows even the large buckets (and still gives the same values)
    xr = ops.block_fp_quantize_aligned_rows(torch.from_numpy(x).to(dev), wx, 8, 127, bucket_cap=1016)
    if M >= 256 and K >= 512:
        assert ops.row_list_fill(xr.sparse, M, 1016)[0] > 0
    y2 = ops.bfp_gemm_aligned(xr, wa, bq).cpu().numpy()
    np.testing.assert_allclose(y2, ref, rtol=0, atol=2e-6 * np.abs(ref).max() * max(1, K // 256))


def test_row_aligned_gemm_bucket_overflow_takes_the_fallback():
    from oracle import np_oracle as O
    r = np.random.default_rng(12)
    M, N, K = 700, 128, 512
    x = r.normal(size=(M, K)).astype(np.float32)
    x[256:512, 7::64] *= 500.0                  # 8 far-off blocks in each row of the second bucket: 2048 > 120 entries
    w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
    cfg = _cfg(6, 6)
    y = _run(x, w, None, cfg, aligned="rows")
    assert _run.last_counts[0] > 0 and _run.last_flags[0] < 1.0
    ref = O.bfp_linear_int(x, w, None, cfg)
    np.testing.assert_allclose(y, ref, rtol=0, atol=4e-6 * np.abs(ref).max())


@pytest.mark.parametrize("style,rows", [("rowscale", 300), ("outlier", 40), ("sparse", 24)])
@pytest.mark.parametrize("width", [6, 4])
@pytest.mark.parametrize("K", [1024, 4096, 320, 5120])
def test_fused_quantize_align_rows_equals_two_step(style, rows, width, K, monkeypatch):
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    if K > 1024:
        rows = min(rows, 64 if style == "rowscale" else 6)
    if style == "outlier":              # ~K/53 exception blocks per row: stay inside one bucket (120 entries)
        rows = max(1, 100 // (K // 53 + 1))
    x, _, _ = _inputs(rows, 8, K, 41 + width, style)
    xt = torch.from_numpy(x).to(dev)
    _, xm, xe = ops.block_fp_quantize(xt, width, 8, 127, [1, 16], True, want_fake=False, want_packed=True,
                                      fast_zero_blocks=True)
    ref = ops.bfp_align_rows(xm, xe, width - 1, 127)
    got = ops.block_fp_quantize_aligned_rows(xt, width, 8, 127)
    torch.cuda.synchronize()
    o1, e1 = ops.row_list_entries(ref.sparse, rows)
    o2, e2 = ops.row_list_entries(got.sparse, rows)
    assert o1 == o2 == 0
    assert set(map(tuple, e1)) == set(map(tuple, e2))
    if style == "outlier":
        assert len(e1) > 0
    full = (rows // 16) * 16 * K
    assert torch.equal(got.tiled[:full], ref.tiled[:full])
    assert torch.equal(got.exp, ref.exp.reshape(-1)) and torch.equal(got.rowflag, ref.rowflag)
    assert torch.equal(got.gscale[:rows], ref.gscale[:rows])
    # alternating lists: the second call fills the other list, the third one the first again (emptied in between)
    # (with the shared-activation reuse off: the same tensor again would hand back the operand already in the buffers)
    monkeypatch.setattr(ops, "REUSE_QUANTISED_INPUT", False)
    again = ops.block_fp_quantize_aligned_rows(xt, width, 8, 127)
    third = ops.block_fp_quantize_aligned_rows(xt, width, 8, 127)
    torch.cuda.synchronize()
    assert again.sparse.data_ptr() != got.sparse.data_ptr() and third.sparse.data_ptr() == got.sparse.data_ptr()
    o3, e3 = ops.row_list_entries(third.sparse, rows)
    assert o3 == 0 and set(map(tuple, e3)) == set(map(tuple, e1))


@pytest.mark.parametrize("tile_rows", [256, 128])
def test_row_aligned_gemm_is_reproducible(tile_rows, monkeypatch):
    """no floating-point atomics on the row-aligned path (up to 96 exception blocks per 256 x 256 tile): the sum of a
    row's correction vectors follows the blocks' K order, not the order in which workgroups reserved their list
    slots, so repeated runs -- and repeated quantisations -- give bit-identical results"""
    import torch
    from mi355q import ops
    monkeypatch.setenv("MI355Q_V8_TILE_ROWS", str(tile_rows))
    dev = torch.device("cuda:0")
    M, N, K = 520, 300, 768
    x, w, b = _inputs(M, N, K, 4321, "rowscale")
    x[::12, 256:272] *= 300.0
    w[::16, 256:272] *= 300.0
    x[::28, 512:528] *= 1e-3
    x[::36, 64:80] *= 300.0                    # rows with two or three exception blocks
    xt = torch.from_numpy(x).to(dev)
    _, wm, we = ops.block_fp_quantize(torch.from_numpy(w).to(dev), 6, 8, 127, [1, 16], False, want_fake=False,
                                      want_packed=True, fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    outs = []
    for _ in range(4):
        xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127)      # slot order inside the buckets differs run to run
        outs.append(ops.bfp_gemm_aligned(xa, wa, None).clone())
        outs.append(ops.bfp_gemm_aligned(xa, wa, None).clone())
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_fused_quantize_align_rows_tiny_blocks_keep_the_zero_rule():
    """blocks below 2^-23 with elements equal to -1e-9 (sign(x + 1e-9) = 0 there: block_fp.py:69) and |x| <= 1e-8:
    the fused kernel's shortcut for normal magnitudes must not change them"""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    r = np.random.default_rng(3)
    rows, K = 16, 1024                      # (six tiny blocks per row: 96 exception entries fit one bucket)
    x = (r.normal(size=(rows, K)) * np.exp(r.normal(size=(rows, 1)))).astype(np.float32)
    eps = np.float32(1e-9)
    x[:, 0:64] = (r.integers(-40, 40, size=(rows, 64)) * eps).astype(np.float32)      # multiples of 1e-9, incl. -1e-9 and 0
    x[::2, 64:80] = -eps
    x[1::2, 80:96] = (r.normal(size=(rows // 2, 16)) * 3e-9).astype(np.float32)
    assert (x == -eps).any()
    xt = torch.from_numpy(x).to(dev)
    for width in (6, 4):                    # (W8 has no head-room: its rows overflow the bucket)
        _, xm, xe = ops.block_fp_quantize(xt, width, 8, 127, [1, 16], True, want_fake=False, want_packed=True,
                                          fast_zero_blocks=True)
        ref = ops.bfp_align_rows(xm, xe, width - 1, 127)
        got = ops.block_fp_quantize_aligned_rows(xt, width, 8, 127)
        torch.cuda.synchronize()
        o1, e1 = ops.row_list_entries(ref.sparse, rows)
        o2, e2 = ops.row_list_entries(got.sparse, rows)
        assert o1 == o2 == 0 and set(map(tuple, e1)) == set(map(tuple, e2))
        assert torch.equal(got.tiled[: rows * K], ref.tiled[: rows * K])
        assert torch.equal(got.exp, ref.exp.reshape(-1)) and torch.equal(got.gscale[:rows], ref.gscale[:rows])


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 130, 1024), (100, 72, 96), (33, 16, 32), (520, 260, 4096), (1024, 768, 3072)])
@pytest.mark.parametrize("style", ["rowscale", "outlier", "sparse"])
@pytest.mark.parametrize("wx,ww", [(6, 6), (4, 4), (8, 6)])
def test_bf16_tiled_gemm_vs_oracle(M, N, K, style, wx, ww):
    """operands whose blocks keep their own exponents: quantise straight into tiled bf16, bf16 flavour of the tile GEMM
    (fp32 accumulation of exact products, like the reference's F.linear on the fake-quantised values)"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    x, w, b = _inputs(M, N, K, 4000 + M + N + K, style)
    cfg = _cfg(wx, ww)
    dev = torch.device("cuda:0")
    xt = ops.block_fp_quantize_bf16_tiled(torch.from_numpy(x).to(dev), wx, 8, 127)
    wdev = torch.from_numpy(w).to(dev)
    wt = ops.block_fp_quantize_bf16_tiled(wdev, ww, 8, 127, out_fake=wdev, reuse=False)
    assert np.array_equal(wdev.cpu().numpy(), O.block_fp_quantize(w, ww, 8, 127, [1, 16], False))   # in-place fake-quant
    bq = ops.block_fp_quantize(torch.from_numpy(b).to(dev), ww, 8, 127, [16], False)
    y = ops.bf16_gemm_tiled(xt, wt, M, N, K, bq).cpu().numpy()
    ref = O.linear_ptq(x, w, b, dict(cfg, bias_width=ww))[0]
    scale = np.abs(ref).max() + 1e-30
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * scale * max(1, K // 256))


@pytest.mark.parametrize("M,K,N,P", [(300, 512, 272, 2), (300, 512, 272, 4), (1024, 2048, 512, 8), (130, 192, 64, 2), (4096, 1024, 256, 4),
                                     (77, 256, 48, 8), (4096, 4096, 512, 8)])
@pytest.mark.parametrize("geom", [0, 1, 2, 3, 4])
def test_bf16_tile_gemm_with_x_in_column_segments(M, K, N, P, geom, monkeypatch):
    """mi355q_bf16_gemm_tiled_seg (ABI 20): x handed over as P column segments, each its own tiled bf16 operand, rank-major --
    what an all-gather of per-rank quantised output slices leaves (sharded.py, gather = "quantised") -- gives the plain call's
    result on the re-assembled operand, and the oracle's.  geom: the launcher's choice (0) or a pinned small-tile geometry (the
    small tiles read segments since the end of round 5)"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    if geom:
        monkeypatch.setenv("MI355Q_V10", str(geom))
    r = np.random.default_rng(M + K + P)
    x = (r.normal(size=(M, K)) * np.exp(r.normal(size=(M, 1)))).astype(np.float32)
    w = (r.normal(size=(N, K)) * 0.05).astype(np.float32)
    b = r.normal(size=(N,)).astype(np.float32)
    xt_, wt_, bt = (torch.from_numpy(a).to("cuda:0") for a in (x, w, b))
    wt = ops.block_fp_quantize_bf16_tiled(wt_, 6, 8, 127, reuse=False)
    plain = ops.bf16_gemm_tiled(ops.block_fp_quantize_bf16_tiled(xt_, 6, 8, 127, reuse=False), wt, M, N, K, bt)
    segs = [ops.block_fp_quantize_bf16_tiled(xt_[:, s * K // P:(s + 1) * K // P].contiguous(), 6, 8, 127, reuse=False) for s in range(P)]
    stacked = torch.stack([s_.reshape(-1) for s_ in segs]).contiguous()
    got = ops.bf16_gemm_tiled(stacked, wt, M, N, K, bt, segments=P)
    xq = O.block_fp_quantize(x, 6, 8, 127, [1, 16], True).astype(np.float64)
    wq = O.block_fp_quantize(w, 6, 8, 127, [1, 16], False).astype(np.float64)
    ref = xq @ wq.T + b
    scale = np.abs(ref).max()
    assert np.abs(got.cpu().numpy() - ref).max() <= 2e-6 * scale
    assert (got - plain).abs().max().item() <= 1e-6 * scale
    if K % 128:                                  # (both calls on the same kernel: the same bits)
        assert torch.equal(got, plain)


# ---- round 5: the small tiles (csrc/mi355q_gemm_v10.hip) ----------------------------------------------------------------
# (geometry, ring stages; 0 = the launcher's depth rule); 5 / 6 (round 6): 128 x 128 / 128 x 256 with two K-groups in an 8-wave workgroup
V10_GEOMS = [(1, 0), (1, 6), (2, 0), (3, 0), (3, 4), (3, 8), (4, 0), (4, 4), (5, 0), (6, 0)]


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 130, 1024), (100, 72, 384), (33, 16, 128), (520, 260, 512), (64, 64, 192),
                                   (640, 384, 64)])
@pytest.mark.parametrize("style", ["randn", "rowscale", "outlier", "sparse"])
@pytest.mark.parametrize("wx,ww", [(6, 6), (4, 4), (8, 8), (6, 4)])
@pytest.mark.parametrize("geom,ns", V10_GEOMS)
@pytest.mark.parametrize("splits", [0, 2])
def test_small_tile_gemm_vs_oracle(M, N, K, style, wx, ww, geom, ns, splits, monkeypatch):
    """every geometry / ring depth of the small-tile kernel (128 x 256, 256 x 128, 128 x 128; two or three workgroups a compute
    unit or one with a deep ring; unsplit and split in two) against the oracle's exact integer contraction: exception
    add-back from buckets fetched in front of the K loop or behind it, odd numbers of K-steps, ragged edges, overflowing
    buckets (the blockwise fallback inside the launch)"""
    from oracle import np_oracle as O
    if splits and (K // 64) % splits:
        pytest.skip("K-steps do not split evenly")
    monkeypatch.setenv("MI355Q_V10", str(geom))
    if ns:
        monkeypatch.setenv("MI355Q_V10_NS", str(ns))
    if splits:
        monkeypatch.setenv("MI355Q_V8_SPLITS", str(splits))
    x, w, b = _inputs(M, N, K, 7000 + M + N + K, style)
    cfg = _cfg(wx, ww)
    y = _run(x, w, b, cfg, aligned="rows")
    ref = O.bfp_linear_int(x, w, b, cfg)
    scale = np.abs(ref).max() + 1e-30
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * scale * max(1, K // 256))


def _exception_rich_operands(M, N, K, seed=4321):
    x, w, b = _inputs(M, N, K, seed, "rowscale")
    x[::12, 256:272] *= 300.0
    w[::16, 256:272] *= 300.0                   # (the same K position in both lists: exception x exception terms)
    x[::28, 512:528] *= 1e-3
    x[::36, 64:80] *= 300.0                     # rows with two or three exception blocks
    w[5::24, 128:144] *= 200.0
    return x, w, b


def test_small_tiles_equal_the_256_tile_bit_for_bit(monkeypatch):
    """the small-tile kernel forms and adds a row's / column's corrections in the 256 x 256 kernel's order (one vector per
    entry, exception x exception terms behind it, rows with several entries folded in ascending block order, row vector then
    column vectors in the store): every geometry gives the 256 x 256 kernel's bits -- what lets the launcher pick a tile per
    shape, and a grouped launch differ from its separate launches in tile size, without changing a result"""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    M, N, K = 520, 300, 768
    x, w, b = _exception_rich_operands(M, N, K)
    xt = torch.from_numpy(x).to(dev)
    _, wm, we = ops.block_fp_quantize(torch.from_numpy(w).to(dev), 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True,
                                      fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    bq = ops.block_fp_quantize(torch.from_numpy(b).to(dev), 6, 8, 127, [16], False)
    xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127)
    assert int(xa.sparse[0]) == 0 and int(wa.sparse[0]) == 0
    monkeypatch.setenv("MI355Q_V8_TILE_ROWS", "256")            # (a pinned tile height keeps the launch on the 256 x 256 kernel)
    ref = ops.bfp_gemm_aligned(xa, wa, bq).clone()
    monkeypatch.delenv("MI355Q_V8_TILE_ROWS")
    for geom, ns in V10_GEOMS:
        monkeypatch.setenv("MI355Q_V10", str(geom))
        monkeypatch.setenv("MI355Q_V10_NS", str(ns)) if ns else monkeypatch.delenv("MI355Q_V10_NS", raising=False)
        for _ in range(2):
            xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127)      # (slot order inside the buckets differs run to run)
            y = ops.bfp_gemm_aligned(xa, wa, bq)
            torch.cuda.synchronize()
            assert torch.equal(y, ref), (geom, ns, float((y - ref).abs().max()))
    monkeypatch.delenv("MI355Q_V10")
    monkeypatch.delenv("MI355Q_V10_NS", raising=False)
    y = ops.bfp_gemm_aligned(xa, wa, bq)                         # the launcher's own choice for this shape (6 tiles of 256 x 256)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 130, 1024), (100, 72, 96), (520, 260, 4096), (1024, 768, 3072), (2048, 512, 704)])
@pytest.mark.parametrize("geom,ns", V10_GEOMS)
def test_small_tile_bf16_gemm_vs_oracle(M, N, K, geom, ns, monkeypatch):
    """the bf16 flavour of every small-tile geometry (operands whose blocks keep their own exponents) against the oracle"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    if (2 * K) % 64:
        pytest.skip("K bytes not a multiple of the K-step")
    monkeypatch.setenv("MI355Q_V10", str(geom))
    if ns:
        monkeypatch.setenv("MI355Q_V10_NS", str(ns))
    dev = torch.device("cuda:0")
    x, w, b = _inputs(M, N, K, 9000 + M + N + K, "outlier")
    cfg = _cfg(6, 6)
    xt = ops.block_fp_quantize_bf16_tiled(torch.from_numpy(x).to(dev), 6, 8, 127)
    wt = ops.block_fp_quantize_bf16_tiled(torch.from_numpy(w).to(dev), 6, 8, 127, reuse=False)
    bq = ops.block_fp_quantize(torch.from_numpy(b).to(dev), 6, 8, 127, [16], False)
    y = ops.bf16_gemm_tiled(xt, wt, M, N, K, bq).cpu().numpy()
    ref = O.linear_ptq(x, w, b, cfg)[0]
    scale = np.abs(ref).max() + 1e-30
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * scale * max(1, K // 256))


@pytest.mark.parametrize("geom,splits", [(3, 2), (3, 4), (1, 2), (3, 8)])
@pytest.mark.parametrize("bf16", [False, True])
def test_split_k_protocol_under_repetition(geom, splits, bf16, monkeypatch):
    """the split-K hand-over of the small-tile kernel (round 5: slabs stored / loaded at agent scope instead of behind release /
    acquire fences; order-free sums take the ticket first and only the non-final slices publish) relies on what the hardware does
    with sc1 accesses: 300 launches per setting over grids of 32-512 workgroups, every one bit-equal to the unsplit launch (int32
    sums, and two fp32 slices, are order-free; more fp32 slices are summed in slice order) -- a stale slab or a missed ticket
    would show"""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    M, N, K = 1024, 512, 4096
    x, w, b = _exception_rich_operands(M, N, K, seed=99)
    xt, wt = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    monkeypatch.setenv("MI355Q_V10", str(geom))
    if bf16:
        xa = ops.block_fp_quantize_bf16_tiled(xt, 6, 8, 127, reuse=False)
        wa = ops.block_fp_quantize_bf16_tiled(wt, 6, 8, 127, reuse=False)
        run = lambda: ops.bf16_gemm_tiled(xa, wa, M, N, K, None)
    else:
        _, wm, we = ops.block_fp_quantize(wt, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
        wa = ops.bfp_align_rows(wm, we, 5, 127)
        xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127)
        run = lambda: ops.bfp_gemm_aligned(xa, wa, None)
    unsplit = run().clone()
    monkeypatch.setenv("MI355Q_V8_SPLITS", str(splits))
    # (int32 slices: the unsplit launch's bits.  fp32 slices: their own -- another association of the same terms -- but the same every
    #  launch: two slices are order-free, more are summed in slice order)
    ref = unsplit if not bf16 else run().clone()
    bad = 0
    for it in range(300):
        y = run()
        if it % 25 == 24 or it < 3:
            bad += 0 if torch.equal(y, ref) else 1
        del y
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of the compared launches differ"
    if bf16:
        torch.testing.assert_close(ref, unsplit, rtol=1e-5, atol=1e-5 * float(unsplit.abs().max()))
