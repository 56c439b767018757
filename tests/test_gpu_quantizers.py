"""GPU parity, through the C ABI (ctypes -> libmi355q.so): HIP quantisers vs the golden vectors
captured from the reference and vs the numpy oracle.  Bit-exact on every integer and on the
fake-quantised fp32 values."""
import json

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
_META = json.loads((GOLDEN / "quantizers.json").read_text())


def _dev():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def _blocks_view(arr, shape, block_size, skip):
    """tensor-shaped per-element array -> oracle block order [n_blocks, block_elems]"""
    from oracle import np_oracle as O
    meta = O.block_meta(shape, block_size, skip)
    return O.to_blocks(np.asarray(arr, dtype=np.float32), meta)


@pytest.mark.parametrize("tag", sorted(_META))
def test_quantizer_golden(tag, golden_quantizers):
    import torch
    from mi355q import ops
    meta, data = golden_quantizers
    m = meta[tag]
    p, skip = dict(m["params"]), m["skip_first_dim"]
    x = torch.from_numpy(data[f"{tag}/x"]).to(_dev())
    y_ref = data[f"{tag}/y"]
    if m["quantizer"] == "block_fp":
        y, mant, exp = ops.block_fp_quantize(x, p["width"], p["exponent_width"], p["exponent_bias"],
                                             p["block_size"], skip, want_packed=True)
        torch.cuda.synchronize()
        assert _same(y.cpu().numpy(), y_ref), f"fake-quant differs, max {np.nanmax(np.abs(y.cpu().numpy() - y_ref))}"
        bias = 2 ** (p["exponent_width"] - 1) - 1 if p["exponent_bias"] is None else p["exponent_bias"]
        assert _same(exp.cpu().numpy().astype(np.int32) - bias, data[f"{tag}/exp"]), "shared exponents differ"
        got = _blocks_view(mant.cpu().numpy(), x.shape, p["block_size"], skip).astype(np.int32)
        ref = data[f"{tag}/mant"].astype(np.int32)
        # padded tail elements of ragged blocks exist only in the reference's padded copy
        pad = _blocks_view(np.ones(x.shape, np.float32), x.shape, p["block_size"], skip) == 0
        assert _same(got[~pad], ref[~pad]), "mantissas differ"
        # y-only call takes the single-launch path and must agree
        y2 = ops.block_fp_quantize(x, p["width"], p["exponent_width"], p["exponent_bias"], p["block_size"], skip)
        assert _same(y2.cpu().numpy(), y_ref)
    elif m["quantizer"] == "block_minifloat":
        y, bias = ops.block_minifloat_quantize(x, p["width"], p["exponent_width"], p["exponent_bias_width"],
                                               p["block_size"], skip, want_bias=True)
        assert _same(y.cpu().numpy(), y_ref), f"max {np.nanmax(np.abs(y.cpu().numpy() - y_ref))}"
        assert _same(bias.cpu().numpy().astype(np.int32), data[f"{tag}/bias"])
    else:
        y, bias = ops.block_log_quantize(x, p["width"], p["exponent_bias_width"], p["block_size"], skip,
                                         want_bias=True)
        assert _same(y.cpu().numpy(), y_ref), f"max {np.nanmax(np.abs(y.cpu().numpy() - y_ref))}"
        assert _same(bias.cpu().numpy().astype(np.int32), data[f"{tag}/bias"])


@pytest.mark.parametrize("shape,style", [((64, 4096), "rowscale"), ((2, 33, 1024), "outlier"), ((257, 768), "sparse")])
@pytest.mark.parametrize("fmt", ["bfp6", "bfp4", "bfp8", "bm", "bl"])
def test_quantizer_vs_oracle_medium(shape, style, fmt):
    """sizes the oracle finishes in seconds; seeded inputs; all three formats"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    r = np.random.default_rng(hash((shape, style, fmt)) % 2 ** 31)
    x = r.normal(size=shape).astype(np.float32)
    if style == "rowscale":
        x *= np.exp(2 * r.normal(size=shape[:-1] + (1,))).astype(np.float32)
    elif style == "outlier":
        x[..., ::97] *= 50
    else:
        x[r.random(shape) < 0.5] = 0
        x.reshape(-1)[: 16 * 40] = 0
    xt = torch.from_numpy(x).to(_dev())
    if fmt.startswith("bfp"):
        w = int(fmt[3:])
        y, mant, exp = ops.block_fp_quantize(xt, w, 8, 127, [1, 16], True, want_packed=True)
        code = O.bfp_encode(x, w, 8, 127, [1, 16], True)
        assert _same(exp.cpu().numpy().astype(np.int32) - 127, code.exp)
        assert _same(mant.cpu().numpy().reshape(-1, 16).astype(np.int32), code.mant)
        assert _same(y.cpu().numpy(), O.block_fp_quantize(x, w, 8, 127, [1, 16], True))
    elif fmt == "bm":
        xs = x * 37
        y = ops.block_minifloat_quantize(torch.from_numpy(xs).to(_dev()), 8, 4, 8, [1, 16], True)
        assert _same(y.cpu().numpy(), O.block_minifloat_quantize(xs, 8, 4, 8, [1, 16], True))
    else:
        y = ops.block_log_quantize(xt, 8, 8, [1, 16], True)
        assert _same(y.cpu().numpy(), O.block_log_quantize(x, 8, 8, [1, 16], True))


def test_fast_zero_block_mode_only_touches_zero_blocks():
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    r = np.random.default_rng(5)
    x = r.normal(size=(32, 256)).astype(np.float32) * 1e-3
    x[3, 32:64] = 0
    x[9, :] = 0
    xt = torch.from_numpy(x).to(_dev())
    y, mant, exp = ops.block_fp_quantize(xt, 6, 8, 127, [1, 16], True, want_packed=True, fast_zero_blocks=True)
    code = O.bfp_encode(x, 6, 8, 127, [1, 16], True)
    zero = np.abs(x).reshape(-1, 16).max(1) == 0
    e = exp.cpu().numpy().astype(np.int32) - 127
    assert _same(e[~zero], code.exp[~zero]) and np.all(e[zero] == 0)
    mg = mant.cpu().numpy().reshape(-1, 16).astype(np.int32)
    assert _same(mg[~zero], code.mant[~zero]) and np.all(mg[zero] == 0)
    assert _same(y.cpu().numpy(), O.block_fp_quantize(x, 6, 8, 127, [1, 16], True))


def test_cpu_tensor_is_refused():
    import torch
    from mi355q import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.block_fp_quantize(torch.zeros(4, 16), 6, 8, 127, [1, 16], True)


@pytest.mark.parametrize("shape,skip", [((300, 1024), True), ((3, 50, 64), True), ((64, 48), False), ((7, 40), True)])
@pytest.mark.parametrize("width", [4, 6, 9])
def test_block_fp_quantize_bf16_equals_fp32_quantiser_cast(shape, skip, width):
    """bf16 output of the block_fp quantiser (operand of the bf16 GEMM in the unaligned Linear mode): the fp32 fake-quant
    values are exact in bf16, so both routes give identical bits; |x| <= 1e-8 elements are rounded"""
    import torch
    from mi355q import ops
    r = np.random.default_rng(sum(shape) + width)
    x = (r.normal(size=shape) * np.exp(2 * r.normal(size=shape[:-1] + (1,)))).astype(np.float32)
    x[..., :16] = 0.0
    x.reshape(-1)[3] = 3e-9
    xt = torch.from_numpy(x).to("cuda:0")
    a = ops.block_fp_quantize_bf16(xt, width, 8, 127, [1, 16], skip)
    b = ops.block_fp_quantize(xt, width, 8, 127, [1, 16], skip)
    assert a.dtype == torch.bfloat16 and a.shape == xt.shape
    assert torch.equal(a, b.to(torch.bfloat16))
    big = b.abs() > 1e-8
    assert torch.equal(a.float()[big], b[big])               # quantised values survive the format exactly


@pytest.mark.parametrize("bias,ew", [(-3, 4), (-20, 8), (0, 3), (None, 5)])
def test_block_fp_explicit_bias_is_literal(bias, ew):
    """a given exponent_bias, negative ones included, is used as it is (block_fp.py:61-62: e in [-bias, 2^ew-1-bias]);
    only None selects the default"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    r = np.random.default_rng(3)
    x = (r.normal(size=(37, 96)) * np.exp(3 * r.normal(size=(37, 1)))).astype(np.float32)
    y = ops.block_fp_quantize(torch.from_numpy(x).to(_dev()), 6, ew, bias, [1, 16], True)
    assert _same(y.cpu().numpy(), O.block_fp_quantize(x, 6, ew, bias, [1, 16], True))


def test_elementwise_quantizers_bit_exact():
    """QUANTIZER_MAP["minifloat_ieee" | "minifloat_denorm" | "log"] on the HIP path == the reference's outputs
    (tests/golden/elementwise.npz, generated by importing the reference), bit for bit, and == the oracle on a large tensor"""
    import json
    from pathlib import Path
    import numpy as np
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    g = Path(__file__).parent / "golden"
    z, cases = np.load(g / "elementwise.npz"), json.loads((g / "elementwise.json").read_text())
    for tag, c in cases.items():
        x = torch.from_numpy(z[f"x/{c['input']}"]).to("cuda:0")
        got = Q.get_quantizer("", {"name": c["quantizer"]})(x, **c["params"]).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), z[f"y/{tag}"].view(np.uint32)), tag
    from mi355q.quantize.quantizers import QUANTIZER_MAP
    r = np.random.default_rng(3)
    big = (r.normal(size=(1031, 517)) * np.exp(r.normal(size=(1031, 1)) * 3)).astype(np.float32)
    bt = torch.from_numpy(big).to("cuda:0")[:, 1:]            # (an unaligned, non-contiguous view)
    for name, f, kw in (("minifloat_ieee", O.minifloat_ieee_quantize, dict(width=8, exponent_width=4, exponent_bias=None)),
                        ("minifloat_denorm", O.minifloat_denorm_quantize, dict(width=8, exponent_width=4, exponent_bias=7)),
                        ("log", O.log_quantize, dict(width=8, exponent_bias=None))):
        got = QUANTIZER_MAP[name](bt, **kw).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), f(big[:, 1:], **kw).view(np.uint32)), name
    # straight-through backward (minifloat.py:100-103, log.py:66-69)
    xg = torch.randn(64, 33, device="cuda:0", requires_grad=True)
    QUANTIZER_MAP["minifloat_ieee"](xg, 8, 4, None).sum().backward()
    assert torch.equal(xg.grad, torch.ones_like(xg))


@pytest.mark.parametrize("rows,K,w,ew,ebw", [(300, 512, 8, 4, 8), (64, 2752, 8, 4, 8), (17, 96, 6, 3, 4), (128, 256, 9, 1, 8)])
def test_block_minifloat_straight_into_tiled_bf16(rows, K, w, ew, ebw):
    """mi355q_block_minifloat_quantize_bf16_tiled writes the fake-quantiser's values (zero blocks, tiny and huge magnitudes
    included): the tile GEMM of that operand against a tiled identity returns them bit for bit"""
    import torch
    from mi355q import ops
    g = torch.Generator().manual_seed(rows + K)
    x = (torch.randn(rows, K, generator=g) * torch.exp(3 * torch.randn(rows, 1, generator=g))).to("cuda:0")
    x[1, :32] = 0
    x[2, 16:32] = torch.tensor([0.0, -0.0, 1e-9, -1e-9, 1e-8, 5e-9, 1e-30, 3e38, -3e38, 0.5, 2.0, 255.0, 256.0, 1.999, 4.0, -7.5])
    fq = ops.block_minifloat_quantize(x, w, ew, ebw, [1, 16], False)
    eye = ops.bf16_tile(torch.eye(K, device="cuda:0"))
    got = ops.bf16_gemm_tiled(ops.block_minifloat_quantize_bf16_tiled(x.contiguous(), w, ew, ebw), eye, rows, K, K)
    two = ops.bf16_gemm_tiled(ops.bf16_tile(fq.contiguous()), eye, rows, K, K)
    # (|x| <= 1e-8 passes through the quantiser unquantised and is rounded to bf16 in the operand: compare the operands'
    #  images, and the image with the fake-quantised values wherever those are bf16 numbers)
    assert torch.equal(got, two)
    exact = fq.to(torch.bfloat16).float() == fq
    assert torch.equal(got[exact], fq[exact]) and exact.float().mean().item() > 0.99


@pytest.mark.parametrize("name", ["block_log", "block_fp", "block_minifloat"])
def test_many_all_zero_blocks_like_causal_probabilities(name):
    """a causal attention-probability tensor: half of its [1,16] blocks are exactly zero (the fix-up pass rewrites them with
    the tensor-global fill, block_fp.py:54-58 -- for block_log a non-zero value): == the oracle, incl. -0.0 elements"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    T = 272
    g = torch.Generator().manual_seed(0)
    p = torch.softmax(torch.randn(3, T, T, generator=g) * 3 + torch.full((T, T), float("-inf")).triu(1), dim=-1)
    p[0, 5, :16] = -0.0
    x = p.to("cuda:0")
    kw = {"block_log": dict(width=8, exponent_bias_width=8, block_size=[1, 16]),
          "block_fp": dict(width=6, exponent_width=8, exponent_bias=None, block_size=[1, 16]),
          "block_minifloat": dict(width=8, exponent_width=4, exponent_bias_width=8, block_size=[1, 16])}[name]
    got = Q.get_quantizer("", dict(name=name))(x, **kw, skip_first_dim=True).cpu().numpy()
    want = getattr(O, name + "_quantize")(p.numpy(), **kw, skip_first_dim=True)
    assert np.array_equal(got.view(np.uint32), np.asarray(want, dtype=np.float32).view(np.uint32))
    assert (p == 0).float().mean().item() > 0.4


def test_zero_block_map_path_on_a_large_tensor():
    """tensors of 128 MiB and more take the zero-block map (kernel 1 leaves the all-zero blocks to the fix-up pass, which
    finds them through the map instead of reading x again): block_log on causal probabilities [17, 2048, 1024] == the same
    rows quantised in small pieces through the re-reading pass, given the same tensor-global fill"""
    import torch
    import mi355q.quantize as Q
    g = torch.Generator().manual_seed(1)
    T = 1024
    p = torch.softmax(torch.randn(17, 2048, T, generator=g) * 3 + torch.full((2048, T), float("-inf")).triu(1), dim=-1).to("cuda:0")
    assert p.numel() >= 1 << 25
    q = Q.get_quantizer("", dict(name="block_log"))
    kw = dict(width=8, exponent_bias_width=8, block_size=[1, 16], skip_first_dim=True)
    big = q(p, **kw)
    # the fill is the smallest non-zero block maximum of the WHOLE tensor: give every piece a row that carries it
    bm = p.view(-1, 16).abs().amax(1)
    row = int((bm == bm[bm > 0].min()).nonzero()[0, 0]) * 16 // T
    carrier = p.view(-1, T)[row:row + 1]
    for i in (0, 5, 16):
        piece = torch.cat([p[i], carrier], 0).unsqueeze(0)
        small = q(piece, **kw)[0, :-1]
        assert torch.equal(small.view(torch.int32), big[i].view(torch.int32)), i


@pytest.mark.parametrize("name", ["block_log", "block_minifloat", "block_fp"])
def test_zero_block_fill_speculation_hits_and_misses(name):
    """kernel 1 writes all-zero blocks with the fill of the LAST tensor that had any (a workspace word), the fix-up launch
    rewrites them only if that was not this tensor's: sequences where the guess is right (the same tensor again), wrong
    (another tensor's smallest block maximum, another format's parameters in between) and absent (tensors without zero
    blocks in between) -- every result word for word the oracle's"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    kw = {"block_log": dict(width=8, exponent_bias_width=8, block_size=[1, 16]),
          "block_fp": dict(width=6, exponent_width=8, exponent_bias=None, block_size=[1, 16]),
          "block_minifloat": dict(width=8, exponent_width=4, exponent_bias_width=8, block_size=[1, 16])}[name]
    g = torch.Generator().manual_seed(3)
    T = 96
    causal = torch.softmax(torch.randn(2, T, T, generator=g) * 3 + torch.full((T, T), float("-inf")).triu(1), dim=-1)
    big = torch.randn(4, 64, 64, generator=g) * 50.0
    big[:, ::3, 16:48] = 0.0                                   # zero blocks, smallest block maximum ~ 10
    tiny = big * 2.0 ** -40
    dense = torch.randn(3, 32, 64, generator=g)                # no zero block at all
    mixed_sign = causal.clone()
    mixed_sign[0, 0, 16:32] = -0.0
    q = Q.get_quantizer("", dict(name=name))
    other = Q.get_quantizer("", dict(name="block_log" if name != "block_log" else "block_minifloat"))
    okw = dict(width=8, exponent_bias_width=8, block_size=[1, 16]) if name != "block_log" else \
        dict(width=8, exponent_width=4, exponent_bias_width=8, block_size=[1, 16])
    for step, t in enumerate([causal, causal, big, big, tiny, dense, tiny, causal, mixed_sign, big, "other", causal, causal]):
        if isinstance(t, str):
            other(big.to("cuda:0"), **okw, skip_first_dim=True)    # another format leaves ITS fill behind
            continue
        got = q(t.to("cuda:0"), **kw, skip_first_dim=True).cpu().numpy()
        want = np.asarray(getattr(O, name + "_quantize")(t.numpy(), **kw, skip_first_dim=True), dtype=np.float32)
        # (word for word, except the SIGN of a zero output: the reference's own depends on the blocking path its tensor took
        #  -- -0.0 for 2-D activations, +0.0 for 3-D ones and weights in the fixtures -- and is not pinned by the oracle)
        gw, ww = got.view(np.uint32), want.view(np.uint32)
        assert np.array_equal(gw | np.where(got == 0, np.uint32(0x80000000), np.uint32(0)),
                              ww | np.where(want == 0, np.uint32(0x80000000), np.uint32(0))), (name, step)


@pytest.mark.parametrize("rows,K,P,relu", [(300, 4096, 8, False), (256, 2048, 4, True), (256, 2048, 4, False), (70, 11008, 8, False),
                                            (512, 1024, 1, False)])
def test_aligned_rows_quantiser_reads_row_segments_in_place(rows, K, P, relu):
    """mi355q_block_fp_quantize_aligned_rows_seg: x as the rank-major [P, rows, K / P] result of an all-gather over
    out_features shards == the plain call on the re-assembled [rows, K] tensor, every output buffer byte for byte"""
    import torch
    from mi355q import ops
    dev = _dev()
    g = torch.Generator().manual_seed(rows + K)
    x = (torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, 1, generator=g))).to(dev)
    x[::5, 64:80] *= 2.0 ** -12                         # blocks below their row's window: exception entries
    seg = x.view(rows, P, K // P).permute(1, 0, 2).contiguous()          # what the collective leaves
    pre = ("relu", None) if relu else None
    w = torch.randn(256, K, generator=g).to(dev) * 0.05
    _, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    prev = ops.REUSE_QUANTISED_INPUT
    ops.REUSE_QUANTISED_INPUT = False
    try:
        a = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127, pre=pre)
        keep = [t.clone() for t in (a.tiled, a.exp, a.rowflag, a.gscale)]
        ya = ops.bfp_gemm_aligned(a, wa).clone()
        c = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127, pre=pre)
        again = [t.clone() for t in (c.tiled, c.exp, c.rowflag, c.gscale)]
        b = ops.block_fp_quantize_aligned_rows(seg, 6, 8, 127, pre=pre, segments=True)
        yb = ops.bfp_gemm_aligned(b, wa)
    finally:
        ops.REUSE_QUANTISED_INPUT = prev
    # (rows of one 256-row bucket take their exception slots in arrival order and settle for a wider window when the bucket
    #  runs full: with contended buckets the operand's bytes differ from call to call -- never the product -- so the bytes
    #  are compared whenever two plain calls agree on them)
    if all(torch.equal(u, v) for u, v in zip(keep, again)):
        for name, u, v in zip(("tiled", "exp", "rowflag", "rowscale"), keep, (b.tiled, b.exp, b.rowflag, b.gscale)):
            assert torch.equal(u, v), name
        assert torch.equal(ya, yb)
    else:
        assert relu
    assert float((ya - yb).abs().max()) <= 2e-6 * float(ya.abs().max())
