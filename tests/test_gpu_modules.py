"""GPU parity of the registry-level API (Linear classes, matmul/bmm functions, quantiser autograd
wrappers) against golden outputs of the reference's own modules (tests/golden/modules.npz)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TAGS = ["bfp_6bit", "bfp_4bit", "block_fp", "block_minifloat", "block_log"]


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("has_bias", [1, 0])
def test_linear_ptq_golden(tag, has_bias, golden_modules):
    import torch
    import mi355q.quantize as Q
    meta, data = golden_modules
    cfg = meta[tag]["linear_config"]
    k = f"{tag}/linear_bias{has_bias}"
    fp = torch.nn.Linear(96, 48, bias=bool(has_bias))
    with torch.no_grad():
        fp.weight.copy_(torch.from_numpy(data[f"{k}/w"]))
        if has_bias:
            fp.bias.copy_(torch.from_numpy(data[f"{k}/b"]))
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    y1 = lin(_t(data[f"{k}/x1"]))
    # first PTQ call overwrote the parameters with their quantised values, bit for bit
    assert np.array_equal(lin.weight.detach().cpu().numpy(), data[f"{k}/wq"])
    if has_bias:
        assert np.array_equal(lin.bias.detach().cpu().numpy(), data[f"{k}/bq"])
    assert lin.weight_requires_quantisation is False
    y2 = lin(_t(data[f"{k}/x2"]))
    for y, ref in ((y1, data[f"{k}/y1"]), (y2, data[f"{k}/y2"])):
        assert y.shape == ref.shape
        got = y.detach().cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
        # far tighter than the 1e-3 contract
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5 * np.abs(ref).max())
    # in_features = 96 is not a multiple of 64: this layer takes the fake-quant + fp32 GEMM route
    assert lin._packed is None


@pytest.mark.parametrize("wx,ww", [(6, 6), (4, 4), (8, 6)])
@pytest.mark.parametrize("has_bias", [True, False])
def test_linear_int8_path_vs_oracle(wx, ww, has_bias):
    """in_features % 64 == 0: quantise+pack -> align -> int8-MFMA GEMM; 3-D input; in-place weight
    overwrite bit-exact; output vs the oracle's exact integer contraction"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=wx, data_in_exponent_width=8,
               data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=ww, weight_exponent_width=8,
               weight_exponent_bias=None, weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8,
               bias_exponent_bias=127, bias_block_size=[16])
    torch.manual_seed(3)
    fp = torch.nn.Linear(320, 200, bias=has_bias)
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    w0 = fp.weight.detach().numpy().copy()
    b0 = fp.bias.detach().numpy().copy() if has_bias else None
    x = torch.randn(2, 37, 320) * torch.exp(torch.randn(2, 37, 1))
    x[..., 100:104] *= 400.0                 # outlier channels: unaligned row-groups -> sparse correction
    ocfg = dict(cfg, weight_exponent_bias=127)
    for call in range(2):
        y = lin(x.to("cuda:0"))
        assert lin._packed is not None, "int8 MFMA path was not taken"
        ref = O.bfp_linear_int(x.numpy().reshape(-1, 320), w0, b0, ocfg).reshape(2, 37, 200)
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())
    _, wq, bq = O.linear_ptq(x.numpy().reshape(-1, 320), w0, b0, ocfg)
    assert np.array_equal(lin.weight.detach().cpu().numpy(), wq)
    if has_bias:
        assert np.array_equal(lin.bias.detach().cpu().numpy(), bq)


@pytest.mark.parametrize("K", [48, 80, 144, 1104])
def test_linear_block_fp_with_in_features_not_a_multiple_of_64(K):
    """in_features a multiple of the block but not of the tile kernels' K-step: the contraction is padded with all-zero
    blocks and runs on the bf16 tile GEMM (no library GEMM); same weights / bias overwrite, output vs the oracle"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    from oracle import np_oracle as O
    cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8,
               data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=5, weight_exponent_width=8,
               weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8,
               bias_exponent_bias=127, bias_block_size=[16])
    torch.manual_seed(K)
    fp = torch.nn.Linear(K, 112, bias=True)
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    w0, b0 = fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()
    x = torch.randn(3, 37, K) * torch.exp(torch.randn(3, 37, 1))
    calls, real = [], ops.bf16_gemm_tiled
    ops.bf16_gemm_tiled = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    try:
        for call in range(2):
            y = lin(x.to("cuda:0"))
            ref = O.bfp_linear_int(x.numpy().reshape(-1, K), w0, b0, cfg).reshape(3, 37, 112)
            np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())
        assert len(calls) == 2, "the padded tile-GEMM route was not taken"
        off = Q.get_quantized_cls("linear", dict(cfg, mi355q_pad_k=False)).from_float(fp, dict(cfg, mi355q_pad_k=False)).to("cuda:0")
        y2 = off(x.to("cuda:0"))
        assert len(calls) == 2
        np.testing.assert_allclose(y2.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=0, atol=4e-6 * np.abs(ref).max())
    finally:
        ops.bf16_gemm_tiled = real
    _, wq, bq = O.linear_ptq(x.numpy().reshape(-1, K), w0, b0, cfg)
    assert np.array_equal(lin.weight.detach().cpu().numpy(), wq) and np.array_equal(lin.bias.detach().cpu().numpy(), bq)


@pytest.mark.parametrize("align", ["auto", "rows", "rows_post", "blocks"])
@pytest.mark.parametrize("outliers", [False, True])
def test_linear_int8_align_modes(align, outliers):
    """the exponent-alignment flavour of the packed operands is an implementation knob: every choice gives the
    oracle's result; "auto" takes whole rows when weights and first activations fit them, with the activations'
    exceptions in the GEMM's LDS add-back while a tile's share fits it and through the row post-pass otherwise"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8,
               data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8,
               weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8,
               bias_exponent_bias=127, bias_block_size=[16], mi355q_align=align)
    torch.manual_seed(5)
    fp = torch.nn.Linear(512, 192, bias=True)
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    w0, b0 = fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()
    x = torch.randn(3, 100, 512) * torch.exp(torch.randn(3, 100, 1))
    if outliers:
        x[..., 100:104] *= 400.0             # an outlier channel: one exception block in EVERY row
    for call in range(3):
        y = lin(x.to("cuda:0"))
        ref = O.bfp_linear_int(x.numpy().reshape(-1, 512), w0, b0, cfg).reshape(3, 100, 192)
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())
    assert lin._align_mode == "rows"
    if align == "auto":
        # 300 rows x 1 exception: too many for a tile's LDS add-back.  Round 6: the outlier block column becomes class 1 of the
        # mixed contraction (the rest of the row fits its window again: int8 MFMA for class 0, bf16 MFMA for class 1, one launch);
        # before: no alignment at all, the whole layer on the bf16 flavour
        assert lin._x_cap == 120 and (lin._mixed is not None) == outliers
    if align == "rows_post":
        assert lin._x_cap == 1016
    if align == "blocks":                    # nothing aligned: bf16 GEMM on the exactly representable quantised values
        assert lin._x_cap == -1 and lin._w_bf16 is not None
        cfg2 = dict(cfg, mi355q_blocks_gemm="int8")          # ... or the blockwise-exact int8 kernel
        lin2 = Q.get_quantized_cls("linear", cfg2).from_float(fp, cfg2).to("cuda:0")
        y2 = lin2(x.to("cuda:0"))
        assert lin2._w_bf16 is None
        np.testing.assert_allclose(y2.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())


def test_linear_auto_align_leaves_row_mode_when_activations_stop_fitting():
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8,
               data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8,
               weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8,
               bias_exponent_bias=127, bias_block_size=[16])
    torch.manual_seed(6)
    fp = torch.nn.Linear(256, 64, bias=False)
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    w0 = fp.weight.detach().numpy().copy()
    calm = torch.randn(300, 256)
    wild = calm.clone()
    wild[:, 32:36] *= 500.0
    wilder = calm.clone()
    for c0 in (32, 80, 128, 176, 224):       # five far-off blocks per row: 1280 entries in rows 0..255 > 1016
        wilder[:, c0:c0 + 4] *= 500.0
    lin(calm.to("cuda:0"))
    assert lin._align_mode == "rows" and lin._x_cap == 120
    for call in range(3):                    # overflow of the 120-entry bucket seen at calls 2 and 4 -> no alignment
        y = lin(wild.to("cuda:0"))
        ref = O.bfp_linear_int(wild.numpy(), w0, None, cfg)
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())
    assert lin._align_mode == "rows" and lin._x_cap == -1
    for call in range(3):
        y = lin(wilder.to("cuda:0"))
        ref = O.bfp_linear_int(wilder.numpy(), w0, None, cfg)
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())
    y = lin(calm.to("cuda:0"))
    ref = O.bfp_linear_int(calm.numpy(), w0, None, cfg)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("op", ["bmm0", "bmm1", "mm4d", "mm2d"])
def test_matmul_golden(tag, op, golden_modules):
    import mi355q.quantize as Q
    meta, data = golden_modules
    cfg = meta[tag]["matmul_config"]
    f = Q.get_quantized_func("bmm" if op.startswith("bmm") else "matmul", cfg)
    out = f(_t(data[f"{tag}/{op}/x"]), _t(data[f"{tag}/{op}/y"]), cfg)
    ref = data[f"{tag}/{op}/out"]
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-3, atol=2e-5 * np.abs(ref).max())


def test_qat_forward_backward_ste():
    """is_ptq = False: all three operands re-quantised every call, identity backward (reference
    linear.py:72-76, block_fp.py:119-124)"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = dict(name="block_fp", is_ptq=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
               data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
               weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127,
               bias_block_size=[16])
    torch.manual_seed(0)
    lin = Q.get_quantized_cls("linear", cfg)(64, 32, config=cfg).to("cuda:0")
    w0 = lin.weight.detach().clone()
    x = torch.randn(5, 64, device="cuda:0", requires_grad=True)
    y = lin(x)
    y.sum().backward()
    assert torch.equal(lin.weight.detach(), w0), "QAT must not overwrite the weights"
    ref, wq, bq = O.linear_ptq(x.detach().cpu().numpy(), w0.cpu().numpy(), lin.bias.detach().cpu().numpy(), cfg)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    # STE: dL/dx = 1^T W_q, dL/dW = 1 x_q^T
    np.testing.assert_allclose(x.grad.cpu().numpy(), np.tile(wq.sum(0), (5, 1)), rtol=1e-4, atol=1e-5)
    xq = O.block_fp_quantize(x.detach().cpu().numpy(), 6, 8, 127, [1, 16], True)
    np.testing.assert_allclose(lin.weight.grad.cpu().numpy(), np.tile(xq.sum(0), (32, 1)), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("arith", ["block_fp", "block_minifloat", "block_log"])
def test_qat_on_the_tile_gemm_gradients(arith):
    """is_ptq = False at an OPT-350m layer shape (fc1: 1024 -> 4096, 512 tokens): forward and both backward products on the bf16
    tile GEMM (quantised operands exact in bf16; the gradient dY as two bf16 planes) against the oracle's quantisers and
    float64 products -- reference: autograd through F.linear(x_q, W_q, b_q), quantized_modules/linear.py:72-76, with the
    quantisers' straight-through estimators (block_fp.py:119-124); mi355q_qat_gemm = "fp32" gives the library route and the
    same numbers to fp32 accuracy"""
    import torch
    import mi355q.quantize as Q
    from mi355q.quantize.quantized_modules import linear as L
    from oracle import np_oracle as O
    cfg6 = dict(name="block_fp", bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
                weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=6,
                bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
    cfg = {"block_fp": dict(cfg6, is_ptq=False),
           "block_minifloat": dict(name="block_minifloat", is_ptq=False, bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias_width=8,
                                   data_in_block_size=[1, 16], weight_width=8, weight_exponent_width=4, weight_exponent_bias_width=8, weight_block_size=[1, 16],
                                   bias_width=8, bias_exponent_width=4, bias_exponent_bias_width=8, bias_block_size=[16]),
           "block_log": dict(name="block_log", is_ptq=False, bypass=False, data_in_width=8, data_in_exponent_bias_width=8, data_in_block_size=[1, 16],
                             weight_width=8, weight_exponent_bias_width=8, weight_block_size=[1, 16], bias_width=8, bias_exponent_bias_width=8,
                             bias_block_size=[16])}[arith]
    M, K, N = 512, 1024, 4096
    torch.manual_seed(3)
    scale = 4.0 if arith == "block_minifloat" else 1.0           # (block_minifloat flushes |w| << 1 to zero: SURVEY quirk 5)
    cfg = dict(cfg, mi355q_qat_gemm="bf16_always")               # (the default takes the tile GEMM from 2^34 multiply-adds on)
    lin = Q.get_quantized_cls("linear", cfg)(K, N, config=cfg).to("cuda:0")
    with torch.no_grad():
        lin.weight.mul_(32.0 * scale)
    x = (torch.randn(M, K, device="cuda:0") * scale).requires_grad_(True)
    dy = torch.randn(M, N, device="cuda:0")
    calls, real = [], L.ops.bf16_gemm_tiled
    L.ops.bf16_gemm_tiled = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        y = lin(x)
        y.backward(dy)
    finally:
        L.ops.bf16_gemm_tiled = real
    assert len(calls) == 5, len(calls)                           # forward + 2 planes x (dX, dW)
    quant = {"block_fp": lambda t, p, skip: O.block_fp_quantize(t, cfg[p + "_width"], 8, 127, cfg[p + "_block_size"], skip),
             "block_minifloat": lambda t, p, skip: O.block_minifloat_quantize(t, 8, 4, 8, cfg[p + "_block_size"], skip),
             "block_log": lambda t, p, skip: O.block_log_quantize(t, 8, 8, cfg[p + "_block_size"], skip)}[arith]
    xq = quant(x.detach().cpu().numpy(), "data_in", True).astype(np.float64)
    wq = quant(lin.weight.detach().cpu().numpy(), "weight", False).astype(np.float64)
    bq = quant(lin.bias.detach().cpu().numpy(), "bias", False).astype(np.float64)
    g = dy.cpu().numpy().astype(np.float64)
    for name, got, ref in (("y", y, xq @ wq.T + bq), ("dX", x.grad, g @ wq), ("dW", lin.weight.grad, g.T @ xq), ("db", lin.bias.grad, g.sum(0))):
        err = float(np.abs(got.detach().cpu().numpy() - ref).max() / np.abs(ref).max())
        assert err < 2e-5, (arith, name, err)
    # the library route (F.linear forward and backward) on the same module: same numbers to fp32 accuracy
    lin2 = Q.get_quantized_cls("linear", dict(cfg, mi355q_qat_gemm="fp32"))(K, N, config=dict(cfg, mi355q_qat_gemm="fp32")).to("cuda:0")
    lin2.load_state_dict(lin.state_dict())
    x2 = x.detach().clone().requires_grad_(True)
    lin2(x2).backward(dy)
    assert float((x2.grad - x.grad).abs().max() / x.grad.abs().max()) < 2e-5
    assert float((lin2.weight.grad - lin.weight.grad).abs().max() / lin.weight.grad.abs().max()) < 2e-5


def test_requantize_after_weight_reload():
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = dict(name="block_fp", is_ptq=True, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
               data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
               weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127,
               bias_block_size=[16])
    torch.manual_seed(1)
    lin = Q.get_quantized_cls("linear", cfg)(64, 48, config=cfg).to("cuda:0")
    x = torch.randn(7, 64, device="cuda:0")
    lin(x)
    w_new = torch.randn(48, 64) * 0.05
    with torch.no_grad():
        lin.weight.copy_(w_new.to("cuda:0"))
    # like the reference, a reload after the first call is used as is (not re-quantised) ...
    y_raw = lin(x)
    xq = O.block_fp_quantize(x.cpu().numpy(), 6, 8, 127, [1, 16], True)
    np.testing.assert_allclose(y_raw.detach().cpu().numpy(), xq @ w_new.numpy().T + lin.bias.detach().cpu().numpy(),
                               rtol=1e-4, atol=1e-5)
    # ... until requantize() asks for a fresh quantise + pack
    lin.requantize()
    y = lin(x)
    ref, wq, _ = O.linear_ptq(x.cpu().numpy(), w_new.numpy(), None, cfg)
    assert np.array_equal(lin.weight.detach().cpu().numpy(), wq)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref + lin.bias.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_rope_integer_quantised_tables():
    """every shipped TOML quantises the RoPE tables with `integer` (bfp_6bit.toml:18-22)"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = dict(name="integer", data_in_width=8, data_in_frac_width=7)
    f = Q.get_quantized_func("rotary_positional_encoding", cfg)
    T, hd = 12, 16
    pos = torch.arange(T)[:, None] * (1.0 / 10000 ** (torch.arange(0, hd, 2) / hd))[None, :]
    emb = torch.cat((pos, pos), -1)
    cos, sin = emb.cos()[None, None].to("cuda:0"), emb.sin()[None, None].to("cuda:0")
    q = torch.randn(1, 2, T, hd, device="cuda:0")
    k = torch.randn(1, 2, T, hd, device="cuda:0")
    ids = torch.arange(T, device="cuda:0")[None]
    qe, ke = f(q, k, cos, sin, ids, cfg)
    cq = O.integer_quantize(cos.cpu().numpy()[0, 0], 8, 7)
    sq = O.integer_quantize(sin.cpu().numpy()[0, 0], 8, 7)
    qn = q.cpu().numpy()
    rot = np.concatenate((-qn[..., hd // 2:], qn[..., : hd // 2]), -1)
    np.testing.assert_allclose(qe.cpu().numpy(), qn * cq + rot * sq, rtol=1e-6, atol=1e-6)


def _mm_cfg(wx, wy, fused=True):
    return dict(name="block_fp", bypass=False, data_in_width=wx, data_in_exponent_width=8, data_in_exponent_bias=127,
                data_in_block_size=[1, 16], weight_width=wy, weight_exponent_width=8, weight_exponent_bias=127,
                weight_block_size=[1, 16], mi355q_fused_matmul=fused)


@pytest.mark.parametrize("shape", [((3, 100, 256), (3, 256, 64)),      # probs x V: long contraction, head_dim columns
                                   ((3, 100, 64), (3, 64, 272)),       # Q x K^T: one K step, many column chunks
                                   ((2, 33, 48), (2, 48, 16)),         # K % 128 != 0, ragged rows
                                   ((1, 16, 400), (1, 400, 80)),       # several steps + a partial one, partial chunk
                                   ((2, 2, 5, 128), (2, 2, 128, 32)),  # 4-D operands (flattened leading dims)
                                   ((40, 144), (144, 48))])            # 2-D operands
@pytest.mark.parametrize("wx,wy", [(6, 6), (4, 8), (8, 4)])
def test_fused_block_fp_matmul_vs_oracle_and_two_step(shape, wx, wy):
    """quantise-in-registers + bf16 MFMA product (mi355q_bfp_matmul) against the oracle's restatement of
    matmul.py:146-196 and against the two-quantisers + GEMM route of the same registry function"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    r = np.random.default_rng(sum(shape[0]) + wx)
    x = (r.normal(size=shape[0]) * np.exp(r.normal(size=shape[0][:-1] + (1,)))).astype(np.float32)
    y = r.normal(size=shape[1]).astype(np.float32)
    x[..., :16] = 0.0                                   # all-zero blocks
    x.reshape(-1)[5] = 3e-9                             # |x| <= 1e-8 passes through unquantised
    if x.ndim == 3:
        x[0] = np.maximum(x[0], 0)                      # softmax-like: many exact zeros
    style = "matmul" if x.ndim != 3 else "bmm"
    fused = Q.get_quantized_func(style, _mm_cfg(wx, wy))(_t(x), _t(y), _mm_cfg(wx, wy))
    plain = Q.get_quantized_func(style, _mm_cfg(wx, wy, False))(_t(x), _t(y), _mm_cfg(wx, wy, False))
    ref = O.matmul_quantized(x, y, _mm_cfg(wx, wy))
    scale = np.abs(ref).max() + 1e-30
    assert fused.shape == plain.shape == ref.shape
    np.testing.assert_allclose(fused.cpu().numpy(), ref, rtol=0, atol=2e-6 * scale * max(1, shape[0][-1] // 64))
    np.testing.assert_allclose(fused.cpu().numpy(), plain.cpu().numpy(), rtol=0, atol=2e-6 * scale * max(1, shape[0][-1] // 64))


def test_fused_block_fp_matmul_falls_back_where_it_does_not_apply():
    """odd sizes, broadcasting, wide mantissas and autograd keep the two-quantisers route (same results as before)"""
    import torch
    import mi355q.quantize as Q
    from mi355q.quantize import quantized_functions as F_
    from oracle import np_oracle as O
    r = np.random.default_rng(9)
    for xs, ys, wx in (((2, 7, 24), (2, 24, 10), 6), ((3, 8, 32), (3, 32, 16), 12)):
        x, y = r.normal(size=xs).astype(np.float32), r.normal(size=ys).astype(np.float32)
        cfg = _mm_cfg(wx, 6)
        assert F_._fused_block_fp_matmul(_t(x), _t(y), cfg, "bmm") is None
        out = Q.get_quantized_func("bmm", cfg)(_t(x), _t(y), cfg)
        ref = O.matmul_quantized(x, y, cfg)
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-3, atol=2e-5 * np.abs(ref).max())
    xg = _t(r.normal(size=(2, 16, 32)).astype(np.float32)).requires_grad_(True)
    yg = _t(r.normal(size=(2, 32, 16)).astype(np.float32))
    assert F_._fused_block_fp_matmul(xg, yg, _mm_cfg(6, 6), "bmm") is None
    Q.get_quantized_func("bmm", _mm_cfg(6, 6))(xg, yg, _mm_cfg(6, 6)).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_linear_auto_takes_blocks_for_weights_with_outlier_input_channels():
    """an outlier input channel puts one exception block into EVERY weight row (256 per bucket, far beyond a tile's LDS
    add-back): auto leaves alignment alone and multiplies the exactly representable quantised values (bf16 MFMA GEMM,
    fp32 accumulation) -- same values as the oracle's exact block dots"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8,
               data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8,
               weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8,
               bias_exponent_bias=127, bias_block_size=[16])
    torch.manual_seed(8)
    fp = torch.nn.Linear(512, 320, bias=True)
    with torch.no_grad():
        fp.weight[:, 40] *= 300.0
        fp.weight[:, 333] *= 300.0
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    w0, b0 = fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()
    x = torch.randn(2, 70, 512) * torch.exp(torch.randn(2, 70, 1))
    for _ in range(2):
        y = lin(x.to("cuda:0"))
    # round 6: the two outlier block columns go to class 1 of the mixed contraction, the other thirty stay on the int8 MFMA
    assert lin._align_mode == "rows" and lin._x_cap == 120 and lin._mixed is not None
    assert {2, 20} <= set(lin._mixed["classes"].blocks1.cpu().tolist())
    ref = O.bfp_linear_int(x.numpy().reshape(-1, 512), w0, b0, cfg).reshape(2, 70, 320)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())
    # mi355q_mixed = False: as before round 6 -- no alignment, the exactly representable quantised values on the bf16 MFMA
    cfg2 = dict(cfg, mi355q_mixed=False)
    lin2 = Q.get_quantized_cls("linear", cfg2).from_float(fp, cfg2).to("cuda:0")
    for _ in range(2):
        y2 = lin2(x.to("cuda:0"))
    assert lin2._align_mode == "rows" and lin2._x_cap == -1 and lin2._w_bf16 is not None and lin2._mixed is None
    np.testing.assert_allclose(y2.detach().cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())


def _lin_cfg(width, **extra):
    return dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=width, data_in_exponent_width=8,
                data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=width, weight_exponent_width=8,
                weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=width, bias_exponent_width=8,
                bias_exponent_bias=127, bias_block_size=[16], **extra)


@pytest.mark.parametrize("width,act", [(6, "plain"), (4, "plain"), (6, "silu"), (5, "relu")])
def test_packed_weight_storage(width, act):
    """true width-bit weight storage: width + 0.5 bits per value at rest, expanded into a scratch operand
    every forward; outputs bit-identical to the int8 / bf16 resident operands, memory density vs fp32 >= 32 / (width + 0.55);
    the fp32 Parameter can be released afterwards"""
    import torch
    import mi355q.quantize as Q
    torch.manual_seed(3)
    K, N, M = 1024, 512, 300
    fp = torch.nn.Linear(K, N)
    w0, b0 = fp.weight.detach().clone().numpy(), fp.bias.detach().clone().numpy()
    h = torch.randn(M, K) * torch.exp(torch.randn(M, 1))
    x = {"plain": h, "relu": torch.relu(h), "silu": torch.nn.functional.silu(h) * torch.randn(M, K)}[act].to("cuda:0")
    ref_cfg, cfg = _lin_cfg(width), _lin_cfg(width, mi355q_weight_storage="packed")
    ref = Q.get_quantized_cls("linear", ref_cfg).from_float(fp, ref_cfg).to("cuda:0")
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    y_ref = ref(x)
    y = lin(x)
    assert lin._w_packed is not None
    if act != "relu":                                       # (post-ReLU rows fit a window of 4 exponents, W5 has one)
        assert lin._w_packed.row_scale_flavour == (act == "plain")
    assert torch.equal(y, y_ref)
    # ... and both the oracle's exact contraction of the quantised integers (not only each other)
    from oracle import np_oracle as O
    want = O.bfp_linear_int(x.cpu().numpy(), w0, b0, ref_cfg)
    np.testing.assert_allclose(y.detach().cpu().numpy(), want, rtol=0, atol=4e-6 * np.abs(want).max())
    bits = lin.weight_storage_bits()
    pw = lin._w_packed
    assert 8.0 * (pw.packed.numel() + pw.codes.numel()) / (K * N) == width + 0.5      # the format's own density
    # (+ per-row scales and the fixed-size exception buckets: 62 KB per 4096 rows, 0.03 bits at 4096 x 4096; this layer
    #  is small).  W6: 32 / 6.5 = 4.9x fp32, the README's "5x memory density" (README.md:11)
    assert bits <= width + 0.75, bits
    assert torch.equal(lin.weight, ref.weight)              # (the in-place fake-quantised fp32 values, as the reference)
    lin.release_fp32_weight()
    assert lin.weight.numel() == 0
    assert torch.equal(lin(x), y_ref)
    np.testing.assert_allclose(lin(x).detach().cpu().numpy(), want, rtol=0, atol=4e-6 * np.abs(want).max())


@pytest.mark.parametrize("act", ["plain", "silu"])
def test_hybrid_weight_storage(act):
    """mi355q_weight_storage = "hybrid" (round 6): a layer on the per-block-exponent route keeps its weights at width + 0.5 bits and
    expands them per forward, a layer on the row-scale route keeps its resident int8 operand -- both bit-identical to the resident
    layer, with the fp32 Parameter released too"""
    import torch
    import mi355q.quantize as Q
    torch.manual_seed(4)
    K, N, M = 1024, 512, 300
    fp = torch.nn.Linear(K, N)
    h = torch.randn(M, K) * torch.exp(torch.randn(M, 1))
    x = {"plain": h, "silu": torch.nn.functional.silu(h) * torch.randn(M, K)}[act].to("cuda:0")
    ref_cfg, cfg = _lin_cfg(6), _lin_cfg(6, mi355q_weight_storage="hybrid")
    ref = Q.get_quantized_cls("linear", ref_cfg).from_float(fp, ref_cfg).to("cuda:0")
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    y_ref, y = ref(x), lin(x)
    assert torch.equal(y, y_ref)
    assert lin._uses_bf16_route() == ref._uses_bf16_route() == (act == "silu")
    if act == "silu":
        assert lin._w_packed is not None and not lin._w_packed.row_scale_flavour and lin.weight_storage_bits() <= 6.75
    else:
        assert lin._w_packed is None and lin._packed[0] is not None          # the resident int8 operand
    lin.release_fp32_weight()
    assert lin.weight.numel() == 0 and torch.equal(lin(x), y_ref)


def test_packed_storage_packs_when_the_weights_arrive():
    """mi355q_weight_storage = "packed": quantised and packed when the module reaches the GPU and when a state dict is loaded,
    before any forward (the loader half); outputs equal a layer that packed at its first forward"""
    import torch
    import mi355q.quantize as Q
    torch.manual_seed(6)
    fp, other = torch.nn.Linear(512, 256), torch.nn.Linear(512, 256)
    x = (torch.randn(70, 512) * torch.exp(torch.randn(70, 1))).to("cuda:0")
    cfg, lazy_cfg = _lin_cfg(6, mi355q_weight_storage="packed"), _lin_cfg(6)
    cls = Q.get_quantized_cls("linear", cfg)
    lin = cls.from_float(fp, cfg)
    assert lin._w_packed is None and lin.weight_requires_quantisation          # (on the CPU: nothing to pack with)
    lin = lin.to("cuda:0")
    assert lin._w_packed is not None and not lin.weight_requires_quantisation, "not packed on arrival at the GPU"
    assert lin.weight_storage_bits() <= 7.0            # (6.5 + the fixed-size exception buckets, large against a 512 x 256 layer)
    ref = cls.from_float(fp, lazy_cfg).to("cuda:0")
    assert torch.equal(lin(x), ref(x)) and torch.equal(lin(x), ref(x))
    # a checkpoint loaded into the packed layer: packed again at once, from the new values
    lin.load_state_dict({k: v.to("cuda:0") for k, v in other.state_dict().items()})
    assert lin._w_packed is not None and not lin.weight_requires_quantisation
    ref2 = cls.from_float(other, lazy_cfg).to("cuda:0")
    assert torch.equal(lin(x), ref2(x))
    assert not torch.equal(lin(x), ref(x))


def test_pack_now_and_master_requantize():
    """pack at load (no forward needed) and the search loop's re-quantise-in-place with kept fp32 master weights: another
    width is applied without reloading anything, results equal a freshly built layer's"""
    import torch
    import mi355q.quantize as Q
    torch.manual_seed(4)
    fp = torch.nn.Linear(512, 256)
    x = (torch.randn(100, 512) * torch.exp(torch.randn(100, 1))).to("cuda:0")
    cfg6, cfg4 = _lin_cfg(6, mi355q_keep_master=True), _lin_cfg(4)
    lin = Q.get_quantized_cls("linear", cfg6).from_float(fp, cfg6).to("cuda:0")
    lin.pack_now()
    assert not lin.weight_requires_quantisation and lin._packed is not None
    y6 = lin(x)
    fresh6 = Q.get_quantized_cls("linear", _lin_cfg(6)).from_float(fp, _lin_cfg(6)).to("cuda:0")
    assert torch.equal(y6, fresh6(x))
    lin.requantize(cfg4)                                     # trial 2: W4A4, nothing reloaded
    y4 = lin(x)
    fresh4 = Q.get_quantized_cls("linear", cfg4).from_float(fp, cfg4).to("cuda:0")
    assert torch.equal(y4, fresh4(x)) and torch.equal(lin.weight, fresh4.weight)
    lin.requantize(_lin_cfg(6))                              # and back
    assert torch.equal(lin(x), y6)


@pytest.mark.parametrize("B,T,hd", [(6, 320, 64), (4, 2048, 128), (3, 1000, 64)])
@pytest.mark.parametrize("width", [6, 4])
def test_softmax_folded_into_the_product(B, T, hd, width):
    """softmax_bmm_block_fp(scores, v) == bmm_block_fp(softmax(scores), v) (matmul.py:146-196 on the probabilities):
    causal mask included; against the oracle's float64 evaluation and against the three-step route on the GPU"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = _lin_cfg(width)
    r = np.random.default_rng(B + T)
    s = (r.normal(size=(B, T, T)) * 3).astype(np.float32)
    s += np.triu(np.full((T, T), np.finfo(np.float32).min, np.float32), 1)[None]
    s = np.maximum(s, np.finfo(np.float32).min)
    v = r.normal(size=(B, T, hd)).astype(np.float32)
    st, vt = torch.from_numpy(s).to("cuda:0"), torch.from_numpy(v).to("cuda:0")
    fused = Q.get_quantized_func("softmax_bmm", cfg)(st, vt, config=cfg)
    three = Q.get_quantized_func("bmm", cfg)(torch.softmax(st, -1), vt, config=cfg)
    e = np.exp(s - s.max(-1, keepdims=True))
    ref = O.matmul_quantized((e / e.sum(-1, keepdims=True)).astype(np.float32), v, cfg)
    scale = np.abs(ref).max()
    # (a probability one ulp apart may round to the next mantissa: one 2^-(width-1) step of one of T terms of a row)
    assert np.abs(fused.cpu().numpy() - ref).max() <= 2e-3 * scale
    assert (fused - three).abs().max().item() <= 2e-3 * scale
    assert np.abs(fused.cpu().numpy() - ref).mean() <= 2e-5 * scale


@pytest.mark.parametrize("B,T,hd,mode", [(4, 320, 64, "causal"), (2, 2048, 128, "causal"), (3, 520, 64, "mask"), (2, 640, 64, "both"),
                                         (2, 100, 64, "causal")])
def test_softmax_product_with_mask_and_causal(B, T, hd, mode):
    """the reference's attention between its two products -- `w = w + mask; w = max(w, finfo.min); p = softmax(w);
    bmm_1(p, v)` (modeling_opt.py:262-312) -- as one call: additive mask tensor, the causal rule without a mask tensor,
    both; rows shorter than the fused kernel takes fall back to the same steps"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = _lin_cfg(6)
    r = np.random.default_rng(T)
    fmin = np.finfo(np.float32).min
    s = (r.normal(size=(B, T, T)) * 3).astype(np.float32)
    v = r.normal(size=(B, T, hd)).astype(np.float32)
    causal_mask = np.triu(np.full((T, T), fmin, np.float32), 1)
    add_mask = (r.normal(size=(T, T)) * 0.5).astype(np.float32)
    add_mask[:, ::7] = fmin                                    # (a padding-style mask: whole key columns off)
    add_mask[:, 0] = 0
    m = {"causal": causal_mask, "mask": add_mask, "both": np.maximum(add_mask + causal_mask, fmin)}[mode]
    w = np.maximum(s + m[None], fmin)
    e = np.exp(w - w.max(-1, keepdims=True))
    ref = O.matmul_quantized((e / e.sum(-1, keepdims=True)).astype(np.float32), v, cfg)
    st, vt = torch.from_numpy(s).to("cuda:0"), torch.from_numpy(v).to("cuda:0")
    kw = {"causal": dict(causal=True), "mask": dict(mask=torch.from_numpy(add_mask).to("cuda:0")),
          "both": dict(mask=torch.from_numpy(add_mask).to("cuda:0")[None], causal=True)}[mode]
    out = Q.get_quantized_func("softmax_bmm", cfg)(st, vt, cfg, **kw).cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(out - ref).max() <= 2e-3 * scale and np.abs(out - ref).mean() <= 2e-5 * scale


def test_shared_activation_is_quantised_once():
    """q / k / v style: layers called on the same activation reuse the quantised operand (one quantiser launch); an in-place
    write or another tensor at the same address quantises again; results equal layers that never share"""
    import torch
    import mi355q.quantize as Q
    from mi355q import _lib, ops
    cfg = _lin_cfg(6, mi355q_align="rows")
    dev = "cuda:0"
    torch.manual_seed(4)
    layers = [Q.get_quantized_cls("linear", cfg)(256, 128, bias=True, config=dict(cfg)).to(dev) for _ in range(3)]
    x = torch.randn(2, 96, 256, device=dev)
    lib = _lib.load_library()
    real = lib.mi355q_block_fp_quantize_aligned_rows_seg
    calls = []

    def counting(*a):
        calls.append(1)
        return real(*a)

    with torch.no_grad():
        for l in layers:                                     # first forwards: weight packing + route decision
            l(x.clone())
        ops.REUSE_QUANTISED_INPUT = False
        ref = [l(x).clone() for l in layers]
        ops.REUSE_QUANTISED_INPUT = True
        layers[0](x.clone())                                 # (another tensor of the shape: the record no longer names x)
        lib.mi355q_block_fp_quantize_aligned_rows_seg = counting
        try:
            out = [l(x).clone() for l in layers]
            assert len(calls) == 1, len(calls)
            assert all(torch.equal(a, b) for a, b in zip(out, ref))
            x.mul_(1.5)                                      # in-place write: version counter moves
            out2 = layers[1](x).clone()
            assert len(calls) == 2
            ops.REUSE_QUANTISED_INPUT = False
            assert torch.equal(out2, layers[1](x))
            ops.REUSE_QUANTISED_INPUT = True
            n0 = len(calls)
            y = layers[0](x)                                 # (a hit: the record names x since the call above)
            del x
            z = torch.randn(2, 96, 256, device=dev)          # may land on the freed address: the record keeps x alive, so it cannot
            got = layers[2](z).clone()
            assert len(calls) == n0 + 1
            ops.REUSE_QUANTISED_INPUT = False
            assert torch.equal(got, layers[2](z))
        finally:
            lib.mi355q_block_fp_quantize_aligned_rows_seg = real
            ops.REUSE_QUANTISED_INPUT = True


@pytest.mark.parametrize("B,H,T,hd,strided", [(1, 12, 2048, 64, True), (2, 3, 40, 128, False), (1, 4, 100, 32, True)])
def test_rope_one_launch_bit_exact(B, H, T, hd, strided):
    """apply_rotary_pos_emb_block_fp through the one-launch kernel (mi355q_rope_apply) == the reference's op sequence
    `(q * cos) + (rotate_half(q) * sin)` evaluated op by op in fp32 (rotary_positional_encoding.py:59-82), bit for bit;
    q / k as the transposed views the Llama attention hands over, shuffled position ids"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = _lin_cfg(6)
    f = Q.get_quantized_func("rotary_positional_encoding", cfg)
    g = torch.Generator().manual_seed(T + hd)
    pos = torch.arange(T)[:, None] * (1.0 / 10000 ** (torch.arange(0, hd, 2) / hd))[None, :]
    emb = torch.cat((pos, pos), -1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]
    if strided:
        q = torch.randn(B, T, H, hd, generator=g).transpose(1, 2)
        k = torch.randn(B, T, H, hd, generator=g).transpose(1, 2)
    else:
        q, k = torch.randn(B, H, T, hd, generator=g), torch.randn(B, H, T, hd, generator=g)
    ids = torch.stack([torch.randperm(T, generator=g) for _ in range(B)])
    dev = "cuda:0"
    qd, kd = q.to(dev), k.to(dev)
    if strided:
        qd, kd = qd.transpose(1, 2).contiguous().transpose(1, 2), kd.transpose(1, 2).contiguous().transpose(1, 2)
        assert not qd.is_contiguous()
    qe, ke = f(qd, kd, cos.to(dev), sin.to(dev), ids.to(dev), cfg)
    cq = O.block_fp_quantize(cos.numpy()[0, 0], 6, 8, 127, [1, 16], False)
    sq = O.block_fp_quantize(sin.numpy()[0, 0], 6, 8, 127, [1, 16], False)
    for got, x in ((qe, q), (ke, k)):
        xn = x.numpy()
        c, s = cq[ids.numpy()][:, None], sq[ids.numpy()][:, None]
        rot = np.concatenate((-xn[..., hd // 2:], xn[..., : hd // 2]), -1)
        ref = (xn * c).astype(np.float32) + (rot * s).astype(np.float32)
        assert got.shape == x.shape and got.is_contiguous()
        assert np.array_equal(got.cpu().numpy(), ref.astype(np.float32))


def test_out_buffers_move_their_version_counter():
    """a GEMM that writes into a caller's `out` tensor through its raw pointer bumps the tensor's version like an in-place
    torch op: the shared-activation reuse (keyed by the version) never serves a stale operand for a re-filled buffer"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    dev = "cuda:0"
    cfg = _lin_cfg(6, mi355q_align="rows")
    torch.manual_seed(8)
    a, b = (torch.randn(64, 256, device=dev) for _ in range(2))
    w = torch.randn(256, 256, device=dev) * 0.05
    _, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
    wa = ops.bfp_align_rows(wm, we, 5, 127)
    lin = Q.get_quantized_cls("linear", cfg)(256, 128, bias=False, config=dict(cfg)).to(dev)
    buf = torch.empty(64, 256, device=dev)
    with torch.no_grad():
        outs = []
        for src in (a, b):
            v0 = buf._version
            ops.bfp_gemm_aligned(ops.block_fp_quantize_aligned_rows(src, 6, 8, 127), wa, None, out=buf)
            assert buf._version > v0
            outs.append(lin(buf).clone())
        ops.REUSE_QUANTISED_INPUT = False
        try:
            ops.bfp_gemm_aligned(ops.block_fp_quantize_aligned_rows(b, 6, 8, 127), wa, None, out=buf)
            assert torch.equal(outs[1], lin(buf))
        finally:
            ops.REUSE_QUANTISED_INPUT = True
    assert not torch.equal(outs[0], outs[1])


def test_linear_and_matmul_in_the_unblocked_arithmetics():
    """LinearMinifloatIEEE / LinearMinifloatDenorm and the matmul functions of the un-blocked arithmetics run on the HIP
    quantisers (reference linear.py:206-259, matmul.py:36-143); LinearLog fails at its first forward like the reference's
    (it hands `exponent_width` to log_quantizer: SURVEY 8a quirk 3)"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    dev = "cuda:0"
    torch.manual_seed(2)
    x = torch.randn(3, 40, 96, device=dev)
    for name, f in (("minifloat_ieee", O.minifloat_ieee_quantize), ("minifloat_denorm", O.minifloat_denorm_quantize)):
        cfg = dict(name=name, is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias=None,
                   weight_width=8, weight_exponent_width=4, weight_exponent_bias=9, bias_width=8, bias_exponent_width=4,
                   bias_exponent_bias=None)
        lin = Q.get_quantized_cls("linear", cfg)(96, 64, bias=True, config=cfg).to(dev)
        w0, b0 = lin.weight.detach().cpu().numpy().copy(), lin.bias.detach().cpu().numpy().copy()
        y = lin(x).detach().cpu().numpy()
        xq = f(x.cpu().numpy(), 8, 4, None)
        wq, bq = f(w0, 8, 4, 9), f(b0, 8, 4, None)
        assert np.array_equal(lin.weight.detach().cpu().numpy(), wq)
        np.testing.assert_allclose(y, xq @ wq.T + bq, rtol=1e-4, atol=1e-5)
        mm = Q.get_quantized_func("matmul", cfg)(x, x.transpose(1, 2).contiguous(), cfg).cpu().numpy()
        np.testing.assert_allclose(mm, xq @ np.swapaxes(f(np.swapaxes(x.cpu().numpy(), 1, 2).copy(), 8, 4, 9), 0, 0), rtol=1e-4, atol=1e-4)
    cfg = dict(name="log", is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias=None,
               weight_width=8, weight_exponent_width=4, weight_exponent_bias=None, bias_width=8, bias_exponent_width=4,
               bias_exponent_bias=None)
    with pytest.raises(TypeError):
        Q.get_quantized_cls("linear", cfg)(96, 64, bias=True, config=cfg).to(dev)(x)


@pytest.mark.parametrize("M,K,N,n", [(300, 512, 256, 3), (2048, 2048, 2048, 3), (1000, 1024, 2816, 2),
                                     (256, 8192, 256, 3), (130, 4096, 512, 2),        # (split-K slices)
                                     (2048, 4096, 4096, 3)])   # (384 tiles: one and a half rounds over the compute units)
def test_grouped_linear_equals_separate_calls(M, K, N, n):
    """grouped_linear(x, [q, k, v]) -- one quantisation, ONE launch of the tile GEMM over all column tiles -- == the layers
    called one by one, bit for bit (exceptions of x and of every w included); groups that do not qualify fall back"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    dev = "cuda:0"
    cfg = _lin_cfg(6, mi355q_align="rows")
    torch.manual_seed(M + N)
    layers = [Q.get_quantized_cls("linear", cfg)(K, N, bias=True, config=dict(cfg)).to(dev) for _ in range(n)]
    with torch.no_grad():
        for i, l in enumerate(layers):
            l.weight[i::7, 48:64] *= 2.0 ** 6                 # exception blocks in every weight, different rows
    if K <= 2048:
        x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
    else:
        # (rows this long spread their block maxima over more exponents than a row window holds: the buckets would
        #  overflow and the launch take its blockwise product, whose atomic add-back is not reproducible to the bit.
        #  Magnitudes within a factor of two keep the exception blocks to the ones placed below.)
        x = (torch.rand(M, K, device=dev) * 0.9 + 0.6) * torch.sign(torch.randn(M, K, device=dev)) * torch.exp(torch.randn(M, 1, device=dev))
    x[::9, 32:48] *= 2.0 ** -9
    with torch.no_grad():
        first = Q.grouped_linear(x, layers)                   # first PTQ forward: not packed yet -> separate calls
        ref = [l(x).clone() for l in layers]
        calls, real = [], ops.bfp_gemm_aligned_multi
        ops.bfp_gemm_aligned_multi = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            got = Q.grouped_linear(x, layers)
        finally:
            ops.bfp_gemm_aligned_multi = real
    assert len(calls) == 1
    for a, b, c in zip(first, ref, got):
        assert torch.equal(a, b) and torch.equal(b, c)
    # a group with different widths does not qualify: same results through the separate calls
    other = Q.get_quantized_cls("linear", _lin_cfg(4))(K, N, bias=True, config=_lin_cfg(4, mi355q_align="rows")).to(dev)
    with torch.no_grad():
        o1 = other(x).clone()
        mixed = Q.grouped_linear(x, [layers[0], other])
    assert torch.equal(mixed[0], ref[0]) and torch.equal(mixed[1], o1)



@pytest.mark.parametrize("M,K,N,n,with_norm", [(300, 512, 256, 3, False), (2048, 2048, 2048, 3, True), (1000, 1024, 2816, 2, True)])
def test_grouped_linear_with_width_bit_weight_storage(M, K, N, n, with_norm):
    """mi355q_weight_storage = "packed" layers take the grouped launch too (round 5): every member expands into its own scratch
    slot, then ONE launch -- the same bits as resident layers called one by one, the RMSNorm inside the quantiser included"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    dev = "cuda:0"
    torch.manual_seed(M + N)
    fps = [torch.nn.Linear(K, N) for _ in range(n)]
    with torch.no_grad():
        for i, f in enumerate(fps):
            f.weight[i::7, 48:64] *= 2.0 ** 6                 # exception blocks in every weight, different rows
    res_cfg, pk_cfg = _lin_cfg(6, mi355q_align="rows"), _lin_cfg(6, mi355q_align="rows", mi355q_weight_storage="packed")
    resident = [Q.get_quantized_cls("linear", res_cfg).from_float(f, dict(res_cfg)).to(dev) for f in fps]
    packed = [Q.get_quantized_cls("linear", pk_cfg).from_float(f, dict(pk_cfg)).to(dev) for f in fps]
    x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
    x[::9, 32:48] *= 2.0 ** -9
    norm = (torch.rand(K, device=dev) + 0.5, 1e-6) if with_norm else None
    with torch.no_grad():
        Q.grouped_linear(x, resident, norm=norm)              # first forwards: pack
        Q.grouped_linear(x, packed, norm=norm)
        ref = [y.clone() for y in Q.grouped_linear(x, resident, norm=norm)]
        calls, real = [], ops.bfp_gemm_aligned_multi
        ops.bfp_gemm_aligned_multi = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            got = Q.grouped_linear(x, packed, norm=norm)
        finally:
            ops.bfp_gemm_aligned_multi = real
    assert all(l._w_packed is not None and l.weight_storage_bits() < 7.0 for l in packed)
    assert len(calls) == 1
    for a, b in zip(ref, got):
        if with_norm:
            # (per-channel norm weights on top of the 2^-9 blocks fill some tiles' lists past what the tile keeps as vectors: those
            #  entries are added with atomics, whose order is not fixed -- the resident layers differ from THEMSELVES run to run in
            #  a handful of last bits there, mi355q_gemm_v10.hip / v9 "mode 3")
            torch.testing.assert_close(a, b, rtol=0, atol=4e-7 * float(a.abs().max()))
        else:
            assert torch.equal(a, b)

@pytest.mark.parametrize("which", ["x", "w"])
def test_grouped_linear_bucket_overflow_takes_the_blockwise_product(which):
    """an exception bucket that overflows (x rows or one weight's rows with more out-of-window blocks than a bucket
    holds) sends the WHOLE grouped launch to the blockwise-exact product its workgroups carry, each weight's share
    computed by that weight's workgroups: still == the layers called one by one"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    dev = "cuda:0"
    M, K, N, n = 512, 2048, 512, 3
    cfg = _lin_cfg(6, mi355q_align="rows")
    torch.manual_seed(11)
    layers = [Q.get_quantized_cls("linear", cfg)(K, N, bias=True, config=dict(cfg)).to(dev) for _ in range(n)]
    x = torch.randn(M, K, device=dev)
    with torch.no_grad():
        if which == "w":
            layers[1].weight[:64].view(64, K // 16, 16)[:, ::2] *= 2.0 ** -8      # every other block of 64 rows: > 120 per bucket
        else:
            x[:64].view(64, K // 16, 16)[:, ::2] *= 2.0 ** -8
        first = Q.grouped_linear(x, layers)
        ref = [l(x).clone() for l in layers]
        calls, real = [], ops.bfp_gemm_aligned_multi
        ops.bfp_gemm_aligned_multi = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            got = Q.grouped_linear(x, layers)
        finally:
            ops.bfp_gemm_aligned_multi = real
    assert len(calls) == 1
    for a, b, c in zip(first, ref, got):                      # (the blockwise product adds its exception blocks with fp32
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)    #  atomics: the order of a row's additions is not fixed)
        torch.testing.assert_close(b, c, rtol=1e-5, atol=1e-5)
    from oracle import np_oracle as O
    xq = O.block_fp_quantize(x.cpu().numpy(), 6, 8, 127, [1, 16], skip_first_dim=False)
    for l, y in zip(layers, got):
        want = xq.astype(np.float64) @ l.weight.detach().cpu().numpy().astype(np.float64).T + l.bias.detach().cpu().numpy()
        np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("op", ["relu", "silu_mul"])
@pytest.mark.parametrize("route", ["bf16", "rows"])
@pytest.mark.parametrize("M,K", [(777, 2048), (130, 11008)])
def test_forward_after_equals_torch_ops_then_forward(op, route, M, K):
    """Linear.forward_after(x, op[, other]) -- relu / silu(x) * other read by the layer's x quantiser itself -- == the
    layer called on the torch result (fc2(relu(.)), modeling_opt.py:412-420; down_proj(act(gate) * up),
    modeling_llama.py:216), bit for bit: the fused arithmetic rounds operation by operation like the separate kernels"""
    import torch
    import torch.nn.functional as F
    import mi355q.quantize as Q
    from mi355q import ops
    dev = "cuda:0"
    N = 512
    cfg = _lin_cfg(6, mi355q_align="rows")
    torch.manual_seed(5)
    lin = Q.get_quantized_cls("linear", cfg)(K, N, bias=True, config=cfg).to(dev)
    x = torch.randn(M, K, device=dev) * 3
    if K > 2048 and route == "rows":     # (long rows: magnitudes within a factor of two, so that no bucket overflows)
        x = (torch.rand(M, K, device=dev) * 0.9 + 0.6) * torch.sign(x) * 3
    other = torch.randn(M, K, device=dev) if op == "silu_mul" else None
    x[3, :40] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1e-9, -1e-9, 12.0, -12.0, 20.0, -20.0] * 4, device=dev)[:40]
    x[5, :16] = 0
    with torch.no_grad():
        plain = F.relu(x) if op == "relu" else F.silu(x) * other
        first = lin.forward_after(x, op, other)                # first PTQ forward: packs the weight, torch ops in front
        assert torch.equal(first, lin(plain))
        lin._x_cap = ops.ROW_NO_ALIGN if route == "bf16" else ops.ROW_BUCKET_CAP_MAX
        assert lin._uses_bf16_route() == (route == "bf16")
        want = lin(plain).clone()
        calls = []
        real_t, real_r = ops.block_fp_quantize_bf16_tiled, ops.block_fp_quantize_aligned_rows
        ops.block_fp_quantize_bf16_tiled = lambda *a, **k: (calls.append(k.get("pre")), real_t(*a, **k))[1]
        ops.block_fp_quantize_aligned_rows = lambda *a, **k: (calls.append(k.get("pre")), real_r(*a, **k))[1]
        try:
            got = lin.forward_after(x, op, other)
            got3 = lin.forward_after(x.view(1, M, K), op, None if other is None else other.view(1, M, K))
        finally:
            ops.block_fp_quantize_bf16_tiled, ops.block_fp_quantize_aligned_rows = real_t, real_r
    assert len(calls) == 2 and all(c is not None and c[0] == op for c in calls)
    assert torch.equal(got, want) and torch.equal(got3.view(M, N), want)
    # what does not qualify runs the torch ops in front of forward(): same values
    byp = Q.get_quantized_cls("linear", dict(cfg, bypass=True))(K, N, bias=True, config=dict(cfg, bypass=True)).to(dev)
    with torch.no_grad():
        torch.testing.assert_close(byp.forward_after(x, op, other), byp(plain), rtol=1e-5, atol=1e-4)     # (library fp32 GEMM)
    with pytest.raises(ValueError):
        lin.forward_after(x, "gelu")


def test_silu_matches_torch_bit_for_bit_over_the_float_range():
    """the fused quantiser's silu(x) * u against torch's two kernels on 2^24 values spread over every exponent: the bf16
    operand (exact image of the fake-quantised values) is identical"""
    import torch
    import torch.nn.functional as F
    from mi355q import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(1)
    bits = torch.randint(0, 2 ** 31 - 1, (4096, 4096), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    x = bits.view(torch.float32).clone()
    x = torch.where(torch.isfinite(x) & (x.abs() < 1e30), x, torch.randn_like(x))
    x = torch.where(torch.rand(x.shape, device=dev, generator=g) < 0.5, x, torch.randn(x.shape, device=dev, generator=g) * 4)
    u = torch.randn(x.shape, device=dev, generator=g)
    want = F.silu(x) * u
    want = torch.where(torch.isfinite(want), want, torch.zeros_like(want))
    x = torch.where(torch.isfinite(F.silu(x) * u), x, torch.zeros_like(x))
    a = ops.block_fp_quantize_bf16_tiled(want.contiguous(), 8, 8, None, reuse=False)
    b = ops.block_fp_quantize_bf16_tiled(x.contiguous(), 8, 8, None, reuse=False, pre=("silu_mul", u))
    assert torch.equal(a, b)
    r1 = ops.block_fp_quantize_bf16_tiled(F.relu(x).contiguous(), 8, 8, None, reuse=False)
    r2 = ops.block_fp_quantize_bf16_tiled(x.contiguous(), 8, 8, None, reuse=False, pre=("relu", None))
    assert torch.equal(r1, r2)


def test_rope_tables_are_quantised_once():
    """the cos / sin tables are model constants: their quantised images are kept between calls (same storage, version and
    quantiser parameters), re-made after an in-place change or under another config; outputs identical either way"""
    import torch
    import mi355q.quantize as Q
    from mi355q.quantize import quantized_functions as QF
    cfg = _lin_cfg(6)
    f = Q.get_quantized_func("rotary_positional_encoding", cfg)
    dev = "cuda:0"
    T, hd = 96, 64
    pos = torch.arange(T)[:, None] * (1.0 / 10000 ** (torch.arange(0, hd, 2) / hd))[None, :]
    emb = torch.cat((pos, pos), -1)
    cos, sin = emb.cos()[None, None].to(dev), emb.sin()[None, None].to(dev)
    q, k = torch.randn(2, 4, T, hd, device=dev), torch.randn(2, 4, T, hd, device=dev)
    ids = torch.arange(T, device=dev)[None].expand(2, T).contiguous()
    calls, real = [], QF.QUANTIZER_MAP["block_fp"]
    QF._ROPE_TABLES.clear()
    QF.QUANTIZER_MAP["block_fp"] = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    try:
        a = f(q, k, cos, sin, ids, cfg)
        assert len(calls) == 2
        b = f(q, k, cos[:, :, :T], sin[:, :, :T], ids, cfg)          # (new view objects of the same storage: the model's slices)
        assert len(calls) == 2
        cfg4 = _lin_cfg(4)
        f(q, k, cos, sin, ids, cfg4)
        assert len(calls) == 4
        cos.mul_(0.5)
        c = f(q, k, cos, sin, ids, cfg)
        assert len(calls) == 5                                        # (cos changed in place, sin did not)
    finally:
        QF.QUANTIZER_MAP["block_fp"] = real
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and not torch.equal(a[0], c[0])
    QF._ROPE_TABLES.clear()
    d = f(q, k, cos, sin, ids, cfg)
    assert torch.equal(c[0], d[0]) and torch.equal(c[1], d[1])


@pytest.mark.parametrize("kind", ["rms", "layer"])
@pytest.mark.parametrize("M,K,N,n", [(300, 512, 256, 3), (2048, 4096, 512, 2), (77, 1024, 256, 1), (64, 11008 // 2 + 384, 256, 2),
                                     (40, 16384, 256, 1)])
def test_grouped_linear_with_the_norm_inside_the_quantiser(M, K, N, n, kind):
    """grouped_linear(x, layers, norm=(weight, eps)) -- LlamaRMSNorm applied by the row quantiser itself -- against the layers
    on the separately normalised tensor: the quantised operand agrees except where the last bit of the mean moved an
    element across a rounding boundary (a handful in a million), outputs to that accuracy; identical run after run; equal
    to the oracle's norm + quantiser to the same accuracy"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    from oracle import np_oracle as O
    dev = "cuda:0"
    cfg = _lin_cfg(6, mi355q_align="rows")
    torch.manual_seed(M + K)
    layers = [Q.get_quantized_cls("linear", cfg)(K, N, bias=False, config=dict(cfg)).to(dev) for _ in range(n)]
    x = torch.randn(M, K, device=dev)
    if K > 2048:       # (long rows of normal values overflow the exception buckets: see test_grouped_linear_equals_separate_calls)
        x = (torch.rand(M, K, device=dev) * 0.9 + 0.6) * torch.sign(x)
    x = x * torch.exp(torch.randn(M, 1, device=dev))
    x[::9, 32:48] *= 2.0 ** -9
    w = (1 + 0.1 * torch.randn(K, device=dev)).contiguous()
    eps = 1e-6
    if kind == "layer":
        x = x + 0.3 * x.abs().mean(-1, keepdim=True)              # (a non-zero row mean)
        b = (0.05 * torch.randn(K, device=dev)).contiguous()
        norm, tag = (w, b, 1e-5), "layernorm"
    else:
        norm, tag = (w, eps), "rmsnorm"
    with torch.no_grad():
        if kind == "layer":
            h = torch.nn.functional.layer_norm(x, (K,), w, b, 1e-5)
        else:
            h = w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))
        first = Q.grouped_linear(x, layers, norm=norm)            # first PTQ forward: torch norm + separate calls
        ref = [l(h).clone() for l in layers]
        for a, b in zip(first, ref):
            assert torch.equal(a, b)
        calls, real = [], ops.block_fp_quantize_aligned_rows
        ops.block_fp_quantize_aligned_rows = lambda *a, **k: (calls.append(k.get("pre")), real(*a, **k))[1]
        try:
            got = Q.grouped_linear(x, layers, norm=norm)
            again = Q.grouped_linear(x, layers, norm=norm)
        finally:
            ops.block_fp_quantize_aligned_rows = real
        assert len(calls) == 2 and all(c is not None and c[0] == tag for c in calls)
    for a, b, c in zip(ref, got, again):
        assert torch.equal(b, c)                                  # reproducible
        scale = a.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-3 * scale         # a few elements moved by one quantisation step
        assert ((a - b).abs() > 1e-6 * scale).float().mean().item() < 0.2
    hq = O.block_fp_quantize(h.cpu().numpy(), 6, 8, 127, [1, 16], skip_first_dim=False).astype(np.float64)
    want = hq @ layers[0].weight.detach().cpu().numpy().astype(np.float64).T
    np.testing.assert_allclose(got[0].cpu().numpy(), want, rtol=0, atol=2e-3 * float(np.abs(want).max()))


@pytest.mark.parametrize("arith", ["block_minifloat", "block_log", "minifloat_ieee", "minifloat_denorm", "integer"])
def test_minifloat_and_log_linear_take_the_bf16_tile_gemm(arith):
    """PTQ LinearBlockMinifloat / LinearBlockLog (linear.py:145-203): their fake-quantised values are exact in bf16, so
    F.linear(x_q, W_q, b_q) runs as the bf16 flavour of the tile GEMM; == the library fp32 GEMM on the same quantised
    values to fp32 summation-order accuracy, == the oracle's quantisers + float64 contraction; 3-D input, QAT and grad
    keep F.linear"""
    import torch
    import torch.nn.functional as F
    import mi355q.quantize as Q
    from mi355q import ops
    from oracle import np_oracle as O
    dev = "cuda:0"
    if arith == "block_minifloat":
        cfg = dict(name=arith, is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias_width=8,
                   data_in_block_size=[1, 16], weight_width=8, weight_exponent_width=4, weight_exponent_bias_width=8,
                   weight_block_size=[1, 16], bias_width=8, bias_exponent_width=4, bias_exponent_bias_width=8, bias_block_size=[16])
        scale = 40.0        # (quirk 5: blocks with max < 2 quantise to zeros; give the layer something to do)
    elif arith in ("minifloat_ieee", "minifloat_denorm"):
        cfg = dict(name=arith, is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias=None,
                   weight_width=8, weight_exponent_width=4, weight_exponent_bias=None, bias_width=8, bias_exponent_width=4,
                   bias_exponent_bias=None)
        scale = 1.0
    elif arith == "integer":
        cfg = dict(name=arith, is_ptq=True, bypass=False, data_in_width=8, data_in_frac_width=4, weight_width=8, weight_frac_width=6,
                   bias_width=8, bias_frac_width=6)
        scale = 1.0
    else:
        cfg = dict(name=arith, is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_bias_width=8, data_in_block_size=[1, 16],
                   weight_width=8, weight_exponent_bias_width=8, weight_block_size=[1, 16], bias_width=8,
                   bias_exponent_bias_width=8, bias_block_size=[16])
        scale = 1.0
    torch.manual_seed(3)
    M, K, N = 300, 512, 272
    lin = Q.get_quantized_cls("linear", cfg)(K, N, bias=True, config=cfg).to(dev)
    with torch.no_grad():
        lin.weight.mul_(scale * 20)
        lin.bias.mul_(scale * 20)
    w0, b0 = lin.weight.detach().cpu().numpy().copy(), lin.bias.detach().cpu().numpy().copy()
    x = torch.randn(2, M // 2, K, device=dev) * scale
    calls, real = [], ops.bf16_gemm_tiled
    ops.bf16_gemm_tiled = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            y = lin(x)
            y2 = lin(x)
            assert len(calls) == 2 and torch.equal(y, y2)
            xq = lin.x_quantizer(x)
            want = F.linear(xq, lin.weight, lin.bias)
            slow = Q.get_quantized_cls("linear", cfg)(K, N, bias=True, config=dict(cfg, mi355q_values_gemm="fp32")).to(dev)
            slow.load_state_dict({"weight": torch.from_numpy(w0), "bias": torch.from_numpy(b0)})
            n0 = len(calls)
            ys = slow(x)
            assert len(calls) == n0
        yg = lin(x.clone().requires_grad_(True))                     # d/dx wanted: the differentiable library product
        assert len(calls) == n0 and yg.requires_grad
        lin(x)                                                       # (grad mode on, nothing to differentiate: tile GEMM)
        assert len(calls) == n0 + 1
    finally:
        ops.bf16_gemm_tiled = real
    mx = want.abs().max().item()
    assert (y - want).abs().max().item() <= 2e-6 * mx and (ys - want).abs().max().item() <= 2e-6 * mx
    assert torch.equal(lin.weight, slow.weight)
    if arith in ("minifloat_ieee", "minifloat_denorm", "integer"):
        return                                                # (their quantisers have their own bit-exact tests)
    if arith == "block_minifloat":
        qx = O.block_minifloat_quantize(x.cpu().numpy(), 8, 4, 8, [1, 16], True)
        qw = O.block_minifloat_quantize(w0, 8, 4, 8, [1, 16], False)
        qb = O.block_minifloat_quantize(b0, 8, 4, 8, [16], False)
    else:
        qx = O.block_log_quantize(x.cpu().numpy(), 8, 8, [1, 16], True)
        qw = O.block_log_quantize(w0, 8, 8, [1, 16], False)
        qb = O.block_log_quantize(b0, 8, 8, [16], False)
    ref = qx.astype(np.float64) @ qw.astype(np.float64).T + qb
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=3e-6 * float(np.abs(ref).max()))


@pytest.mark.parametrize("T,block", [(12, [1, 16]), (24, [1, 32]), (20, [1, 32]), (28, [1, 32]), (48, [1, 16])])
def test_block_minifloat_attention_products_at_token_counts_that_refit_the_block(T, block):
    """T = 12 under block [1,16] re-fits the block to 12 values (b1 / 4 = 3), which the bf16-output quantiser's vector
    path does not take -- the product must fall back to the fp32 quantiser instead of raising (quantized_functions/matmul.py:199-249)"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = dict(name="block_minifloat", bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias_width=8,
               data_in_block_size=block, weight_width=8, weight_exponent_width=4, weight_exponent_bias_width=8, weight_block_size=block)
    g = torch.Generator().manual_seed(T)
    probs = torch.softmax(torch.randn(2, 3, T, T, generator=g) * 2, dim=-1)
    v = torch.randn(2, 3, T, 64, generator=g)
    q = torch.randn(2, 3, T, 64, generator=g)
    kt = torch.randn(2, 3, 64, T, generator=g)
    f = Q.get_quantized_func("matmul", cfg)
    for x, y in ((probs, v), (q, kt)):
        got = f(x.to("cuda:0"), y.to("cuda:0"), dict(cfg))
        ref = O.matmul_quantized(x.numpy(), y.numpy(), cfg)
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max()


def test_forward_after_settles_a_layer_that_packed_on_arrival():
    """a packed-storage layer packs when it reaches the GPU and keeps both flavours until its first forward; when that
    first forward is the fused forward_after (fc2 behind relu), the post-op activations must still settle the route and the
    int8 copies must be dropped (width + 0.5 bits at rest)"""
    import torch
    import mi355q.quantize as Q
    torch.manual_seed(9)
    fp = torch.nn.Linear(512, 256)
    x = (torch.randn(64, 512) * torch.exp(torch.randn(64, 1))).to("cuda:0")
    cfg = _lin_cfg(6, mi355q_weight_storage="packed")
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    assert lin._pending_flavour is not None
    y = lin.forward_after(x, "relu")
    assert lin._pending_flavour is None, "the first forward_after left both flavours resident"
    ref = Q.get_quantized_cls("linear", _lin_cfg(6)).from_float(fp, _lin_cfg(6)).to("cuda:0")
    assert torch.equal(y, ref(torch.relu(x)))
    assert torch.equal(lin.forward_after(x, "relu"), y)          # (and the fused route afterwards gives the same bits)


def test_non_fp32_masks_take_the_generic_route():
    """only fp32 additive masks are handed to the fused kernels; a bool mask must not become an additive 0 / 1 mask"""
    import torch
    from mi355q.quantize import quantized_functions as QF
    m = torch.zeros(1, 1, 8, 8, dtype=torch.bool, device="cuda:0")
    assert QF._mask_2d(m, 8, 8) is None
    assert QF._mask_2d(m.to(torch.float16), 8, 8) is None
    assert QF._mask_2d(m.to(torch.float32), 8, 8) is not None


def test_linear_past_the_row_format_contraction_length():
    """in_features > 16384 (Llama-30B/65B down_proj: 17920, 22016) is past the row-aligned format: the layer keeps every
    block's exponent and runs on the bf16 tile GEMM (round 5; before, the 256-value-group flavour took these); and the removed
    flavour's knob value is refused by name"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    K, N = 17920, 64
    cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8,
               data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8,
               weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8,
               bias_exponent_bias=127, bias_block_size=[16])
    torch.manual_seed(3)
    fp = torch.nn.Linear(K, N, bias=True)
    w0, b0 = fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    x = torch.randn(40, K) * torch.exp(torch.randn(40, 1))
    for _ in range(2):
        y = lin(x.to("cuda:0"))
    assert lin._uses_bf16_route()
    ref = O.bfp_linear_int(x.numpy(), w0, b0, cfg)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=4e-6 * np.abs(ref).max())
    bad = dict(cfg, mi355q_align="groups")
    lin = Q.get_quantized_cls("linear", bad).from_float(torch.nn.Linear(256, 64), bad).to("cuda:0")
    with pytest.raises(ValueError, match="groups"):
        lin(torch.randn(4, 256, device="cuda:0"))


@pytest.mark.parametrize("with_norm", [True, False])
@pytest.mark.parametrize("M,K,Ns", [(300, 512, (256, 256, 256)), (2048, 4096, (512, 768)), (77, 1024, (256,))])
def test_grouped_linear_on_the_per_block_route(M, K, Ns, with_norm):
    """layers on the per-block-exponent route (bf16 tile GEMM) that share their input: ONE activation operand for the group, LlamaRMSNorm
    applied by its quantiser (mi355q_block_fp_quantize_bf16_tiled_norm) -- against the layers called one by one on the separately
    normalised tensor (identical without the norm; with it, up to the elements the last bit of the mean moves across a rounding
    boundary) and against the oracle's norm + quantiser + float64 product"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    from oracle import np_oracle as O
    if not with_norm and len(Ns) == 1:
        pytest.skip("a single layer without a norm is the plain call")
    dev = "cuda:0"
    cfg = _lin_cfg(6, mi355q_align="blocks")
    torch.manual_seed(M + K)
    layers = [Q.get_quantized_cls("linear", cfg)(K, n, bias=True, config=dict(cfg)).to(dev) for n in Ns]
    x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev)) * torch.exp(0.7 * torch.randn(1, K, device=dev))
    w, eps = (1 + 0.1 * torch.randn(K, device=dev)).contiguous(), 1e-6
    norm = (w, eps) if with_norm else None
    with torch.no_grad():
        h = w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)) if with_norm else x
        first = Q.grouped_linear(x, layers, norm=norm)            # first PTQ forward: separate calls
        ref = [l(h).clone() for l in layers]
        assert all(l._uses_bf16_route() for l in layers)
        calls, real = [], ops.block_fp_quantize_bf16_tiled
        ops.block_fp_quantize_bf16_tiled = lambda *a, **k: (calls.append(k.get("pre")), real(*a, **k))[1]
        try:
            got = Q.grouped_linear(x, layers, norm=norm)
            again = Q.grouped_linear(x, layers, norm=norm)
        finally:
            ops.block_fp_quantize_bf16_tiled = real
        assert len(calls) == 2 and all((c is not None and c[0] == "rmsnorm") == with_norm for c in calls)
    for a, b, c in zip(ref, got, again):
        assert torch.equal(b, c)
        if not with_norm:
            assert torch.equal(a, b)
        else:
            scale = a.abs().max().item()
            assert (a - b).abs().max().item() <= 2e-3 * scale
            assert ((a - b).abs() > 1e-6 * scale).float().mean().item() < 0.2
    hq = O.block_fp_quantize(h.cpu().numpy(), 6, 8, 127, [1, 16], skip_first_dim=False).astype(np.float64)
    wq = layers[0].weight.detach().cpu().numpy().astype(np.float64)
    want = hq @ wq.T + layers[0].bias.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got[0].cpu().numpy(), want, rtol=0, atol=2e-3 * float(np.abs(want).max()))


@pytest.mark.parametrize("M,K,N", [(300, 512, 256), (2048, 4096, 4096), (64, 1024, 776), (2048, 4096, 11008), (2048, 8192, 256), (4096, 4096, 4096)])
@pytest.mark.parametrize("after", [None, "relu", "silu_mul"])
def test_linear_with_the_residual_add_in_its_stores(M, K, N, after):
    """residual + layer(x) (the add a decoder layer puts behind o_proj / fc2 / down_proj) as ONE launch on the per-block-exponent
    route (mi355q_bf16_gemm_tiled_res) and, since round 6, on the row-scale int8 route's one-launch form
    (mi355q_bfp_gemm_aligned_res): the same bits as the two steps -- the product's result rounded, then the add"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    dev = "cuda:0"
    for align in ("blocks", "rows"):
        cfg = _lin_cfg(6, mi355q_align=align)
        torch.manual_seed(M + N)
        lin = Q.get_quantized_cls("linear", cfg)(K, N, bias=True, config=cfg).to(dev)
        x = torch.randn(3 if M % 3 == 0 else 1, M // (3 if M % 3 == 0 else 1), K, device=dev)
        other = torch.randn_like(x) if after == "silu_mul" else None
        res = torch.randn(*x.shape[:-1], N, device=dev)
        with torch.no_grad():
            pre = (lambda t: t) if after is None else ((lambda t: torch.relu(t)) if after == "relu" else (lambda t: torch.nn.functional.silu(t) * other))
            lin(pre(x))                                         # first PTQ forward
            want = res + (lin(x) if after is None else lin.forward_after(x, after, other))
            calls, real = [], ops.bf16_gemm_tiled
            calls8, real8 = [], ops.bfp_gemm_aligned
            ops.bf16_gemm_tiled = lambda *a, **k: (calls.append(k.get("residual") is not None), real(*a, **k))[1]
            ops.bfp_gemm_aligned = lambda *a, **k: (calls8.append(k.get("residual") is not None), real8(*a, **k))[1]
            try:
                got = lin.forward_residual(x, res) if after is None else lin.forward_after(x, after, other, residual=res)
            finally:
                ops.bf16_gemm_tiled, ops.bfp_gemm_aligned = real, real8
        if align == "blocks":
            assert torch.equal(got, want)
        elif after is None:
            assert calls8 == [True] and torch.equal(got, want)          # (plain inputs fit their buckets: the add rode the int8 product's stores)
        else:
            # (post-ReLU inputs forced onto the row-scale route overflow their exception buckets: the launch's blockwise fallback adds
            #  back with fp32 atomics whose order is not fixed -- DESIGN 2 "Reproducibility" -- so two launches agree to the last bits only)
            torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
        assert calls == ([True] if align == "blocks" else []), (align, calls)
