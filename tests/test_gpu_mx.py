"""W4A4 / W5A5 block_fp Linear on the MX scaled matrix instruction (csrc/mi355q_mx.hip, through the C ABI) against the oracle's
exact integer contraction (reference: quantized_modules/linear.py:59-76 at the widths of configs/quantization/bfp_4bit.toml and
the section-4.4 search, configs/search/opt_1.3b_sst2.toml:24-37)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(wx, ww):
    return dict(name="block_fp", data_in_width=wx, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
                weight_width=ww, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
                bias_width=ww, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])


def _inputs(M, N, K, seed, style):
    r = np.random.default_rng(seed)
    x = r.normal(size=(M, K)).astype(np.float32)
    w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
    b = (r.normal(size=(N,)) * 0.02).astype(np.float32)
    if style == "rowscale":
        x *= np.exp(r.normal(size=(M, 1))).astype(np.float32)
    elif style == "blockscale":              # every [1,16] block its own magnitude, neighbours within the format's reach
        x *= np.repeat(2.0 ** r.integers(-2, 2, size=(M, K // 32)), 32, axis=1).astype(np.float32)
        x *= np.repeat(2.0 ** r.integers(0, 2, size=(M, K // 16)), 16, axis=1).astype(np.float32)
    elif style == "outlier":                 # neighbouring blocks far apart: the 32-group cannot share a scale
        x[:, ::53] *= 200.0
    elif style == "sparse":
        x[r.random((M, K)) < 0.6] = 0
        x[:, 32:80] = 0
    elif style == "w_outlier":
        w[::7, 5::41] *= 64.0
    return x, w, b


def _run(x, w, b, cfg):
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    xt = torch.from_numpy(x).to(dev)
    wt = torch.from_numpy(w).to(dev)
    wq = ops.block_fp_quantize(wt, cfg["weight_width"], 8, 127, [1, 16], False)      # the fake-quantised weights (the exact route reads them)
    bq = ops.block_fp_quantize(torch.from_numpy(b).to(dev), cfg["bias_width"], 8, 127, [16], False)
    # (from the RAW weights: block_fp is not idempotent -- a block whose largest quantised value is an exact power of two gets a
    #  smaller exponent, and a saturated mantissa, the second time: SURVEY quirk 7)
    wop = ops.block_fp_quantize_mx(wt, cfg["weight_width"], 8, 127, reuse=False)
    xop = ops.block_fp_quantize_mx(xt, cfg["data_in_width"], 8, 127)
    y = ops.mx_gemm(xop, wop, wq, bq)
    torch.cuda.synchronize()
    return y.cpu().numpy(), int(xop.bad[0]), int(wop.bad[0])


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 130, 1024), (100, 72, 384), (33, 16, 128), (520, 260, 512), (1024, 768, 4096)])
@pytest.mark.parametrize("style", ["randn", "rowscale", "blockscale", "sparse"])
@pytest.mark.parametrize("wx,ww", [(4, 4), (5, 5), (5, 4), (3, 2), (4, 5)])
def test_mx_gemm_vs_oracle(M, N, K, style, wx, ww):
    from oracle import np_oracle as O
    x, w, b = _inputs(M, N, K, 11000 + M + N + K, style)
    cfg = _cfg(wx, ww)
    y, xbad, wbad = _run(x, w, b, cfg)
    ref = O.bfp_linear_int(x, w, b, cfg)
    scale = np.abs(ref).max() + 1e-30
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * scale * max(1, K // 256))
    if style in ("randn", "rowscale") and wx <= 4 and ww <= 4:
        # (W4: a 32-group fits unless its blocks lie more than 3 exponents apart -- for Gaussian data one in 10^11; at W5 the reach
        #  is 2 exponents and one group in 10^7 does not fit: the launch takes the exact route then, the numbers above hold either way)
        assert xbad == 0 and wbad == 0, "these operands fit the format: the scaled MFMA must have formed the product"


@pytest.mark.parametrize("style", ["outlier", "w_outlier"])
@pytest.mark.parametrize("wx,ww", [(4, 4), (5, 5)])
def test_mx_gemm_groups_that_do_not_fit_take_the_exact_route(style, wx, ww):
    """a 32-group whose two blocks lie more than 3 (W4) / 2 (W5) exponents apart raises the operand's flag word; the launch
    forms the product from the fp32 tensors then (x quantised in registers): the same numbers"""
    from oracle import np_oracle as O
    M, N, K = 300, 130, 1024
    x, w, b = _inputs(M, N, K, 12345, style)
    cfg = _cfg(wx, ww)
    y, xbad, wbad = _run(x, w, b, cfg)
    assert (xbad if style == "outlier" else wbad) == 1
    ref = O.bfp_linear_int(x, w, b, cfg)
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6 * (np.abs(ref).max() + 1e-30) * 4)


def test_mx_flag_word_is_per_call():
    """the flag raised by one activation tensor does not stick to the next call on the same buffers"""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    x_bad, w, b = _inputs(256, 64, 512, 5, "outlier")
    x_ok, _, _ = _inputs(256, 64, 512, 6, "randn")
    seq = [x_bad, x_ok, x_ok, x_bad, x_ok]
    flags = []
    for x in seq:
        op = ops.block_fp_quantize_mx(torch.from_numpy(x).to(dev), 4, 8, 127)
        torch.cuda.synchronize()
        flags.append(int(op.bad[0]))
    assert flags == [1, 0, 0, 1, 0], flags


def _lin_cfg(wx, ww, **extra):
    return dict(_cfg(wx, ww), is_ptq=True, bypass=False, **extra)


def test_linear_block_fp_w4a4_takes_the_mx_route_at_full_size():
    """LinearBlockFP under the default policy: W4A4, a launch of 256 tiles of 256 x 256 -> the MX scaled MFMA; the same layer at
    2048 tokens x 2048 (64 tiles) stays on the int8 kernels; W6A6 never takes it.  Outputs against the oracle on sampled rows."""
    import torch
    import mi355q.quantize as Q
    from mi355q.quantize.quantized_modules import linear as L
    from oracle import np_oracle as O
    dev = torch.device("cuda:0")
    calls, real = [], L.ops.mx_gemm
    L.ops.mx_gemm = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        for (M, K, N, wx, ww, expect) in ((4096, 512, 4096, 4, 4, True), (2048, 512, 2048, 4, 4, False), (4096, 512, 4096, 6, 6, False),
                                          (4096, 512, 4096, 3, 4, True)):
            cfg = _lin_cfg(wx, ww)
            torch.manual_seed(M + N + wx)
            fp = torch.nn.Linear(K, N)
            with torch.no_grad():
                fp.weight.mul_(8.0)
            w0, b0 = fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()
            lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to(dev)
            x = torch.randn(M, K) * torch.exp(torch.randn(M, 1))
            calls.clear()
            with torch.no_grad():
                y1 = lin(x.to(dev))
                y2 = lin(x.to(dev))
            assert (len(calls) == 2) == expect, (M, K, N, wx, ww, len(calls))
            assert torch.equal(y1, y2)
            pick = np.sort(np.random.default_rng(3).choice(M, size=48, replace=False))
            ref = O.bfp_linear_int(x.numpy()[pick], w0, b0, cfg)
            err = np.abs(y1.cpu().numpy()[pick] - ref).max() / np.abs(ref).max()
            assert err < 5e-6, (M, K, N, wx, ww, err)
    finally:
        L.ops.mx_gemm = real


def test_linear_block_fp_leaves_the_mx_route_when_the_activations_do_not_fit():
    """activations with outlier channels: some 32-group of every row lies more than three exponents apart -- the first call is
    exact all the same (the launch's own exact route) and the policy moves the layer to the other kernels"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    dev = torch.device("cuda:0")
    M, K, N = 512, 256, 384
    cfg = _lin_cfg(4, 4, mi355q_mx=True)
    torch.manual_seed(9)
    fp = torch.nn.Linear(K, N)
    w0, b0 = fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to(dev)
    x = torch.randn(M, K)
    x[:, ::53] *= 200.0
    ref = O.bfp_linear_int(x.numpy(), w0, b0, cfg)
    with torch.no_grad():
        y1 = lin(x.to(dev))
        assert lin._mx_w is None, "the flag word was up: the layer should have left the route"
        y2 = lin(x.to(dev))
    for y in (y1, y2):
        assert np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max() < 5e-6


def test_release_fp32_weight_takes_the_layer_off_the_mx_route():
    """release_fp32_weight() on a W4A4 layer whose launches are large enough for the MX route (>= 192 tiles).  The MX
    product's exact in-launch fallback reads the fp32 weights, so the MX operand goes with them: the forward after the release runs
    on the int8 operand, equals the forward before it to the oracle's tolerance, and frees the fp32 storage (no alias kept)."""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    dev = torch.device("cuda:0")
    M, K, N = 4096, 512, 4096
    cfg = _lin_cfg(4, 4)
    torch.manual_seed(21)
    fp = torch.nn.Linear(K, N)
    with torch.no_grad():
        fp.weight.mul_(8.0)
    w0, b0 = fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to(dev)
    x = (torch.randn(M, K) * torch.exp(torch.randn(M, 1))).to(dev)
    with torch.no_grad():
        lin.pack_now(x)
        assert lin._mx_w is not None and lin._mx_w.source is None, "a weights' MX operand keeps no alias of the fp32 storage"
        y_mx = lin(x)
        lin.release_fp32_weight()
        assert lin._mx_w is None and lin.weight.numel() == 0 and not lin._mx_takes(x)
        y_int8 = lin(x)
    pick = np.sort(np.random.default_rng(5).choice(M, size=48, replace=False))
    ref = O.bfp_linear_int(x.cpu().numpy()[pick], w0, b0, cfg)
    for y in (y_mx, y_int8):
        assert np.abs(y.cpu().numpy()[pick] - ref).max() / np.abs(ref).max() < 5e-6
