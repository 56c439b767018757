"""The library's own product kernels for block_minifloat and block_log attention operands (ABI 19:
mi355q_block_minifloat_matmul / _softmax_matmul, mi355q_block_log_matmul) behind the registry's matmul / bmm functions
(reference quantized_functions/matmul.py:199-249, 252-297, 300-353), against the oracle's float64 evaluation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FMIN = np.finfo(np.float32).min


def _cfg(arith, width=8, ew=4, ebw=8, yw=None):
    if arith == "block_minifloat":
        return dict(name="block_minifloat", bypass=False, data_in_width=width, data_in_exponent_width=ew, data_in_exponent_bias_width=ebw,
                    data_in_block_size=[1, 16], weight_width=yw or width, weight_exponent_width=ew, weight_exponent_bias_width=ebw,
                    weight_block_size=[1, 16])
    return dict(name="block_log", bypass=False, data_in_width=width, data_in_exponent_bias_width=ebw, data_in_block_size=[1, 16],
                weight_width=width, weight_exponent_bias_width=ebw, weight_block_size=[1, 16])


def _spy(ops):
    calls, real = [], ops.values_matmul
    ops.values_matmul = lambda *a, **kw: (calls.append(kw.get("softmax", False)), real(*a, **kw))[1]
    return calls, real


def _probs(r, lead, T, zero_blocks):
    s = (r.normal(size=(*lead, T, T)) * 3).astype(np.float32)
    if zero_blocks:                                   # a causal mask: exact zeros, half of the [1,16] blocks all zero
        s = np.maximum(s + np.triu(np.full((T, T), FMIN, np.float32), 1), FMIN)
    e = np.exp(s - s.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("style", ["matmul", "bmm"])
@pytest.mark.parametrize("arith,width,ew,ebw", [("block_minifloat", 8, 4, 8), ("block_minifloat", 6, 3, 4), ("block_minifloat", 4, 2, 3),
                                                ("block_log", 8, 0, 8), ("block_log", 4, 0, 3), ("block_log", 6, 0, 2)])
@pytest.mark.parametrize("T,hd", [(160, 64), (320, 128), (48, 32), (208, 96)])
def test_attention_products(style, arith, width, ew, ebw, T, hd):
    """P V (long contraction, split over the waves) and Q K^T (short: quantised x resident) at attention shapes, with the
    causal mask's all-zero blocks in P -- block_log gives those the reference's tensor-wide fill (block_fp.py:54-58)"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    from oracle import np_oracle as O
    cfg = _cfg(arith, width, ew, ebw)
    r = np.random.default_rng(T * 7 + hd + width)
    lead = (2, 3) if style == "matmul" else (5,)
    f = Q.get_quantized_func(style, cfg)
    calls, real = _spy(ops)
    try:
        cases = ((_probs(r, lead, T, True), r.normal(size=(*lead, T, hd)).astype(np.float32)),
                 ((r.normal(size=(*lead, T, hd)) * 0.7).astype(np.float32), r.normal(size=(*lead, hd, T)).astype(np.float32)),
                 (_probs(r, lead, T, False) * np.float32(40), (r.normal(size=(*lead, T, hd)) * np.exp(r.normal(size=(*lead, T, 1)) * 2)).astype(np.float32)))
        for x, y in cases:
            n0 = len(calls)
            got = f(torch.from_numpy(x).to("cuda:0"), torch.from_numpy(y).to("cuda:0"), dict(cfg)).cpu().numpy()
            assert len(calls) == n0 + 1, "the fused product kernel was not taken"
            ref = O.matmul_quantized(x, y, cfg)
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
    finally:
        ops.values_matmul = real


@pytest.mark.parametrize("arith", ["block_minifloat", "block_log"])
@pytest.mark.parametrize("B,M,K,N", [(1, 1, 16, 16), (3, 40, 48, 16), (2, 17, 192, 80), (2, 100, 208, 144), (1, 33, 1040, 128),
                                     (2, 16, 64, 272), (1, 250, 2048, 128), (3, 130, 128, 336), (2, 129, 64, 2048)])
def test_ragged_shapes(arith, B, M, K, N):
    """rows that do not fill a 16-row workgroup, contractions that are not whole 64-steps, every resident depth (1, 2, 3
    steps) and the streaming kernel, more columns than one chunk / than the ring holds chunks"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    cfg = _cfg(arith, 7, 3, 5) if arith == "block_minifloat" else _cfg(arith, 5, 0, 3)
    r = np.random.default_rng(M + K + N)
    x = (r.normal(size=(B, M, K)) * np.exp(r.normal(size=(B, M, 1)))).astype(np.float32)
    if K:
        x[:, :, :16] = 0                                  # an all-zero block in every row
        x[:, ::3, -16:] *= 1e-6
    y = r.normal(size=(B, K, N)).astype(np.float32)
    keys = ("width", "exponent_width", "exponent_bias_width") if arith == "block_minifloat" else ("width", "exponent_bias_width")
    xp = tuple(cfg[f"data_in_{k}"] for k in keys)
    yp = tuple(cfg[f"weight_{k}"] for k in keys) if arith == "block_minifloat" else None
    xt, yt = torch.from_numpy(x).to("cuda:0"), torch.from_numpy(y).to("cuda:0")
    got = ops.values_matmul(xt, yt, arith, xp, yp).cpu().numpy()
    ref = O.matmul_quantized(x, y, cfg)
    assert got.shape == ref.shape
    if ref.size:
        assert np.abs(got - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize("style,lead", [("matmul", (2, 2)), ("bmm", (3,))])
@pytest.mark.parametrize("T,hd,mode", [(320, 64, "causal"), (256, 128, "mask"), (520, 64, "both"), (100, 64, "causal")])
def test_softmax_folded_into_the_block_minifloat_product(style, lead, T, hd, mode):
    """softmax_{matmul,bmm}_block_minifloat(scores, v, mask / causal) == {matmul,bmm}_block_minifloat(softmax(max(scores +
    mask, finfo.min)), v): the reference's steps between its two products (modeling_opt.py:262-312,
    modeling_llama.py:318-344) in one call; rows shorter than the fused kernel takes fall back to the same steps"""
    import torch
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    cfg = _cfg("block_minifloat", 8, 4, 8)
    r = np.random.default_rng(T + hd)
    s = (r.normal(size=(*lead, T, T)) * 3).astype(np.float32)
    v = r.normal(size=(*lead, T, hd)).astype(np.float32)
    causal_mask = np.triu(np.full((T, T), FMIN, np.float32), 1)
    add_mask = (r.normal(size=(T, T)) * 0.5).astype(np.float32)
    add_mask[:, ::7] = FMIN
    add_mask[:, 0] = 0
    m = {"causal": causal_mask, "mask": add_mask, "both": np.maximum(add_mask + causal_mask, FMIN)}[mode]
    w = np.maximum(s + m, FMIN)
    e = np.exp(w - w.max(-1, keepdims=True))
    ref = O.matmul_quantized((e / e.sum(-1, keepdims=True)).astype(np.float32), v, cfg)
    st, vt = torch.from_numpy(s).to("cuda:0"), torch.from_numpy(v).to("cuda:0")
    kw = {"causal": dict(causal=True), "mask": dict(mask=torch.from_numpy(add_mask).to("cuda:0")),
          "both": dict(mask=torch.from_numpy(add_mask).to("cuda:0"), causal=True)}[mode]
    out = Q.get_quantized_func("softmax_" + style, cfg)(st, vt, cfg, **kw).cpu().numpy()
    scale = np.abs(ref).max()
    # (a probability one ulp apart may round to the next mantissa: one step of one of T terms of a row)
    assert np.abs(out - ref).max() <= 2e-3 * scale and np.abs(out - ref).mean() <= 2e-5 * scale


def test_routes_that_decline():
    """what the fused kernels do not take goes to the next route with the same answer: block sizes other than [1,16], > 7
    mantissa bits (block_minifloat), an explicit config["mi355q_values_matmul"]"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    from oracle import np_oracle as O
    r = np.random.default_rng(5)
    x = r.normal(size=(3, 64, 64)).astype(np.float32)
    y = r.normal(size=(3, 64, 32)).astype(np.float32)
    xt, yt = torch.from_numpy(x).to("cuda:0"), torch.from_numpy(y).to("cuda:0")
    calls, real = _spy(ops)
    try:
        for cfg in (dict(_cfg("block_minifloat"), data_in_block_size=[1, 32]), dict(_cfg("block_minifloat", 14, 4, 8)),
                    dict(_cfg("block_minifloat"), mi355q_values_matmul="bf16"), dict(_cfg("block_log"), mi355q_values_matmul="fp32"),
                    dict(_cfg("block_log"), data_in_block_size=[2, 16])):
            got = Q.get_quantized_func("bmm", cfg)(xt, yt, dict(cfg)).cpu().numpy()
            assert not calls
            ref = O.matmul_quantized(x, y, cfg)
            assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
    finally:
        ops.values_matmul = real
