"""Exception corrections formed by the PRODUCERS (include/mi355q.h, csrc/mi355q_corr.h; opt-in: ops.CORR): the row-scale product
whose add-back the activation quantiser wrote against the oracle's exact integer contraction (quantized_modules/linear.py:59-76)
and against the product that forms its add-back itself -- single launches, grouped launches, ragged shapes, and the cases
that must fall back on the device (more rows / columns with exception blocks than slots)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = dict(name="block_fp", is_ptq=True, bypass=False,
           data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
           weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
           bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])


def _inputs(M, N, K, seed=0, x_exc=0.0, w_exc=0.0, x_rows=None):
    import torch
    g = lambda s: torch.Generator().manual_seed(s + 10 * seed)
    x = torch.randn(M, K, generator=g(0)) * torch.exp(torch.randn(M, 1, generator=g(1)))
    w = torch.randn(N, K, generator=g(2)) * 0.02
    b = torch.randn(N, generator=g(3)) * 0.02
    rng = np.random.default_rng(seed)

    def spike(t, frac, rows=None):        # blocks pushed out of their row's exponent window
        if frac <= 0:
            return
        tb = t.view(t.shape[0], -1, 16)
        n = int(frac * tb.shape[0] * tb.shape[1])
        r = rng.integers(0, tb.shape[0] if rows is None else rows, n)
        c = rng.integers(0, tb.shape[1], n)
        tb[r, c] *= torch.tensor(2.0 ** rng.choice([-6, -5, 5, 6], n), dtype=torch.float32)[:, None]
    spike(x, x_exc, x_rows)
    spike(w, w_exc)
    return x, w, b


def _pack(ops, w, b, ww=6):
    import torch
    dev = torch.device("cuda:0")
    _, wm, we = ops.block_fp_quantize(w.to(dev), ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    return ops.bfp_align_rows(wm, we, ww - 1, 127), ops.block_fp_quantize(b.to(dev), ww, 8, 127, [16], False)


@pytest.fixture
def corr_ops(monkeypatch):
    from mi355q import ops
    monkeypatch.setattr(ops, "CORR", True)
    monkeypatch.setattr(ops, "REUSE_QUANTISED_INPUT", False)
    return ops


@pytest.mark.parametrize("tag,M,N,K,wx,ww,kw,bound,fits", [
    ("bench operands", 4096, 4096, 4096, 6, 6, {}, True, True),
    ("W4A4, no exception blocks", 4096, 4096, 4096, 4, 4, {}, True, True),
    ("W rows past their column slots", 4096, 4096, 1024, 6, 6, dict(x_exc=2e-4, w_exc=2e-3), True, False),
    ("x rows past their vector slots", 4096, 4096, 1024, 6, 6, dict(x_exc=4e-3), True, False),
    ("ragged, both sides", 2048 + 17, 4096 - 16, 1024, 6, 6, dict(x_exc=3e-4, w_exc=3e-4), True, True),
    ("128-row tiles: not bound", 300, 520, 512, 6, 6, dict(x_exc=1e-3, w_exc=1e-3), False, True),
])
def test_product_with_producer_formed_corrections(corr_ops, tag, M, N, K, wx, ww, kw, bound, fits):
    import torch
    from oracle import np_oracle as O
    ops = corr_ops
    x, w, b = _inputs(M, N, K, **kw)
    wa, bq = _pack(ops, w, b, ww)
    xt = x.to("cuda:0")
    y0 = ops.bfp_gemm_aligned(ops.block_fp_quantize_aligned_rows(xt, wx, 8, 127), wa, bq).clone()
    xb = ops.block_fp_quantize_aligned_rows(xt, wx, 8, 127, against=wa)
    assert (xb.corr is not None) == bound, tag
    y1 = ops.bfp_gemm_aligned(xb, wa, bq)
    torch.cuda.synchronize()
    assert int(xb.sparse[0]) == 0 and int(wa.sparse[0]) == 0            # (no bucket overflowed: the row-scale product applies)
    if bound:
        plan_bad, slots = ops.corr_plan(wa)[:2].tolist()
        assert ((plan_bad == 0) and int(xb.sparse[1]) == 0) == fits, (tag, plan_bad, slots, int(xb.sparse[1]))
    cfg = dict(CFG, data_in_width=wx, weight_width=ww, bias_width=ww)
    pick = np.sort(np.random.default_rng(1).choice(M, size=32, replace=False))
    pick[:3] = (0, 255, M - 1)
    ref = O.bfp_linear_int(x.numpy()[pick], w.numpy(), b.numpy(), cfg)
    sc = np.abs(ref).max()
    assert np.abs(y1.cpu().numpy()[pick] - ref).max() <= 1e-5 * sc, tag
    assert (y1 - y0).abs().max().item() <= 1e-5 * sc, tag               # the whole output against the in-tile add-back


@pytest.mark.parametrize("M,N,K,n", [(2048, 4096, 4096, 2), (4096, 2048, 2048, 3)])
def test_grouped_launch_with_producer_formed_corrections(corr_ops, M, N, K, n):
    import torch
    ops = corr_ops
    x, _, _ = _inputs(M, N, K, x_exc=1e-4)
    packs = [_pack(ops, *_inputs(M, N, K, seed=5 + i, w_exc=1e-4)[1:]) for i in range(n)]
    ws, bs = [p[0] for p in packs], [p[1] for p in packs]
    xt = x.to("cuda:0")
    xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127)
    singles = [ops.bfp_gemm_aligned(xa, wq, bq).clone() for wq, bq in zip(ws, bs)]
    xb = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, against=ws)
    assert xb.corr is not None
    grouped = ops.bfp_gemm_aligned_multi(xb, ws, bs)
    torch.cuda.synchronize()
    for i in range(n):
        assert (grouped[i] - singles[i]).abs().max().item() <= 1e-5 * singles[i].abs().max().item()
    # a launch against ONE weight of the group still finds its share of the binding
    one = ops.bfp_gemm_aligned(xb, ws[n - 1], bs[n - 1])
    assert (one - singles[n - 1]).abs().max().item() <= 1e-5 * singles[n - 1].abs().max().item()


def test_binding_is_ignored_for_other_weights(corr_ops):
    """x prepared against one weight, multiplied with another: the product launch forms its add-back itself"""
    import torch
    ops = corr_ops
    x, w, b = _inputs(1024, 4096, 1024, x_exc=3e-4, w_exc=3e-4)
    wa, bq = _pack(ops, w, b)
    wb, bq2 = _pack(ops, *_inputs(1024, 4096, 1024, seed=3, w_exc=3e-4)[1:])
    xt = x.to("cuda:0")
    ref = ops.bfp_gemm_aligned(ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127), wb, bq2).clone()
    xb = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, against=wa)
    got = ops.bfp_gemm_aligned(xb, wb, bq2)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
