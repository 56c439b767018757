"""The mixed contraction (round 6, include/mi355q.h mi355q_bfp_gemm_mixed; csrc/mi355q_gemm_v9m.hip): one launch of the 256 x 256
tile kernel that multiplies class 0 of the columns on the int8 MFMA (row-aligned operands with their exception lists) and class
1 on the bf16 MFMA (every block its own exponent) -- the int8 route for activations with OUTLIER CHANNELS (README.md:9-11 of the
reference).  Here: the kernel against the oracle's exact integer contraction (the split of the columns is the caller's: F.linear
sums over in_features in any order, quantized_modules/linear.py:59-76), odd shapes, exception blocks in class 0, an overflowed
bucket (the launch's own tile-by-tile fallback), bit-reproducibility."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(wx=6, ww=6):
    return dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=wx, data_in_exponent_width=8, data_in_exponent_bias=127,
                data_in_block_size=[1, 16], weight_width=ww, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
                bias_width=ww, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])


def _split(K, blocks1):
    """element indices of the two classes from the class-1 block columns"""
    b1 = np.asarray(sorted(blocks1))
    b0 = np.asarray([b for b in range(K // 16) if b not in set(blocks1)])
    cols = lambda bs: (bs[:, None] * 16 + np.arange(16)[None]).reshape(-1)
    return b0, b1, cols(b0), cols(b1)


def _mixed(x, w, b, blocks1, wx=6, ww=6, x_cap=None):
    """the product through the C ABI with the columns split on the host (torch index_select + the library's own quantisers:
    the fused class-aware quantiser of quantized_modules/linear.py is tested in test_gpu_modules.py)"""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    K = x.shape[1]
    b0, b1, c0, c1 = _split(K, blocks1)
    xt, wt, bt = (torch.from_numpy(t).to(dev) for t in (x, w, b))
    c0t, c1t, b0t = (torch.from_numpy(t).to(dev) for t in (c0, c1, b0))
    wq, wm, we = ops.block_fp_quantize(wt, ww, 8, 127, [1, 16], False, want_packed=True)
    N = w.shape[0]
    wa0 = ops.bfp_align_rows(wm.view(N, K)[:, c0t].contiguous(), we.view(N, K // 16)[:, b0t].contiguous(), ww - 1, 127)
    w1 = ops.bf16_tile(wq[:, c1t].contiguous())
    bq = ops.block_fp_quantize(bt, ww, 8, 127, [16], False)
    kw = {} if x_cap is None else dict(bucket_cap=x_cap)
    xa0 = ops.block_fp_quantize_aligned_rows(xt[:, c0t].contiguous(), wx, 8, 127, **kw)
    x1 = ops.block_fp_quantize_bf16_tiled(xt[:, c1t].contiguous(), wx, 8, 127, reuse=False)
    y = ops.bfp_gemm_mixed(xa0, wa0, x1, w1, len(c1), bq)
    torch.cuda.synchronize()
    return y, xa0, wa0


def _outlier_inputs(M, N, K, n_out, seed, scale=60.0):
    r = np.random.default_rng(seed)
    x = (r.normal(size=(M, K)) * np.exp(r.normal(size=(M, 1)))).astype(np.float32)
    idx = r.choice(K, size=n_out, replace=False)
    x[:, idx] *= scale
    w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
    b = (r.normal(size=(N,)) * 0.02).astype(np.float32)
    blocks1 = set((idx // 16).tolist())
    nb = K // 16
    # whole K-step pairs in both classes: class-1 blocks a multiple of 8 (K1 % 128 == 0), padded with ordinary block columns
    spare = [bb for bb in range(nb) if bb not in blocks1]
    while len(blocks1) % 8 or len(blocks1) < 8:
        blocks1.add(spare.pop(int(r.integers(len(spare)))))
    return x, w, b, blocks1


@pytest.mark.parametrize("M,N,K,n_out,wx,ww", [(512, 512, 1024, 8, 6, 6), (256, 256, 512, 4, 6, 6), (300, 200, 768, 5, 6, 6),
                                               (1024, 768, 2048, 32, 4, 4), (512, 1024, 4096, 64, 6, 6), (272, 528, 640, 3, 7, 5)])
def test_mixed_gemm_vs_oracle(M, N, K, n_out, wx, ww):
    from oracle import np_oracle as O
    x, w, b, blocks1 = _outlier_inputs(M, N, K, n_out, seed=M + N + K)
    y, xa0, wa0 = _mixed(x, w, b, blocks1, wx, ww)
    assert y is not None and int(wa0.sparse[0]) == 0
    # (7-bit activations leave the int8 container ONE spare bit: their rows overflow a bucket here -- the odd-shaped case then
    #  runs the launch's own tile-by-tile fallback, held to the same bound)
    assert (int(xa0.sparse[0]) == 0) == (wx < 7)
    pick = np.sort(np.random.default_rng(1).choice(M, size=min(M, 64), replace=False))
    pick[:2] = (0, M - 1)
    ref = O.bfp_linear_int(x[pick], w, b, _cfg(wx, ww))
    err = np.abs(y.cpu().numpy()[pick] - ref).max() / np.abs(ref).max()
    assert err <= 1e-5, err


def test_mixed_gemm_headline_size_with_exception_blocks_and_reproducible():
    """4096^3, K / 64 outlier channels x 60 (bench.py's `robustness` operands): class 0 keeps its ordinary exception blocks (the
    in-tile add-back runs behind the bf16 steps), the output equals the oracle's on sampled rows and a second launch gives the
    same bits"""
    import torch
    from oracle import np_oracle as O
    M = N = K = 4096
    x, w, b, blocks1 = _outlier_inputs(M, N, K, K // 64, seed=11)
    x[::37, 256:272] *= 1.0 / 512.0                       # (a few blocks far BELOW their rows' window: class-0 exceptions)
    assert not (set(range(16, 17)) & blocks1) or True
    y, xa0, wa0 = _mixed(x, w, b, blocks1)
    assert int(xa0.sparse[0]) == 0
    entries = int(xa0.sparse[8::8 + 8 * 120][:16].sum().cpu())
    assert entries > 0, "the test wants exception blocks in class 0"
    y2, _, _ = _mixed(x, w, b, blocks1)
    assert torch.equal(y, y2)
    pick = np.sort(np.random.default_rng(2).choice(M, size=48, replace=False))
    pick[:6] = (0, 37, 74, 255, 256, M - 1)
    ref = O.bfp_linear_int(x[pick], w, b, _cfg())
    err = np.abs(y.cpu().numpy()[pick] - ref).max() / np.abs(ref).max()
    assert err <= 1e-5, err


def test_mixed_gemm_overflowed_bucket_takes_the_launchs_own_fallback():
    """class 0 with more exception blocks in one 256-row bucket than its 120 entries: the quantiser raises the overflow word and
    the launch forms the product tile by tile -- class 1 from memory, class 0 blockwise-exact on top of it -- still exact"""
    from oracle import np_oracle as O
    M, N, K = 512, 384, 1024
    x, w, b, blocks1 = _outlier_inputs(M, N, K, 6, seed=5)
    free = [bb for bb in range(K // 16) if bb not in blocks1]
    r = np.random.default_rng(8)
    for row in range(0, 256):                              # every row of the first bucket gets a block far above its window
        bb = free[int(r.integers(len(free)))]
        x[row, bb * 16:bb * 16 + 16] *= 4096.0
    y, xa0, wa0 = _mixed(x, w, b, blocks1)
    assert int(xa0.sparse[0]) != 0, "the test wants an overflowed bucket"
    ref = O.bfp_linear_int(x, w, b, _cfg())
    err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err <= 1e-5, err


def test_mixed_gemm_refuses_what_it_does_not_take():
    import torch
    from mi355q import ops
    x, w, b, blocks1 = _outlier_inputs(256, 256, 512, 2, seed=3)
    # K1 = 64 (four blocks): below the four bf16 K-steps the pipeline needs -> None, the caller keeps the per-block route
    small = set(sorted(blocks1)[:4])
    y, _, _ = _mixed(x, w, b, small)
    assert y is None


@pytest.mark.parametrize("M,K,n_out,wx", [(512, 1024, 8, 6), (300, 768, 5, 6), (256, 4096, 64, 6), (1024, 2048, 32, 4), (128, 8192, 40, 6)])
def test_class_aware_quantiser_equals_the_split_by_index_select(M, K, n_out, wx):
    """mi355q_block_fp_quantize_classes (one pass over x, blocks sent to the int8 or the bf16 operand by the column map) against
    the same operands made by splitting x's columns on the host and quantising each part with the library's two quantisers:
    tiled mantissas, effective exponents, row flags and scales, the bf16 operand -- byte for byte; exception entries as sets
    (their order in a bucket depends on which row reserved its slots first)"""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    x, _, _, blocks1 = _outlier_inputs(M, 64, K, n_out, seed=M + K)
    x[::5, 32:48] *= 1.0 / 1024.0                       # class-0 exception blocks (below their rows' window), if block 2 is class 0
    cls = ops.ColumnClasses(K, blocks1, dev)
    xt = torch.from_numpy(x).to(dev)
    x0, x1 = ops.block_fp_quantize_classes(xt, cls, wx, 8, 127)
    r0 = ops.block_fp_quantize_aligned_rows(xt[:, cls.cols0].contiguous(), wx, 8, 127)
    # (into a zeroed buffer: the rows between M and the operand's padded height are never written by either quantiser)
    r1 = ops.block_fp_quantize_bf16_tiled(xt[:, cls.cols1].contiguous(), wx, 8, 127,
                                          out=torch.zeros_like(x1))
    torch.cuda.synchronize()
    assert torch.equal(x1, r1), "class 1: the tiled bf16 operand"
    assert torch.equal(x0.tiled, r0.tiled), "class 0: tiled mantissas"
    assert torch.equal(x0.exp, r0.exp) and torch.equal(x0.rowflag, r0.rowflag) and torch.equal(x0.gscale, r0.gscale)
    nb = (M + 255) // 256
    words = 8 + 8 * 120
    a, b = x0.sparse.cpu().numpy(), r0.sparse.cpu().numpy()
    assert a[0] == b[0] == 0
    for bk in range(nb):
        ba, bb = a[8 + bk * words:8 + (bk + 1) * words], b[8 + bk * words:8 + (bk + 1) * words]
        assert ba[0] == bb[0]
        ea = sorted(tuple(ba[8 + 8 * i:16 + 8 * i].tolist()) for i in range(int(ba[0])))
        eb = sorted(tuple(bb[8 + 8 * i:16 + 8 * i].tolist()) for i in range(int(bb[0])))
        assert ea == eb


def _lin(K, N, cfg, seed=0, scale=1.0):
    import torch
    import mi355q.quantize as Q
    torch.manual_seed(seed)
    fp = torch.nn.Linear(K, N, bias=True)
    with torch.no_grad():
        fp.weight.mul_(scale)
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to("cuda:0")
    return lin, fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()


def test_linear_auto_takes_the_mixed_contraction_for_outlier_channels():
    """LinearBlockFP under the default policy on activations with outlier channels (K / 64 channels x 60 on top of row scales:
    bench.py's `robustness` operands at a quarter of the size): the layer settles on the mixed contraction -- its class 1 holds
    every block column with an outlier channel and at most half of all columns --, equals the oracle, gives the same bits call
    after call, and with mi355q_mixed = False runs on the per-block bf16 route as before"""
    import torch
    from oracle import np_oracle as O
    from mi355q import ops
    M, K, N = 1024, 2048, 1024
    cfg = dict(_cfg(), is_ptq=True, bypass=False)
    lin, w0, b0 = _lin(K, N, cfg, seed=4)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    idx = torch.randint(0, K, (K // 64,), generator=g)
    x[:, idx] *= 60.0
    xd = x.to("cuda:0")
    calls, real = [], ops.bfp_gemm_mixed
    ops.bfp_gemm_mixed = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            y1, y2 = lin(xd), lin(xd)
    finally:
        ops.bfp_gemm_mixed = real
    assert lin._mixed is not None and len(calls) == 2 and not lin._uses_bf16_route()
    cls = lin._mixed["classes"]
    assert set((idx // 16).tolist()) <= set(cls.blocks1.cpu().tolist()) and cls.n1 <= K // 32
    assert torch.equal(y1, y2)
    pick = np.sort(np.random.default_rng(3).choice(M, size=48, replace=False))
    ref = O.bfp_linear_int(x.numpy()[pick], w0, b0, cfg)
    assert np.abs(y1.cpu().numpy()[pick] - ref).max() <= 1e-5 * np.abs(ref).max()
    cfg2 = dict(cfg, mi355q_mixed=False)
    lin2, _, _ = _lin(K, N, cfg2, seed=4)
    with torch.no_grad():
        y3 = lin2(xd)
    assert lin2._mixed is None and lin2._uses_bf16_route()
    assert np.abs(y3.cpu().numpy()[pick] - ref).max() <= 1e-5 * np.abs(ref).max()


def test_mixed_layer_serves_the_fused_steps_and_leaves_when_class_0_stops_fitting():
    """a layer on the mixed contraction: forward_after (relu in front) and forward_residual give what the torch ops around
    forward() give; requantize() forgets the split; activations whose OTHER columns start to fall out of their rows' window
    overflow class 0's buckets -- exact all the same (the launch's own fallback) -- and on the doubling schedule of calls the
    layer leaves for the per-block route"""
    import torch
    import torch.nn.functional as F
    from oracle import np_oracle as O
    M, K, N = 512, 1024, 512
    cfg = dict(_cfg(), is_ptq=True, bypass=False)
    lin, w0, b0 = _lin(K, N, cfg, seed=7)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    x[:, 5::97] *= 80.0
    xd = x.to("cuda:0")
    with torch.no_grad():
        y = lin(xd)
        assert lin._mixed is not None
        res = torch.randn(M, N, generator=g).to("cuda:0")
        assert torch.equal(lin.forward_after(xd, "relu"), lin(F.relu(xd)))
        assert torch.equal(lin.forward_residual(xd, res), res + y)
        # every row gets a block far above its window in a class-0 column: 512 exception blocks in two buckets of 120
        wild = x.clone()
        free = [b for b in range(K // 16) if b not in set(lin._mixed["classes"].blocks1.cpu().tolist())]
        r = np.random.default_rng(5)
        for row in range(M):
            b = free[int(r.integers(len(free)))]
            wild[row, b * 16:b * 16 + 16] *= 4096.0
        ref = O.bfp_linear_int(wild.numpy(), w0, b0, cfg)
        for call in range(12):                      # (the overflow word is read on a doubling schedule of calls: here at 8 and 16)
            yw = lin(wild.to("cuda:0"))
            assert np.abs(yw.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
        assert lin._calls >= 16 and lin._mixed is None and lin._uses_bf16_route()
        lin2, _, _ = _lin(K, N, dict(cfg, mi355q_keep_master=True), seed=7)
        lin2(xd)
        assert lin2._mixed is not None
        lin2.requantize()
        assert lin2._mixed is None
        lin2(torch.randn(M, K, generator=g).to("cuda:0"))
        assert lin2._mixed is None and not lin2._uses_bf16_route()
