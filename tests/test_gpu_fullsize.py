"""Parity at the sizes BASELINE.json names (config 2: synthetic 4096 x 4096 block_fp W6A6, "bit-exact exponent check";
config 4's row-sharded shapes): quantiser integers bit for bit against the oracle over the whole tensor, the
steady-state step (fused quantise + row-align -> int8 GEMM) on sampled rows x all columns against the oracle's
exact integer contraction, and the size-independent properties the path offers (row permutation equivariance over the
whole output, run-to-run bit reproducibility, the quantiser applied to its own output)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = dict(name="block_fp", is_ptq=True, bypass=False,
           data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
           weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
           bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])


def _bench_inputs(M=4096, N=4096, K=4096):
    import torch
    g = lambda s: torch.Generator().manual_seed(s)
    x = torch.randn(M, K, generator=g(0)) * torch.exp(torch.randn(M, 1, generator=g(1)))
    w = torch.randn(N, K, generator=g(2)) * 0.02
    b = torch.randn(N, generator=g(3)) * 0.02
    return x, w, b


@pytest.mark.parametrize("width", [6, 4])
def test_quantiser_integers_bit_exact_at_4096x4096(width):
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    x, w, _ = _bench_inputs()
    for t, skip in ((x, True), (w, False)):
        fq, m, e = ops.block_fp_quantize(t.to("cuda:0"), width, 8, 127, [1, 16], skip, want_packed=True)
        code = O.bfp_encode(t.numpy(), width, 8, 127, [1, 16], skip)
        assert np.array_equal(e.cpu().numpy().astype(np.int32) - 127, code.exp)
        assert np.array_equal(m.cpu().numpy().reshape(-1, 16).astype(np.int32), code.mant)
        assert np.array_equal(fq.cpu().numpy(), O.block_fp_quantize(t.numpy(), width, 8, 127, [1, 16], skip))
        # a second application (inputs that sit exactly on grid points, exact powers of two and rounding ties; NOT a
        # fixed point: a block whose max rounds to 2^k is re-scaled and saturates) -- bit for bit
        again = ops.block_fp_quantize(fq, width, 8, 127, [1, 16], skip)
        assert np.array_equal(again.cpu().numpy(), O.block_fp_quantize(fq.cpu().numpy(), width, 8, 127, [1, 16], skip))


def _step(x, w, b, wx=6, ww=6, x_cap=None, out=None):
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    xt, wt, bt = x.to(dev), w.to(dev), b.to(dev)
    _, wm, we = ops.block_fp_quantize(wt, ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
    wa = ops.bfp_align_rows(wm, we, ww - 1, 127)
    bq = ops.block_fp_quantize(bt, ww, 8, 127, [16], False)
    xa = ops.block_fp_quantize_aligned_rows(xt, wx, 8, 127) if x_cap is None else \
        ops.block_fp_quantize_aligned_rows(xt, wx, 8, 127, bucket_cap=x_cap)
    y = ops.bfp_gemm_aligned(xa, wa, bq, out=out)
    torch.cuda.synchronize()
    return y, xa, wa


@pytest.mark.parametrize("M,N,K,wx,ww", [(4096, 4096, 4096, 6, 6), (4096, 4096, 4096, 4, 4),
                                           (4096, 512, 4096, 6, 6),      # the P = 8 shard of the benchmark layer
                                           (2048, 1024, 8192, 4, 4),     # OPT-1.3B fc2 shard at P = 2, W4A4 (config 4)
                                           (2048, 11008, 4096, 6, 6)])   # Llama-7B up_proj
def test_step_sampled_rows_vs_oracle_at_full_size(M, N, K, wx, ww):
    from oracle import np_oracle as O
    x, w, b = _bench_inputs(M, N, K)
    y, xa, wa = _step(x, w, b, wx, ww)
    assert int(xa.sparse[0]) == 0 and int(wa.sparse[0]) == 0, "benchmark operands must stay on the fast path"
    cfg = dict(CFG, data_in_width=wx, weight_width=ww, bias_width=ww)
    pick = np.sort(np.random.default_rng(M + N).choice(M, size=48, replace=False))
    pick[:4] = (0, 255, 256, M - 1)                                   # tile edges
    ref = O.bfp_linear_int(x.numpy()[pick], w.numpy(), b.numpy(), cfg)
    got = y.cpu().numpy()[pick]
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_step_properties_at_4096():
    """size-independent checks over the WHOLE output: permuting rows of x permutes rows of y bit for bit (each row
    lands in a different tile / workgroup / XCD after the permutation); a second run is bit-identical."""
    import torch
    x, w, b = _bench_inputs()
    zero_b = torch.zeros_like(b)
    y0, _, _ = _step(x, w, zero_b)
    y0 = y0.clone()
    y1, _, _ = _step(x, w, zero_b)
    assert torch.equal(y0, y1)
    # (scaling rows by powers of two is NOT an exact symmetry of the reference: it adds 1e-9 to |x| before the
    #  mantissa is formed, block_fp.py:69-71, which moves a rounding here and there for |x| < 0.02)
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(9))
    yp, _, _ = _step(x[perm], w, zero_b)
    assert torch.equal(yp.cpu(), y0.cpu()[perm])


def test_column_slice_output_matches_full():
    """a rank of the row-sharded layer writes its [M, N/P] result into its column slice of the full buffer (ldy)"""
    import torch
    x, w, b = _bench_inputs(1024, 2048, 1024)
    full, _, _ = _step(x, w, b)
    full = full.clone()
    buf = torch.zeros(1024, 2048, device="cuda:0")
    for r in range(4):
        lo, hi = r * 512, (r + 1) * 512
        _step(x, w[lo:hi], b[lo:hi], out=buf[:, lo:hi])
    assert torch.equal(buf, full)


_C5 = {"block_minifloat": dict(width=8, exponent_width=4, exponent_bias_width=8, block_size=[1, 16]),
       "block_log": dict(width=8, exponent_bias_width=8, block_size=[1, 16]),
       "block_fp": dict(width=6, exponent_width=8, exponent_bias=127, block_size=[1, 16])}


@pytest.mark.parametrize("shape", [(2048, 4096), (2048, 11008)])
@pytest.mark.parametrize("name", ["block_minifloat", "block_log"])
def test_config5_activation_shapes_bit_exact(name, shape):
    """BASELINE config 5's activation shapes (Llama-7B hidden / intermediate width, 2048 tokens), the bench's own inputs
    (bench.py quantizer_workload: randn * 4, seed 7): every output word == the oracle's"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(7)) * 4.0
    got = getattr(ops, name + "_quantize")(x.to("cuda:0"), *[_C5[name][k] for k in _C5[name]], True).cpu().numpy()
    want = np.asarray(getattr(O, name + "_quantize")(x.numpy(), **_C5[name], skip_first_dim=True), dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name", ["block_minifloat", "block_log", "block_fp"])
def test_config5_causal_probabilities_planes_bit_exact(name):
    """the worst case of config 5 at its full size: softmax rows under the causal mask, [32, 2048, 2048] (the zero-block-map
    path: 512 MiB) -- 5 of the 32 planes word for word against the oracle.  The fill of an all-zero block is the smallest
    non-zero block maximum of the WHOLE tensor (block_fp.py:54-58 and the same lines of block_minifloat.py / block_log.py):
    the oracle quantises each plane together with the one row that carries it"""
    import torch
    from mi355q import ops
    from oracle import np_oracle as O
    shp = (32, 2048, 2048)
    dev = "cuda:0"
    x = torch.randn(*shp, generator=torch.Generator().manual_seed(7)).to(dev) * 4.0
    p = torch.softmax(x + torch.full(shp[-2:], float("-inf"), device=dev).triu(1), dim=-1)
    del x
    got = getattr(ops, name + "_quantize")(p, *[_C5[name][k] for k in _C5[name]], True)
    bm = p.view(-1, 16).abs().amax(1)
    row = int((bm == bm[bm > 0].min()).nonzero()[0, 0]) * 16 // shp[-1]
    carrier = p.view(-1, shp[-1])[row:row + 1].cpu().numpy()
    assert (p[3] == 0).float().mean().item() > 0.45
    for i in (0, 3, 13, 22, 31):
        piece = np.concatenate([p[i].cpu().numpy(), carrier], 0)[None]
        want = np.asarray(getattr(O, name + "_quantize")(piece, **_C5[name], skip_first_dim=True), dtype=np.float32)[0, :-1]
        assert np.array_equal(got[i].cpu().numpy().view(np.uint32), want.view(np.uint32)), (name, i)
