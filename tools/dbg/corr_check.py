"""producer-formed exception corrections (csrc/mi355q_corr.h): the row-scale product with `against=` vs without, vs the oracle's
exact integer contraction; the cases that must fall back (more rows / columns with exception blocks than slots); timing."""
import os, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from mi355q import ops
ops.CORR = True
from oracle import np_oracle as O
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device('cuda:0')
CFG = dict(name="block_fp", is_ptq=True, bypass=False,
           data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
           weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
           bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])


def inputs(M, N, K, seed=0, x_exc=0.0, w_exc=0.0, x_rows=None):
    g = lambda s: torch.Generator().manual_seed(s + 10 * seed)
    x = torch.randn(M, K, generator=g(0)) * torch.exp(torch.randn(M, 1, generator=g(1)))
    w = torch.randn(N, K, generator=g(2)) * 0.02
    b = torch.randn(N, generator=g(3)) * 0.02
    rng = np.random.default_rng(seed)
    def spike(t, frac, rows=None):
        if frac <= 0: return
        tb = t.view(t.shape[0], -1, 16)
        n = int(frac * tb.shape[0] * tb.shape[1])
        r = rng.integers(0, tb.shape[0] if rows is None else rows, n); c = rng.integers(0, tb.shape[1], n)
        f = torch.tensor(2.0 ** rng.choice([-6, -5, 5, 6], n), dtype=torch.float32)
        tb[r, c] *= f[:, None]
    spike(x, x_exc, x_rows); spike(w, w_exc)
    return x, w, b


def pack(w, b, ww=6):
    _, wm, we = ops.block_fp_quantize(w.to(dev), ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    return ops.bfp_align_rows(wm, we, ww - 1, 127), ops.block_fp_quantize(b.to(dev), ww, 8, 127, [16], False)


def check(tag, M, N, K, wx=6, ww=6, **kw):
    x, w, b = inputs(M, N, K, **kw)
    wa, bq = pack(w, b, ww)
    xt = x.to(dev)
    xa = ops.block_fp_quantize_aligned_rows(xt, wx, 8, 127)
    y0 = ops.bfp_gemm_aligned(xa, wa, bq).clone()
    xb = ops.block_fp_quantize_aligned_rows(xt, wx, 8, 127, against=wa)
    bound = xb.corr is not None
    y1 = ops.bfp_gemm_aligned(xb, wa, bq).clone()
    torch.cuda.synchronize()
    nx = ops.row_list_entries(xb.sparse, M)[1].shape[0]; nw = ops.row_list_entries(wa.sparse, N)[1].shape[0]
    plan = ops.corr_plan(wa).cpu().numpy() if bound else np.zeros(4, np.int32)
    cfg = dict(CFG, data_in_width=wx, weight_width=ww, bias_width=ww)
    pick = np.sort(np.random.default_rng(1).choice(M, size=min(M, 40), replace=False))
    ref = O.bfp_linear_int(x.numpy()[pick], w.numpy(), b.numpy(), cfg)
    sc = np.abs(ref).max()
    e0 = np.abs(y0.cpu().numpy()[pick] - ref).max() / sc; e1 = np.abs(y1.cpu().numpy()[pick] - ref).max() / sc
    d = (y0 - y1).abs().max().item() / sc
    print(f"{tag:34s} M{M} N{N} K{K} W{ww}A{wx} bound={bound} x_entries={nx} w_entries={nw} x_ovf={int(xb.sparse[0])},{int(xb.sparse[1])} "
          f"plan_bad={int(plan[0])} slots={int(plan[1])}: err old {e0:.2e} new {e1:.2e} old-new {d:.2e}", flush=True)
    assert e1 <= 1e-5 and e0 <= 1e-5, tag


def check_multi(M, N, K, n=3, **kw):
    x, w, b = inputs(M, N, K, **kw)
    packs = [pack(*inputs(M, N, K, seed=5 + i, w_exc=kw.get("w_exc", 0.0))[1:]) for i in range(n)]
    ws = [p[0] for p in packs]; bs = [p[1] for p in packs]
    xt = x.to(dev)
    xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127)
    y0 = [t.clone() for t in ops.bfp_gemm_aligned_multi(xa, ws, bs)]
    ys = [ops.bfp_gemm_aligned(xa, wq, bq).clone() for wq, bq in zip(ws, bs)]
    xb = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, against=ws)      # (the buffers of xa are gone from here on)
    y1 = ops.bfp_gemm_aligned_multi(xb, ws, bs)
    torch.cuda.synchronize()
    for i in range(n):
        sc = ys[i].abs().max().item()
        print(f"multi[{i}] bound={xb.corr is not None}: grouped-old vs single {((y0[i]-ys[i]).abs().max().item()/sc):.2e}, grouped-new vs single {((y1[i]-ys[i]).abs().max().item()/sc):.2e}", flush=True)
        assert (y1[i] - ys[i]).abs().max().item() <= 1e-5 * sc


def timeit(fn, n=300, warm=300):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


if __name__ == "__main__":
    check("bench operands", 4096, 4096, 4096)
    check("bench W4A4", 4096, 4096, 4096, 4, 4)
    check("more exceptions both", 4096, 4096, 4096, x_exc=2e-4, w_exc=2e-4)
    check("x rows overflow their slots", 1024, 4096, 1024, x_exc=2e-3, x_rows=64)
    check("w rows overflow their slots", 1024, 4096, 1024, w_exc=2e-3)
    check("ragged", 300, 520, 512, x_exc=1e-3, w_exc=1e-3)
    check("ragged 2", 2048 + 17, 4096 - 16, 1024, x_exc=3e-4, w_exc=3e-4)
    check("llama up_proj", 2048, 11008, 4096, x_exc=1e-4, w_exc=1e-4)
    check_multi(2048, 4096, 4096, 2, x_exc=1e-4, w_exc=1e-4)
    check_multi(4096, 2048, 2048, 3, x_exc=1e-4, w_exc=1e-4)
    if "time" in sys.argv:
        x, w, b = inputs(4096, 4096, 4096)
        wa, bq = pack(w, b)
        xt = x.to(dev); y = torch.empty(4096, 4096, device=dev)
        for rep in range(2):
            xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127)
            t_g0 = timeit(lambda: ops.bfp_gemm_aligned(xa, wa, bq, out=y))
            xb = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, against=wa)
            t_g1 = timeit(lambda: ops.bfp_gemm_aligned(xb, wa, bq, out=y))
            t_q0 = timeit(lambda: ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127))
            t_q1 = timeit(lambda: ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, against=wa))
            def s0():
                xa = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127); ops.bfp_gemm_aligned(xa, wa, bq, out=y)
            def s1():
                xb = ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, against=wa); ops.bfp_gemm_aligned(xb, wa, bq, out=y)
            t_s0 = timeit(s0); t_s1 = timeit(s1)
            print(f"us: gemm old {t_g0:.2f} new {t_g1:.2f} | quantiser old {t_q0:.2f} new {t_q1:.2f} | step old {t_s0:.2f} new {t_s1:.2f}", flush=True)
