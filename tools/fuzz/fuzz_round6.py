"""Round 6's fused paths against the separate launches they replace, random shapes (bit for bit unless noted):
  attention: rotary embedding / q scaling in the pack launch, packed Q fragments, the out-projection's operand from the store epilogue --
             against rope_apply / the torch multiply, q quantised in the kernels, block_fp_quantize_bf16_tiled of the fp32 output;
  linear:    the residual add in the int8 product's stores against product + torch add; the relu / gated epilogues against the
             separate launches; the split-bf16 fp32 product against a float64 product (<= 2e-6 of the magnitude sum).
    python tools/fuzz/fuzz_round6.py [seed [cases]]"""
import json, math, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np
import torch
from mi355q import ops

dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
g = torch.Generator().manual_seed(seed)
bad = []


def rows_of(buf, rows, cols):
    v = buf.view(-1, cols // 32, 4, 16, 16)[: (rows + 15) // 16].cpu().numpy().copy()
    if rows % 16:
        v[-1, :, :, rows % 16:] = 0
    return v


for case in range(N):
    kind = case % 4
    try:
        if kind in (0, 1):                                   # ---- attention
            D = int(rng.choice([64, 128]))
            H = int(rng.integers(1, 9))
            T = 16 * int(rng.integers(1, 160 if kind == 0 else 60))
            rope = bool(rng.integers(2))
            M = T if rope or rng.integers(2) else int(rng.integers(1, T + 1))
            kernel = int(rng.choice([0, 1, 2, 3])) if T <= 2048 else 2
            if kernel == 3 and D != 64:
                kernel = 1
            causal = bool(rng.integers(2))
            cons = (int(rng.integers(4, 9)), 8, 127) if rng.integers(2) else None
            qs = None if rope or not rng.integers(2) else float(rng.choice([0.125, 0.0883883461356163, 0.3]))
            wq, wp = int(rng.integers(4, 9)), int(rng.integers(4, 9))
            par, parp = (wq, 8, 127, wq, 8, 127), (wp, 8, 127, wp, 8, 127)
            mk = lambda n: (torch.randn(1, n, H, D, generator=g) * float(rng.choice([0.3, 1.0, 3.0]))).to(dev).transpose(1, 2)
            q, k, v = mk(M), mk(T), mk(T)
            tabs = None
            if rope:
                rows = T + int(rng.integers(0, 9))
                inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
                emb = torch.cat([torch.outer(torch.arange(rows).float(), inv)] * 2, dim=-1)
                tabs = ((torch.round(emb.cos() * 128) / 128).to(dev).contiguous(), (torch.round(emb.sin() * 128) / 128).to(dev).contiguous(),
                        torch.from_numpy(rng.integers(-2, rows + 2, size=(1, T))).to(dev).contiguous())
            sd = math.sqrt(D) if rng.integers(2) else None
            prev = ops.attention_set_kernel(kernel)
            try:
                q1, k1 = (ops.rope_apply(q, k, *tabs) if rope else (q, k))
                if qs:
                    q1 = q1 * qs
                pq = ops.attention_set_qpack(0)
                want = ops.bfp_attention(q1, k1, v, par, parp, causal=causal and T >= M, scale_div=sd, token_major=True)
                ops.attention_set_qpack(int(rng.choice([1, 2])))
                got = ops.bfp_attention(q, k, v, par, parp, causal=causal and T >= M, scale_div=sd, token_major=True, rope=tabs, consumer=cons, q_scale=qs)
                ops.attention_set_qpack(pq)
            finally:
                ops.attention_set_kernel(prev)
            if cons is None:
                ok = torch.equal(got, want)
            else:
                wt = ops.block_fp_quantize_bf16_tiled(want.transpose(1, 2).reshape(M, H * D).contiguous(), cons[0], 8, 127)
                ok = np.array_equal(rows_of(got.buf, M, H * D), rows_of(wt, M, H * D))
            desc = dict(kind="attention", H=H, M=M, T=T, D=D, kernel=kernel, rope=rope, q_scale=qs, consumer=cons, causal=causal, widths=(wq, wp))
        elif kind == 2:                                      # ---- int8 product with a residual; relu epilogue
            M = int(rng.integers(1, 700)); K = 128 * int(rng.integers(2, 33)); Nn = 4 * int(rng.integers(8, 700))
            x = (torch.randn(M, K, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).to(dev)
            w = (torch.randn(Nn, K, generator=g) * 0.05).to(dev)
            b = torch.randn(Nn, generator=g).to(dev) if rng.integers(2) else None
            res = torch.randn(M, Nn, generator=g).to(dev)
            wx, ww = int(rng.integers(4, 7)), int(rng.integers(4, 7))
            xa = ops.block_fp_quantize_aligned_rows(x, wx, 8, 127)
            _, wm, we = ops.block_fp_quantize(w, ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
            wa = ops.bfp_align_rows(wm, we, ww - 1, 127)
            want = ops.bfp_gemm_aligned(xa, wa, b) + res
            got = ops.bfp_gemm_aligned(xa, wa, b, residual=res)
            ok = torch.equal(got, want)
            desc = dict(kind="residual", M=M, N=Nn, K=K, widths=(wx, ww), bias=b is not None)
            if ok and Nn % 32 == 0 and K >= 256:
                cw = int(rng.integers(4, 9))
                h = ops.bfp_gemm_aligned_relu(xa, wa, cw, 8, 127, b)
                if h is not None:
                    hr = ops.block_fp_quantize_bf16_tiled(ops.bfp_gemm_aligned(xa, wa, b), cw, 8, 127, pre=("relu", None))
                    ok = np.array_equal(rows_of(h, M, Nn), rows_of(hr, M, Nn))
                    desc["relu_consumer_width"] = cw
        else:                                                # ---- split-bf16 product
            M = int(rng.integers(1, 600)); K = 32 * int(rng.integers(1, 130)); Nn = int(rng.integers(1, 900))
            sp = float(rng.choice([0.0, 0.5, 2.0]))
            x = torch.randn(M, K, generator=g) * torch.exp(sp * torch.randn(M, K, generator=g))
            w = torch.randn(Nn, K, generator=g) * 0.02 * torch.exp(sp * torch.randn(Nn, K, generator=g))
            y = ops.fp32_gemm_split(ops.fp32_split_tile(x.to(dev), 0), ops.fp32_split_tile(w.to(dev), 1), M, Nn, K)
            ref = x.double() @ w.double().t()
            mag = x.double().abs() @ w.double().abs().t() + 1e-300
            err = float(((y.cpu().double() - ref).abs() / mag).max())
            ok = err <= 2e-6 and bool(torch.isfinite(y).all())
            desc = dict(kind="split", M=M, N=Nn, K=K, spread=sp, err=err)
        if not ok:
            bad.append(desc)
            print("MISMATCH", json.dumps(desc, default=str), flush=True)
    except Exception as e:                                   # noqa: BLE001
        bad.append(dict(case=case, error=repr(e)[:300]))
        print("ERROR", case, repr(e)[:300], flush=True)
print(json.dumps({"seed": seed, "cases": N, "failures": len(bad)}))
sys.exit(1 if bad else 0)
