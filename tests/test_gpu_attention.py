"""The one-pass attention kernel (mi355q_bfp_attention behind the registry's "attention" function) against the oracle's
restatement of the reference's steps -- bmm_0 (matmul.py:146-196), scale, mask + clamp, softmax, bmm_1
(modeling_opt.py:246-312, modeling_llama.py:309-344) -- and against the step-by-step route on the GPU."""
import math
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd"))
sys.path.insert(0, str(ROOT))

pytestmark = pytest.mark.gpu
FMIN = np.finfo(np.float32).min


def _cfg(width, **extra):
    return dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=width, data_in_exponent_width=8,
                data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=width, weight_exponent_width=8,
                weight_exponent_bias=127, weight_block_size=[1, 16], **extra)


def _oracle(q, k, v, c0, c1, mask=None, causal=False, scale_div=None):
    from oracle import np_oracle as O
    w = O.matmul_quantized(q, np.swapaxes(k, -1, -2), c0)
    if scale_div:
        w = (w / np.float32(scale_div)).astype(np.float32)
    tq, tk = w.shape[-2:]
    m = np.zeros((tq, tk), np.float32)
    if causal:
        m = np.triu(np.full((tq, tk), FMIN, np.float32), 1 + tk - tq)
    if mask is not None:
        with np.errstate(over="ignore"):
            m = np.maximum(m + mask, FMIN)
    if causal or mask is not None:
        with np.errstate(over="ignore"):
            w = np.maximum(w + m, FMIN)
    e = np.exp((w - w.max(-1, keepdims=True)).astype(np.float64))
    p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return O.matmul_quantized(p, v, c1)


def _inputs(B, M, T, hd, seed):
    r = np.random.default_rng(seed)
    q = (r.normal(size=(B, M, hd)) * np.exp(r.normal(size=(B, M, 1)) * 0.5) * 0.7).astype(np.float32)
    k = (r.normal(size=(B, T, hd)) * np.exp(r.normal(size=(B, 1, hd)) * 0.5)).astype(np.float32)
    v = r.normal(size=(B, T, hd)).astype(np.float32)
    return q, k, v


def _check(out, ref):
    scale = np.abs(ref).max()
    # (a score or probability one ulp apart may round to the next mantissa step of one of T terms of a row)
    assert np.abs(out - ref).max() <= 1e-3 * scale, (np.abs(out - ref).max(), scale)          # north_star: fp tolerance 1e-3
    assert np.abs(out - ref).mean() <= 3e-5 * scale, (np.abs(out - ref).mean(), scale)


@pytest.mark.parametrize("B,T,hd,width,mode", [
    (6, 320, 64, 6, "causal"), (3, 2048, 128, 6, "causal_scaled"), (4, 1008, 64, 6, "mask"), (2, 640, 64, 4, "both"),
    (3, 512, 32, 6, "plain"), (2, 1024, 96, 5, "causal"), (5, 16, 64, 6, "causal"), (2, 2048, 64, 6, "plain")])
@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_attention_one_pass_vs_oracle(B, T, hd, width, mode, kernel):
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    prev = ops.attention_set_kernel(kernel)        # 1: scores resident in registers, 2: streaming (scores formed twice), 3: resident, 8 key-waves (head_dim <= 64; else as 1)
    try:
        _attention_vs_oracle(B, T, hd, width, mode)
    finally:
        ops.attention_set_kernel(prev)


def _attention_vs_oracle(B, T, hd, width, mode):
    import torch
    import mi355q.quantize as Q
    c0, c1 = _cfg(width), _cfg(width)
    q, k, v = _inputs(B, T, T, hd, seed=T + hd)
    r = np.random.default_rng(T)
    add_mask = (r.normal(size=(T, T)) * 0.5).astype(np.float32)
    add_mask[:, ::7] = FMIN
    add_mask[:, 0] = 0
    kw_np = {"causal": dict(causal=True), "causal_scaled": dict(causal=True, scale_div=math.sqrt(hd)), "mask": dict(mask=add_mask),
             "both": dict(mask=add_mask, causal=True), "plain": {}}[mode]
    ref = _oracle(q, k, v, c0, c1, **kw_np)
    dev = "cuda:0"
    kw = dict(kw_np)
    if "mask" in kw:
        kw["mask"] = torch.from_numpy(add_mask).to(dev)
    qt, kt, vt = (torch.from_numpy(t).to(dev) for t in (q, k, v))
    out = Q.get_quantized_func("attention", c1)(qt, kt, vt, c0, c1, **kw)
    _check(out.cpu().numpy(), ref)
    # the step-by-step route through the registry's own functions
    steps = Q.get_quantized_func("attention", c1)(qt, kt, vt, dict(c0, mi355q_fused_matmul=False), dict(c1, mi355q_fused_matmul=False), **kw)
    _check(steps.cpu().numpy(), ref)
    assert (out - steps).abs().max().item() <= 1e-3 * np.abs(ref).max()


def test_attention_mixed_widths_4d_and_fewer_queries():
    """Llama-style 4-D operands, W4 first product / W6 second, fewer queries than keys under the causal rule (query i sees
    keys 0 .. i + T_k - T_q)"""
    import torch
    import mi355q.quantize as Q
    c0, c1 = _cfg(4), _cfg(6)
    B, H, M, T, hd = 2, 3, 96, 384, 64
    q, k, v = _inputs(B * H, M, T, hd, seed=9)
    ref = _oracle(q, k, v, c0, c1, causal=True, scale_div=8.0).reshape(B, H, M, hd)
    t4 = lambda a, n: torch.from_numpy(a).to("cuda:0").reshape(B, H, n, hd)
    out = Q.get_quantized_func("attention", c1)(t4(q, M), t4(k, T), t4(v, T), c0, c1, causal=True, scale_div=8.0)
    assert out.shape == (B, H, M, hd)
    _check(out.cpu().numpy(), ref)


def test_attention_masks_of_every_broadcastable_shape():
    """a padding mask [1, 1, 1, T_k] is expanded over the rows and takes the kernel; a per-batch mask [B, 1, T_q, T_k] cannot
    (the kernel reads ONE [T_q, T_k] mask) and takes the generic steps: both give the reference's values, neither asserts
    (ADVICE r2); the same for the softmax_matmul function"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    c0, c1 = _cfg(6), _cfg(6)
    B, H, T, hd = 2, 2, 320, 64
    q, k, v = _inputs(B * H, T, T, hd, seed=21)
    r = np.random.default_rng(5)
    pad = np.zeros((1, 1, 1, T), np.float32)
    pad[..., -37:] = FMIN
    per_batch = (r.normal(size=(B, 1, T, T)) * 0.5).astype(np.float32)
    t4 = lambda a: torch.from_numpy(a).to("cuda:0").reshape(B, H, -1, hd)
    calls, real = [], ops.bfp_attention
    ops.bfp_attention = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    try:
        f = Q.get_quantized_func("attention", c1)
        out = f(t4(q), t4(k), t4(v), c0, c1, mask=torch.from_numpy(pad).to("cuda:0"), causal=True)
        assert len(calls) == 1
        _check(out.cpu().numpy().reshape(B * H, T, hd), _oracle(q, k, v, c0, c1, mask=np.broadcast_to(pad[0, 0], (T, T)), causal=True))
        out = f(t4(q), t4(k), t4(v), c0, c1, mask=torch.from_numpy(per_batch).to("cuda:0"))
        assert len(calls) == 1                                              # (the generic steps)
        ref = np.stack([_oracle(q[b * H:(b + 1) * H], k[b * H:(b + 1) * H], v[b * H:(b + 1) * H], c0, c1, mask=per_batch[b, 0])
                        for b in range(B)]).reshape(B * H, T, hd)
        _check(out.cpu().numpy().reshape(B * H, T, hd), ref)
    finally:
        ops.bfp_attention = real
    # softmax_matmul with the padding mask against the same with the mask spelled out
    scores = torch.randn(B, H, T, T, device="cuda:0")
    g = Q.get_quantized_func("softmax_matmul", c1)
    a = g(scores, t4(v), dict(c1), mask=torch.from_numpy(pad).to("cuda:0"))
    b = g(scores, t4(v), dict(c1), mask=torch.from_numpy(np.ascontiguousarray(np.broadcast_to(pad[0, 0], (T, T)))).to("cuda:0"))
    assert torch.equal(a, b)


def test_attention_falls_back_outside_the_kernel_shapes():
    """T % 16 != 0: the same steps through bmm / softmax_bmm, same answer; T > 2048: the streaming kernel"""
    import torch
    import mi355q.quantize as Q
    c0, c1 = _cfg(6), _cfg(6)
    for B, T, hd in ((3, 100, 64), (1, 2304, 64)):
        q, k, v = _inputs(B, T, T, hd, seed=T)
        ref = _oracle(q, k, v, c0, c1, causal=True)
        qt, kt, vt = (torch.from_numpy(t).to("cuda:0") for t in (q, k, v))
        out = Q.get_quantized_func("attention", c1)(qt, kt, vt, c0, c1, causal=True)
        _check(out.cpu().numpy(), ref)


def test_attention_model_parity():
    """tiny OPT / Llama through the harness with config["mi355q_fused_attention"]: logits against the step-by-step model"""
    import torch
    from mi355q.harness import (TinyLlamaConfig, TinyLlamaForCausalLM, TinyOPTConfig, TinyOPTForCausalLM,
                                expand_llama_quant_config, expand_quant_config)
    base = dict(_cfg(6), bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
    for family in ("opt", "llama"):
        outs = []
        for fused in (False, True):
            torch.manual_seed(1)
            qc = dict(base, mi355q_fused_attention=fused)
            if family == "opt":
                cfg = TinyOPTConfig(vocab_size=512, hidden_size=256, ffn_dim=1024, num_layers=2, num_heads=4, max_positions=512)
                m = TinyOPTForCausalLM(cfg, expand_quant_config(qc, cfg.num_layers))
            else:
                cfg = TinyLlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, max_positions=512)
                m = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(qc, cfg.num_layers))
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if p.ndim == 2 and "embed" not in n:
                        p.mul_(3.0)
            m = m.to("cuda:0").eval()
            ids = torch.randint(0, cfg.vocab_size, (1, 384), generator=torch.Generator().manual_seed(2)).to("cuda:0")
            with torch.no_grad():
                outs.append(m(ids, labels=ids))
        (l0, s0), (l1, s1) = outs
        assert abs(float(s0) - float(s1)) <= 2e-3, (family, float(s0), float(s1))
        assert (l0 - l1).abs().max().item() <= 2e-2 * l0.abs().max().item()


def test_attention_reads_strided_head_views_in_place():
    """[B, heads, T, hd] views of [B, T, heads, hd] projections (what the models hand over): same result as contiguous copies,
    for B == 1 (strides fold, no copy) and B == 2 (a copy inside)"""
    import torch
    import mi355q.quantize as Q
    c0, c1 = _cfg(6), _cfg(6)
    f = Q.get_quantized_func("attention", c1)
    for B in (1, 2):
        g = torch.Generator().manual_seed(B)
        base = [torch.randn(B, 160, 6, 64, generator=g).to("cuda:0") for _ in range(3)]
        views = [t.transpose(1, 2) for t in base]
        assert not views[0].is_contiguous()
        got = f(*views, c0, c1, causal=True)
        want = f(*(t.contiguous() for t in views), c0, c1, causal=True)
        assert got.shape == (B, 6, 160, 64) and got.is_contiguous() and torch.equal(got, want)


@pytest.mark.parametrize("H,T,D,kernel", [(12, 512, 64, 0), (4, 1536, 64, 3), (3, 640, 128, 1), (2, 2304, 64, 2)])
def test_token_major_output_is_the_same_tensor_without_the_transpose_copy(H, T, D, kernel):
    """config_pv["mi355q_token_major_output"]: the kernel stores out where the out-projection reads it ([1, T, H, D]); the
    [1, H, T, D] view it returns holds the same bits as the contiguous result, and `transpose(1, 2).reshape(1, T, H * D)` is a
    view of it (no copy)"""
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    dev = "cuda:0"
    torch.manual_seed(H + T)
    proj = [torch.randn(1, T, H * D, device=dev) for _ in range(3)]
    q, k, v = (p.view(1, T, H, D).transpose(1, 2) for p in proj)
    cfg = dict(name="block_fp", bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=None,
               data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=None,
               weight_block_size=[1, 16])
    f = Q.get_quantized_func("attention", cfg)
    prev = ops.attention_set_kernel(kernel)
    try:
        want = f(q, k, v, cfg, dict(cfg), causal=True, scale_div=float(D) ** 0.5)
        got = f(q, k, v, cfg, dict(cfg, mi355q_token_major_output=True), causal=True, scale_div=float(D) ** 0.5)
    finally:
        ops.attention_set_kernel(prev)
    assert want.is_contiguous() and got.shape == want.shape and torch.equal(got, want)
    flat = got.transpose(1, 2).reshape(1, T, H * D)
    assert flat.data_ptr() == got.data_ptr() and flat.is_contiguous()
    # batches of several sequences keep the contiguous layout (one stride cannot span batch and head)
    q2, k2, v2 = (torch.randn(2, H, 64, D, device=dev) for _ in range(3))
    got2 = f(q2, k2, v2, cfg, dict(cfg, mi355q_token_major_output=True), causal=True)
    assert got2.is_contiguous() and torch.equal(got2, f(q2, k2, v2, cfg, dict(cfg), causal=True))


@pytest.mark.parametrize("B,T,hd,style", [(32, 2048, 128, "llama"), (12, 2048, 64, "opt"), (32, 4096, 128, "llama")])
def test_attention_error_at_model_shapes(B, T, hd, style):
    """the model shapes the attention bench lines are quoted on (Llama-7B: 32 heads x 128, OPT-125m: 12 x 64; 2048 and 4096
    tokens), all heads through the kernel the library picks, three of them against the oracle: the error a user sees, within
    north_star's 1e-3 (relative to the largest output), and recorded (gpurun_out/r3/attention_error_*.json ->
    profiles/r03_attention_error.json)"""
    import json
    import os
    import torch
    import mi355q.quantize as Q
    c0, c1 = _cfg(6), _cfg(6)
    q, k, v = _inputs(B, T, T, hd, seed=B + T + hd)
    kw = dict(causal=True, scale_div=math.sqrt(hd)) if style == "llama" else dict(causal=True)
    if style == "opt":
        q = (q * np.float32(hd ** -0.5)).astype(np.float32)
    qt, kt, vt = (torch.from_numpy(t).to("cuda:0") for t in (q, k, v))
    out = Q.get_quantized_func("attention", c1)(qt, kt, vt, c0, c1, **kw).cpu().numpy()
    rec = {"shape": [B, T, hd], "style": style, "width": 6, "heads_checked": [], "max_rel": 0.0, "mean_rel": 0.0}
    for h in (0, B // 2, B - 1):
        ref = _oracle(q[h:h + 1], k[h:h + 1], v[h:h + 1], c0, c1, **kw)[0]
        scale = float(np.abs(ref).max())
        d = np.abs(out[h] - ref)
        rec["heads_checked"].append({"head": h, "max_abs": float(d.max()), "mean_abs": float(d.mean()), "max_ref": scale,
                                     "max_rel": float(d.max() / scale), "mean_rel": float(d.mean() / scale),
                                     "words_equal": float((out[h].view(np.uint32) == ref.view(np.uint32)).mean())})
        rec["max_rel"] = max(rec["max_rel"], float(d.max() / scale))
        rec["mean_rel"] = max(rec["mean_rel"], float(d.mean() / scale))
    print(json.dumps(rec))
    try:
        os.makedirs("gpurun_out/r3", exist_ok=True)
        with open(f"gpurun_out/r3/attention_error_{B}x{T}x{hd}.json", "w") as f:
            json.dump(rec, f)
    except OSError:
        pass
    assert rec["max_rel"] <= 1e-3, rec
    assert rec["mean_rel"] <= 2e-5, rec


# ---- the rotary embedding applied on load (round 6: mi355q_bfp_attention_rope) -------------------------------------------------
@pytest.mark.parametrize("Bt,H,T,D,kernel", [(1, 4, 256, 128, 0), (2, 3, 512, 64, 0), (1, 2, 1536, 64, 3), (1, 2, 2048, 128, 1),
                                              (1, 2, 2304, 128, 2), (1, 3, 320, 64, 2), (1, 32, 2048, 128, 0)])
@pytest.mark.parametrize("strided", [False, True])
def test_rotary_on_load_is_rope_apply_then_attention_bit_for_bit(Bt, H, T, D, kernel, strided):
    """q, k straight from the projections + (quantised tables, position_ids) into the attention pass == mi355q_rope_apply first
    (modeling_llama.py:289-299) and the same pass on the turned q / k: the same fp32 arithmetic in another place, so the same bits --
    resident (4 / 8 key-waves) and streaming kernels, [T, heads, D] projections read through strided head views, positions that leave
    the table (clamped) and that are not the row index."""
    import torch
    from mi355q import ops
    if strided and H * T * D * Bt > 2 ** 23:
        pytest.skip("one layout at the largest shape")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Bt * 1000 + H * 100 + T + D + kernel)
    mk = lambda: torch.randn(Bt, T, H, D, generator=g).to(dev).transpose(1, 2) if strided else torch.randn(Bt, H, T, D, generator=g).to(dev)
    q, k, v = mk() * 1.5, mk(), mk()
    rows = T + 8
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    emb = torch.cat([torch.outer(torch.arange(rows).float(), inv)] * 2, dim=-1)
    # (tables quantised as the reference's integer rotary quantiser leaves them: multiples of 2^-7)
    cos_q = (torch.round(emb.cos() * 128) / 128).to(dev).contiguous()
    sin_q = (torch.round(emb.sin() * 128) / 128).to(dev).contiguous()
    pos = torch.stack([torch.randperm(rows, generator=g)[:T] for _ in range(Bt)]).to(dev)
    pos[:, 0] = -3; pos[:, 1] = rows + 5                                      # clamped into the table, as mi355q_rope_apply does
    par = (6, 8, 127, 6, 8, 127)
    prev = ops.attention_set_kernel(kernel)
    try:
        qr, kr = ops.rope_apply(q, k, cos_q, sin_q, pos)
        want = ops.bfp_attention(qr, kr, v, par, par, causal=True, scale_div=math.sqrt(D))
        got = ops.bfp_attention(q, k, v, par, par, causal=True, scale_div=math.sqrt(D), rope=(cos_q, sin_q, pos))
        assert torch.equal(got, want)
        # ... and under an additive mask, token-major output
        m = (torch.randn(T, T, generator=g) * 2).to(dev)
        want = ops.bfp_attention(qr, kr, v, par, par, mask=m, token_major=True)
        got = ops.bfp_attention(q, k, v, par, par, mask=m, token_major=True, rope=(cos_q, sin_q, pos))
        assert torch.equal(got, want)
    finally:
        ops.attention_set_kernel(prev)


def test_rotary_on_load_through_the_registry_and_its_declines():
    """attention_block_fp(rope=...) == the registry's rotary function, then attention_block_fp -- on the HIP pass where it applies
    (head_dim 64 / 128) and through the rotary function first where it does not (head_dim 32, 96; more keys than queries)."""
    import torch
    from mi355q.quantize import get_quantized_func
    dev = torch.device("cuda:0")
    c = _cfg(6, mi355q_fused_attention=True)
    rc = dict(name="integer", bypass=False, data_in_width=8, data_in_frac_width=7)
    att, rope = get_quantized_func("attention", c), get_quantized_func("rotary_positional_encoding", rc)
    for H, T, D in ((4, 128, 128), (2, 64, 64), (2, 64, 32), (2, 64, 96)):
        g = torch.Generator().manual_seed(H + T + D)
        q, k, v = (torch.randn(1, H, T, D, generator=g).to(dev) for _ in range(3))
        inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
        emb = torch.cat([torch.outer(torch.arange(T).float(), inv)] * 2, dim=-1).to(dev)
        cos, sin = emb.cos()[None, None], emb.sin()[None, None]
        pos = torch.arange(T, device=dev)[None]
        qr, kr = rope(q, k, cos, sin, pos, config=rc)
        want = att(qr, kr, v, c, c, causal=True, scale_div=math.sqrt(D))
        got = att(q, k, v, c, c, causal=True, scale_div=math.sqrt(D), rope=(cos, sin, pos, rc))
        assert torch.equal(got, want), (H, T, D)


def test_llama_layer_with_the_rotary_knob_is_the_same_model():
    import torch
    from mi355q import harness
    dev = torch.device("cuda:0")
    q = _cfg(6, bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16], mi355q_fused_attention=True,
             mi355q_grouped_linear=True, mi355q_token_major_output=True)
    cfg = harness.TinyLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=4, max_positions=256)
    outs = []
    for knob in (False, True):
        torch.manual_seed(0)
        qc = {"default": dict(q, mi355q_fused_rotary=knob),
              "rotary_positional_encoding": dict(name="integer", bypass=False, data_in_width=8, data_in_frac_width=7)}
        model = harness.TinyLlamaForCausalLM(cfg, harness.expand_llama_quant_config(qc, 2)).to(dev).eval()
        ids = torch.randint(0, 512, (2, 256), generator=torch.Generator().manual_seed(1)).to(dev)
        with torch.no_grad():
            model(ids)
            outs.append(model(ids, labels=ids))
    assert torch.equal(outs[0][0], outs[1][0]) and float(outs[0][1]) == float(outs[1][1])


@pytest.mark.parametrize("H,M,T,D,kernel,mode", [(4, 256, 256, 128, 0, "causal"), (3, 512, 512, 64, 0, "mask"), (2, 1536, 1536, 64, 3, "causal"),
                                                 (2, 2048, 2048, 128, 1, "causal"), (2, 2304, 2304, 128, 2, "causal"), (3, 100, 320, 64, 2, "mask"),
                                                 (2, 72, 256, 128, 1, "causal"), (12, 2048, 2048, 64, 3, "both"), (12, 2048, 2048, 64, 1, "both"), (12, 2048, 2048, 64, 0, "causal"), (4, 1536, 1536, 64, 0, "mask")])
def test_packed_q_fragments_equal_the_in_kernel_quantiser(H, M, T, D, kernel, mode):
    """round 6: the pack launch leaves the quantised Q fragments and the attention kernels load them (ops.attention_set_qpack(2): wherever
    they fit; 1, the default, where it pays) == every key-wave quantising q itself (0): the same block exponents, the same mantissas, so the same output bits --
    ragged query counts (zero rows in the last fragment tile), fewer queries than keys, all three kernels."""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H + M + T + D)
    q = (torch.randn(H, M, D, generator=g) * torch.exp(torch.randn(H, M, 1, generator=g))).to(dev)
    q[0, 0] = 0.0                                                     # an all-zero row: zero blocks
    q[0, 1, :16] = 2.0 ** -130                                        # a subnormal block maximum
    k, v = torch.randn(H, T, D, generator=g).to(dev), torch.randn(H, T, D, generator=g).to(dev)
    m = (torch.randn(M, T, generator=g) * 2).to(dev) if mode in ("mask", "both") else None
    par = (6, 8, 127, 6, 8, 127)
    prev_k = ops.attention_set_kernel(kernel)
    outs = []
    try:
        for on in (0, 2):
            prev = ops.attention_set_qpack(on)
            try:
                outs.append(ops.bfp_attention(q, k, v, par, par, mask=m, causal=mode in ("causal", "both"), scale_div=math.sqrt(D)))
            finally:
                ops.attention_set_qpack(prev)
    finally:
        ops.attention_set_kernel(prev_k)
    assert torch.equal(outs[0], outs[1])


# ---- the out-projection's operand as the attention output (round 6: mi355q_bfp_attention_fused) ---------------------------------
def _valid_rows(buf, rows, cols):
    """the bytes of rows 0 .. rows - 1 of a tiled bf16 operand [rows, cols] (pieces of 16 rows x 32 values, [8-value group][row][16 B])"""
    v = buf.view(-1, cols // 32, 4, 16, 16)[: (rows + 15) // 16].cpu().numpy().copy()
    if rows % 16:
        v[-1, :, :, rows % 16:] = 0
    return v


@pytest.mark.parametrize("H,M,T,D,kernel,cw,rope", [(4, 256, 256, 128, 0, 6, True), (8, 512, 512, 64, 0, 6, False), (2, 1536, 1536, 64, 3, 5, True),
                                                    (2, 2048, 2048, 128, 1, 8, False), (2, 2304, 2304, 128, 2, 6, True), (3, 100, 320, 64, 2, 4, False),
                                                    (2, 72, 256, 128, 1, 6, False), (32, 2048, 2048, 128, 0, 6, True)])
def test_consumer_operand_from_the_store_epilogue_is_the_separate_quantiser(H, M, T, D, kernel, cw, rope):
    """ops.bfp_attention(consumer=(width, 8, 127)) -- the out-projection's tiled bf16 operand written by the kernels' store epilogue --
    == the fp32 attention output [M, H D] (token-major) through mi355q_block_fp_quantize_bf16_tiled, byte for byte over the valid rows:
    all three kernels, with and without the rotary embedding / packed Q fragments, ragged query counts, consumer widths 4 .. 8."""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * 7 + M + T + D + cw)
    q = (torch.randn(1, M, H, D, generator=g) * 1.5).to(dev).transpose(1, 2)
    k, v = (torch.randn(1, T, H, D, generator=g).to(dev).transpose(1, 2) for _ in range(2))
    v = v * torch.exp(torch.randn(1, H, 1, D, generator=g)).to(dev)         # per-channel spread: block exponents differ along a row
    par = (6, 8, 127, 6, 8, 127)
    rp = None
    if rope:
        assert M == T
        inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
        emb = torch.cat([torch.outer(torch.arange(T).float(), inv)] * 2, dim=-1)
        rp = ((torch.round(emb.cos() * 128) / 128).to(dev).contiguous(), (torch.round(emb.sin() * 128) / 128).to(dev).contiguous(),
              torch.arange(T, device=dev)[None].contiguous())
    prev = ops.attention_set_kernel(kernel)
    try:
        o = ops.bfp_attention(q, k, v, par, par, causal=True, scale_div=math.sqrt(D), token_major=True, rope=rp)
        want = ops.block_fp_quantize_bf16_tiled(o.transpose(1, 2).reshape(M, H * D).contiguous(), cw, 8, 127)
        got = ops.bfp_attention(q, k, v, par, par, causal=True, scale_div=math.sqrt(D), token_major=True, rope=rp, consumer=(cw, 8, 127))
    finally:
        ops.attention_set_kernel(prev)
    assert isinstance(got, ops.TiledBf16) and (got.rows, got.cols) == (M, H * D) and got.buf.numel() == want.numel()
    assert np.array_equal(_valid_rows(got.buf, M, H * D), _valid_rows(want, M, H * D))


@pytest.mark.parametrize("family", ["llama", "opt"])
def test_model_with_the_attention_output_knob_is_the_same_model(family):
    """harness models, out_proj / o_proj on the per-block route (mi355q_align = "blocks"): mi355q_fused_attention_output on == off,
    logits and loss bit for bit -- with the residual add in the product's stores and without"""
    import torch
    from mi355q import harness, ops
    dev = torch.device("cuda:0")
    base = _cfg(6, bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16], mi355q_fused_attention=True,
                mi355q_grouped_linear=True, mi355q_token_major_output=True, mi355q_align="blocks")
    for fres in (False, True):
        outs = []
        for knob in (False, True):
            torch.manual_seed(0)
            d = dict(base, mi355q_fused_attention_output=knob, mi355q_fused_residual=fres, mi355q_fused_norm=fres, mi355q_fused_activation=fres,
                     mi355q_fused_rotary=knob)
            if family == "llama":
                cfg = harness.TinyLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=4, max_positions=256)
                qc = {"default": d, "rotary_positional_encoding": dict(name="integer", bypass=False, data_in_width=8, data_in_frac_width=7)}
                model = harness.TinyLlamaForCausalLM(cfg, harness.expand_llama_quant_config(qc, 2)).to(dev).eval()
            else:
                cfg = harness.TinyOPTConfig(vocab_size=512, hidden_size=512, ffn_dim=1024, num_layers=2, num_heads=8, max_positions=256)
                model = harness.TinyOPTForCausalLM(cfg, harness.expand_quant_config({"default": d}, 2)).to(dev).eval()
            ids = torch.randint(0, 512, (1, 256), generator=torch.Generator().manual_seed(1)).to(dev)
            with torch.no_grad():
                model(ids)
                outs.append(model(ids, labels=ids))
            proj = model.layers[0].self_attn.o_proj if family == "llama" else model.layers[0].self_attn.out_proj
            assert proj.accepts_tiled_input()
        assert torch.equal(outs[0][0], outs[1][0]) and float(outs[0][1]) == float(outs[1][1]), (family, fres)


@pytest.mark.parametrize("H,M,T,D,kernel,scale", [(12, 512, 512, 64, 0, 0.125), (8, 2048, 2048, 64, 0, 0.125), (4, 1024, 1024, 128, 1, 128 ** -0.5),
                                                  (2, 2304, 2304, 128, 2, 128 ** -0.5), (3, 100, 320, 64, 3, 0.3), (2, 48, 256, 64, 0, 0.125)])
def test_q_scale_in_the_fragment_pack_is_the_torch_multiply(H, M, T, D, kernel, scale):
    """OPT's q_proj(x) * scaling (modeling_opt.py:231) formed where the Q fragments are packed (q_scale) == the torch multiply in front of
    the pass: one fp32 multiply either way -- powers of two and not, fewer than 64 queries (the fragments are packed regardless)"""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H + M + D)
    q = (torch.randn(1, M, H, D, generator=g) * 3).to(dev).transpose(1, 2)
    k, v = (torch.randn(1, T, H, D, generator=g).to(dev).transpose(1, 2) for _ in range(2))
    par = (6, 8, 127, 6, 8, 127)
    prev = ops.attention_set_kernel(kernel)
    try:
        want = ops.bfp_attention(q * scale, k, v, par, par, causal=True, token_major=True)
        got = ops.bfp_attention(q, k, v, par, par, causal=True, token_major=True, q_scale=scale)
    finally:
        ops.attention_set_kernel(prev)
    assert torch.equal(got, want)
