"""The gated epilogue (round 6; include/mi355q.h mi355q_bfp_gemm_aligned_gated, csrc/mi355q_gemm_v9g.hip): x against the INTERLEAVED
gate / up weights of a gated MLP (modeling_llama.py:216) with silu(gate) * up and the consumer's block_fp quantiser in the store
epilogue -- the consumer's tiled bf16 operand is the only output.  Held bit for bit to what the separate launches make: the two
int8 products (mi355q_bfp_gemm_aligned), then the quantiser that reads silu(gate) * up (mi355q_block_fp_quantize_bf16_tiled_pre)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _operands(M, I, K, seed, wx=6, ww=6, bias=False, x_exc=0, w_exc=0, wild_rows=0):
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))
    wg, wu = torch.randn(I, K, generator=g) * 0.05, torch.randn(I, K, generator=g) * 0.05
    r = np.random.default_rng(seed)
    for _ in range(x_exc):                                   # exception blocks: far above / below their rows' window
        row, kb = int(r.integers(M)), int(r.integers(K // 16))
        x[row, kb * 16:kb * 16 + 16] *= float(r.choice([1 / 512.0, 300.0]))
    for _ in range(w_exc):
        row, kb = int(r.integers(I)), int(r.integers(K // 16))
        (wg if r.integers(2) else wu)[row, kb * 16:kb * 16 + 16] *= float(r.choice([1 / 512.0, 300.0]))
    for row in range(wild_rows):                             # every one of the first rows: an exception block (a bucket overflows)
        kb = int(r.integers(K // 16))
        x[row, kb * 16:kb * 16 + 16] *= 2000.0
    bg = bu = None
    if bias:
        bg, bu = (torch.randn(I, generator=g) * 0.05).to(dev), (torch.randn(I, generator=g) * 0.05).to(dev)
    xa = ops.block_fp_quantize_aligned_rows(x.to(dev), wx, 8, 127)
    was = []
    for w in (wg, wu):
        _, wm, we = ops.block_fp_quantize(w.to(dev), ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True)
        was.append(ops.bfp_align_rows(wm, we, ww - 1, 127))
    return xa, was[0], was[1], bg, bu


def _both(M, I, K, seed, qw=6, **kw):
    import torch
    from mi355q import ops
    xa, wa_g, wa_u, bg, bu = _operands(M, I, K, seed, **kw)
    w_gu = ops.interleave_gate_up(wa_g, wa_u)
    assert w_gu is not None
    b_gu = None if bg is None else torch.stack((bg.reshape(I // 16, 16), bu.reshape(I // 16, 16)), dim=1).reshape(-1).contiguous()
    xt = ops.bfp_gemm_aligned_gated(xa, w_gu, qw, 8, 127, b_gu)
    assert xt is not None
    xt = xt.clone()
    gate, up = ops.bfp_gemm_aligned(xa, wa_g, bg), ops.bfp_gemm_aligned(xa, wa_u, bu)
    ref = ops.block_fp_quantize_bf16_tiled(gate, qw, 8, 127, pre=("silu_mul", up), out=torch.zeros_like(xt))
    torch.cuda.synchronize()
    return xt, ref, xa, w_gu


@pytest.mark.parametrize("M,I,K,bias,qw", [(512, 256, 512, False, 6), (300, 384, 1024, True, 6), (1024, 1408, 2048, False, 4),
                                            (256, 128, 256, True, 8), (2048, 2816, 1024, False, 6)])
def test_gated_epilogue_equals_the_separate_launches(M, I, K, bias, qw):
    import torch
    xt, ref, xa, w_gu = _both(M, I, K, seed=M + I + K, qw=qw, bias=bias, x_exc=12, w_exc=20)
    assert int(xa.sparse[0]) == 0 and int(w_gu.sparse[0]) == 0
    assert torch.equal(xt, ref)


def test_gated_epilogue_slow_paths():
    """an overflowed activation bucket (the blockwise-exact product into the scratch, converted tile by tile) and a tile with more
    exception entries than its LDS holds (atomics behind the stores, then the conversion): the same operand as the separate launches
    -- whose own slow paths add their exception terms in another order: equal as values up to that, i.e. a quantised value may
    differ where the fp32 sums differ in the last bit; held to the quantiser's resolution on all but a handful of values"""
    import torch
    for kw in (dict(wild_rows=200), dict(x_exc=150, w_exc=150)):
        xt, ref, xa, w_gu = _both(512, 256, 512, seed=7, **kw)
        if "wild_rows" in kw:
            assert int(xa.sparse[0]) != 0
        a = xt.view(torch.bfloat16).float()
        b = ref.view(torch.bfloat16).float()
        differ = (a != b).float().mean().item()
        assert differ < 2e-3, differ
        assert (a - b).abs().max().item() <= 0.07 * b.abs().max().item()


def _llama_layer(knobs):
    import torch
    from mi355q import harness as H
    d = dict(name="block_fp", bypass=False, is_ptq=True, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
             data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
             bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16], **knobs)
    cfg = {"default": d, "rotary_positional_encoding": dict(name="integer", bypass=False, data_in_width=8, data_in_frac_width=7)}
    c = H.TinyLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_layers=2, num_heads=8, max_positions=256)
    torch.manual_seed(3)
    return H.TinyLlamaForCausalLM(c, H.expand_llama_quant_config(cfg, 2)).to("cuda:0").eval()


def test_llama_harness_with_the_gated_mlp_is_bit_identical():
    """the Llama-style harness, every knob on: with the gate / up product writing down_proj's operand itself (gated_mlp) and
    with the grouped launch + the quantiser that reads silu(gate) * up -- the same logits, bit for bit; and the fused path is
    actually taken"""
    import torch
    from mi355q import ops
    knobs = dict(mi355q_fused_attention=True, mi355q_grouped_linear=True, mi355q_fused_activation=True, mi355q_fused_norm=True,
                 mi355q_token_major_output=True, mi355q_fused_residual=True)
    ids = torch.randint(0, 512, (1, 256), generator=torch.Generator().manual_seed(5)).to("cuda:0")
    calls, real = [], ops.bfp_gemm_aligned_gated
    ops.bfp_gemm_aligned_gated = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            m1 = _llama_layer(knobs)
            for _ in range(2):
                l1 = m1(ids)[0]
            n_fused = len(calls)
            m2 = _llama_layer(dict(knobs, mi355q_fused_gate_up=False))
            for _ in range(2):
                l2 = m2(ids)[0]
    finally:
        ops.bfp_gemm_aligned_gated = real
    routes = [(l.gate_proj._uses_bf16_route(), l.down_proj._uses_bf16_route()) for l in m1.layers]
    assert len(calls) == n_fused, "mi355q_fused_gate_up = False must not take the fused path"
    if all(not g and d for g, d in routes):
        assert n_fused == 2, (n_fused, routes)          # (second forward: both layers; the first one packs the weights)
    assert torch.equal(l1, l2)


@pytest.mark.parametrize("M,N,K,bias,qw", [(512, 512, 512, True, 6), (300, 384, 1024, True, 6), (1024, 1408, 2048, False, 4), (256, 96, 256, True, 8)])
def test_relu_epilogue_equals_the_separate_launches(M, N, K, bias, qw):
    """mi355q_bfp_gemm_aligned_relu (OPT's fc1 in front of fc2, modeling_opt.py:412-420): the consumer's tiled bf16 operand from the
    product's store epilogue, bit for bit the int8 product followed by the quantiser that reads relu(.)"""
    import torch
    from mi355q import ops
    xa, wa, _, bg, _ = _operands(M, N, K, seed=M + N + K + 1, bias=bias, x_exc=10, w_exc=16)
    xt = ops.bfp_gemm_aligned_relu(xa, wa, qw, 8, 127, bg)
    assert xt is not None
    xt = xt.clone()
    y = ops.bfp_gemm_aligned(xa, wa, bg)
    ref = ops.block_fp_quantize_bf16_tiled(y, qw, 8, 127, pre=("relu", None), out=torch.zeros_like(xt))
    torch.cuda.synchronize()
    assert torch.equal(xt, ref)


def test_opt_harness_with_the_relu_mlp_is_bit_identical():
    """the OPT-style harness, every knob on, W6A6 (fc2 behind a relu: per-block route): fc1's product writing fc2's operand itself
    (relu_mlp) against the fc1 launch + the quantiser that reads relu(.) -- the same logits, bit for bit; the fused path is taken"""
    import torch
    from mi355q import harness as H, ops
    knobs = dict(mi355q_fused_attention=True, mi355q_grouped_linear=True, mi355q_fused_activation=True, mi355q_fused_norm=True,
                 mi355q_token_major_output=True, mi355q_fused_residual=True)

    def build(extra):
        d = dict(name="block_fp", bypass=False, is_ptq=True, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
                 data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
                 bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16], **knobs, **extra)
        c = H.TinyOPTConfig(vocab_size=512, hidden_size=512, ffn_dim=2048, num_layers=2, num_heads=8, max_positions=256)
        torch.manual_seed(4)
        return H.TinyOPTForCausalLM(c, H.expand_quant_config(d, 2)).to("cuda:0").eval()
    ids = torch.randint(0, 512, (1, 256), generator=torch.Generator().manual_seed(6)).to("cuda:0")
    calls, real = [], ops.bfp_gemm_aligned_relu
    ops.bfp_gemm_aligned_relu = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            m1 = build({})
            for _ in range(2):
                l1 = m1(ids)[0]
            n_fused = len(calls)
            m2 = build(dict(mi355q_fused_gate_up=False))
            for _ in range(2):
                l2 = m2(ids)[0]
    finally:
        ops.bfp_gemm_aligned_relu = real
    assert len(calls) == n_fused
    if all(not l.fc1._uses_bf16_route() and l.fc2._uses_bf16_route() for l in m1.layers):
        assert n_fused == 2, n_fused
    assert torch.equal(l1, l2)
