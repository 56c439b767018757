"""Model-level parity (BASELINE configs 0/2/3 in the only form available here: model-shaped random
weights, seeded token ids): a small OPT-style decoder driven through the registry API on the GPU
(HIP quantisers, int8-MFMA Linear) against the same network evaluated on the CPU with the oracle's
quantisers.  Losses must agree so that exp(loss) matches to 3 d.p."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_forward(model, cfg_default, ids):
    """numpy/float64 evaluation of TinyOPTForCausalLM with oracle quantisers (PTQ semantics)"""
    import torch
    from oracle import np_oracle as O
    sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in model.state_dict().items()}
    c = model.cfg
    B, T = ids.shape
    nh, hd = c.num_heads, c.hidden_size // c.num_heads

    def ln(x, p):
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return ((x - mu) / np.sqrt(var + 1e-5) * sd[p + ".weight"] + sd[p + ".bias"]).astype(np.float32)

    def lin(x, p):
        y, _, _ = O.linear_ptq(x, sd[p + ".weight"], sd[p + ".bias"], cfg_default)
        return y

    x = sd["embed_tokens.weight"][ids] + sd["embed_positions.weight"][np.arange(T)][None]
    mask = np.triu(np.full((T, T), np.finfo(np.float32).min, np.float32), 1)[None, None]
    for i in range(c.num_layers):
        pre = f"layers.{i}."
        h = ln(x, pre + "self_attn_layer_norm")
        shape = lambda t: t.reshape(B, T, nh, hd).transpose(0, 2, 1, 3).reshape(B * nh, T, hd)
        q = shape(lin(h, pre + "self_attn.q_proj") * np.float32(hd ** -0.5))
        k, v = shape(lin(h, pre + "self_attn.k_proj")), shape(lin(h, pre + "self_attn.v_proj"))
        w = O.matmul_quantized(q, np.ascontiguousarray(k.transpose(0, 2, 1)), cfg_default)
        w = np.maximum(w.reshape(B, nh, T, T) + mask, np.finfo(np.float32).min).reshape(B * nh, T, T)
        w = w - w.max(-1, keepdims=True)
        p = (np.exp(w) / np.exp(w).sum(-1, keepdims=True)).astype(np.float32)
        o = O.matmul_quantized(p, v, cfg_default).reshape(B, nh, T, hd).transpose(0, 2, 1, 3).reshape(B, T, c.hidden_size)
        x = x + lin(o, pre + "self_attn.out_proj")
        h2 = x.reshape(-1, c.hidden_size)
        f = lin(np.maximum(lin(ln(h2, pre + "final_layer_norm"), pre + "fc1"), 0), pre + "fc2")
        x = (h2 + f).reshape(B, T, c.hidden_size)
    logits = ln(x, "final_layer_norm").astype(np.float64) @ sd["lm_head.weight"].astype(np.float64).T
    lg = logits[:, :-1].reshape(-1, c.vocab_size)
    tgt = ids[:, 1:].reshape(-1)
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    return float((lse - lg[np.arange(tgt.size), tgt]).mean())


@pytest.mark.parametrize("toml_default", [
    dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
         data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
         weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16]),
    dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=4, data_in_exponent_width=8, data_in_exponent_bias=127,
         data_in_block_size=[1, 16], weight_width=4, weight_exponent_width=8, weight_exponent_bias=127,
         weight_block_size=[1, 16], bias_width=4, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16]),
    dict(name="block_log", is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_bias_width=8,
         data_in_block_size=[1, 16], weight_width=8, weight_exponent_bias_width=8, weight_block_size=[1, 16],
         bias_width=8, bias_exponent_bias_width=8, bias_block_size=[16]),
], ids=["bfp_w6a6", "bfp_w4a4", "block_log_w8"])
def test_tiny_opt_loss_parity(toml_default):
    import torch
    from mi355q.harness import TinyOPTConfig, TinyOPTForCausalLM, eval_lm_perplexity, expand_quant_config
    torch.manual_seed(0)
    cfg = TinyOPTConfig(vocab_size=384, hidden_size=256, ffn_dim=512, num_layers=2, num_heads=4, max_positions=64)
    qc = expand_quant_config(toml_default, cfg.num_layers)
    model = TinyOPTForCausalLM(cfg, qc)
    with torch.no_grad():                      # make the weights less trivial than N(0, 0.02)
        for n, p in model.named_parameters():
            if p.ndim == 2 and "embed" not in n:
                p.mul_(4.0)
            if n.endswith("bias"):
                p.normal_(0, 0.05)
    ids = torch.randint(0, cfg.vocab_size, (2, 48))
    ref_loss = _oracle_forward(model, toml_default, ids.numpy())      # before the PTQ in-place overwrite
    model = model.to("cuda:0")
    res = eval_lm_perplexity(model, [ids], device="cuda:0")
    assert abs(res["loss"] - ref_loss) < 2e-4, (res["loss"], ref_loss)
    assert round(math.exp(res["loss"]), 3) == round(math.exp(ref_loss), 3) or abs(math.exp(res["loss"]) - math.exp(ref_loss)) < 1e-3
    if toml_default["name"] == "block_fp":
        assert model.layers[0].fc1._packed is not None, "int8-MFMA path not taken by the MLP"
    # second pass: steady state (weights already quantised in place) gives the same loss
    res2 = eval_lm_perplexity(model, [ids], device="cuda:0")
    assert abs(res2["loss"] - res["loss"]) < 1e-6


def _oracle_llama_forward(model, cfg_default, ids):
    """numpy evaluation of TinyLlamaForCausalLM with the oracle's quantisers: RMSNorm, rotary embedding with
    quantised cos / sin tables (rotary_positional_encoding.py:59-82), 4-D quantised products (matmul.py:146-196),
    SiLU-gated MLP, all Linear layers bias-free (PTQ semantics)"""
    from oracle import np_oracle as O
    sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in model.state_dict().items()}
    c = model.cfg
    B, T = ids.shape
    nh, hd = c.num_heads, c.hidden_size // c.num_heads
    f32 = np.float32

    def rms(x, p):
        v = (x.astype(np.float32) ** 2).mean(-1, keepdims=True)
        return (sd[p + ".weight"] * (x * (f32(1.0) / np.sqrt(v + f32(c.rms_eps)))).astype(np.float32)).astype(np.float32)

    def lin(x, p):
        return O.linear_ptq(x, sd[p + ".weight"], None, cfg_default)[0]

    def rot_half(t):
        h = t.shape[-1] // 2
        return np.concatenate([-t[..., h:], t[..., :h]], axis=-1)

    # the model's own fp32 cos / sin tables (buffers, not in the state dict): the quantised tables are what is compared
    cos_t = model.layers[0].self_attn.cos.detach().cpu().numpy()[0, 0, :T].astype(np.float32)
    sin_t = model.layers[0].self_attn.sin.detach().cpu().numpy()[0, 0, :T].astype(np.float32)
    kw = {k: cfg_default[f"data_in_{k}"] for k in ("width", "exponent_width", "exponent_bias", "block_size")}
    cos = O.block_fp_quantize(cos_t, **kw, skip_first_dim=False)[None, None]
    sin = O.block_fp_quantize(sin_t, **kw, skip_first_dim=False)[None, None]
    x = sd["embed_tokens.weight"][ids]
    mask = np.triu(np.full((T, T), np.finfo(np.float32).min, np.float32), 1)[None, None]
    for i in range(c.num_layers):
        pre = f"layers.{i}."
        h = rms(x, pre + "input_layernorm")
        shape = lambda t: t.reshape(B, T, nh, hd).transpose(0, 2, 1, 3)
        q, k, v = (shape(lin(h, pre + f"self_attn.{n}_proj")) for n in "qkv")
        q, k = (q * cos + rot_half(q) * sin).astype(np.float32), (k * cos + rot_half(k) * sin).astype(np.float32)
        w = O.matmul_quantized(q, np.ascontiguousarray(k.transpose(0, 1, 3, 2)), cfg_default)
        w = np.maximum((w / f32(math.sqrt(hd))).astype(np.float32) + mask, np.finfo(np.float32).min)
        w = w - w.max(-1, keepdims=True)
        p = (np.exp(w) / np.exp(w).sum(-1, keepdims=True)).astype(np.float32)
        o = O.matmul_quantized(p, np.ascontiguousarray(v), cfg_default).transpose(0, 2, 1, 3).reshape(B, T, c.hidden_size)
        x = x + lin(o, pre + "self_attn.o_proj")
        h2 = rms(x, pre + "post_attention_layernorm")
        g = lin(h2, pre + "gate_proj")
        act = (g / (f32(1.0) + np.exp(-g))).astype(np.float32) * lin(h2, pre + "up_proj")
        x = x + lin(act.astype(np.float32), pre + "down_proj")
    logits = rms(x, "norm").astype(np.float64) @ sd["lm_head.weight"].astype(np.float64).T
    lg = logits[:, :-1].reshape(-1, c.vocab_size)
    tgt = ids[:, 1:].reshape(-1)
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    return float((lse - lg[np.arange(tgt.size), tgt]).mean())


@pytest.mark.parametrize("width", [6, 4])
def test_tiny_llama_loss_parity(width):
    """Llama-style callers of the path: rotary embedding function, 4-D matmul functions (fused quantise + matmul
    where the blocks tile the operands), bias-free Linear layers with a SiLU-gated input (heavy-tailed: exercises the
    exception add-back of the row-aligned GEMM)"""
    import torch
    from mi355q.harness import TinyLlamaConfig, TinyLlamaForCausalLM, eval_lm_perplexity, expand_llama_quant_config
    d = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=width, data_in_exponent_width=8,
             data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=width, weight_exponent_width=8,
             weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=width, bias_exponent_width=8,
             bias_exponent_bias=127, bias_block_size=[16])
    torch.manual_seed(1)
    cfg = TinyLlamaConfig(vocab_size=384, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, max_positions=64)
    model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(d, cfg.num_layers))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 2 and "embed" not in n:
                p.mul_(4.0)
    ids = torch.randint(0, cfg.vocab_size, (2, 48))
    ref_loss = _oracle_llama_forward(model, d, ids.numpy())
    model = model.to("cuda:0")
    res = eval_lm_perplexity(model, [ids], device="cuda:0")
    assert abs(res["loss"] - ref_loss) < 3e-4, (res["loss"], ref_loss)
    assert model.layers[0].down_proj._packed is not None, "int8-MFMA path not taken by the MLP"
    res2 = eval_lm_perplexity(model, [ids], device="cuda:0")
    assert abs(res2["loss"] - res["loss"]) < 1e-6
