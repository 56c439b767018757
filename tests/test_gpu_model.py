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
