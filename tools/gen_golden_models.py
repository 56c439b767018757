#!/usr/bin/env python3
"""G5 golden fixtures: tiny OPT / Llama models built from the REFERENCE's own model classes
(`models/opt_quantized/modeling_opt.py`, `models/llama_quantized/modeling_llama.py`) and its own per-layer
quant-config expansion (`quant_config_opt.py:61-97`, `quant_config_llama.py:70-116`), run here on CPU.

Run in the build container only (needs /root/reference).  Writes tests/golden/models.npz + models.json:
the (fp32, un-quantised) weights, the token ids, the quant config as the TOML-level dict the reference
parser takes, the reference's PARSED per-layer config, and the reference's logits + loss.  Data only.

    python tools/gen_golden_models.py
    python tools/gen_golden_models.py --wide     # models_wide.{npz,json}: 2 layers at OPT-1.3B / Llama-7B width; the
                                                 # weights are a seeded recipe in the JSON, not data
    python tools/gen_golden_models.py --wide2 [mixed|opt2048|llama2048]   # models_wide2: W4A4 + mixed per-layer widths at
                                                 # OPT-1.3B width; T = 2048 cases (loss + 64 logit rows + attention rows)
"""
from __future__ import annotations

import importlib
import importlib.util
import json
import logging
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
SRC = Path("/root/reference/src/llm_mixed_q")
OUT = ROOT / "tests" / "golden"


def _stub_third_party():
    """optuna / toml / colorlog are absent from the image and are not on the forward path"""
    if "optuna" not in sys.modules:
        opt = types.ModuleType("optuna")
        opt.Trial = opt.Study = object
        opt.trial = types.ModuleType("optuna.trial")
        opt.trial.FrozenTrial = object
        sys.modules["optuna"], sys.modules["optuna.trial"] = opt, opt.trial
    if "toml" not in sys.modules:
        import tomli
        t = types.ModuleType("toml")
        t.load = lambda p: tomli.loads(Path(p).read_text())
        t.loads = tomli.loads
        sys.modules["toml"] = t
    if "colorlog" not in sys.modules:
        c = types.ModuleType("colorlog")

        class ColoredFormatter(logging.Formatter):
            def __init__(self, fmt=None, *a, **k):
                import re
                super().__init__(re.sub(r"%\((log_color|reset)\)s", "", fmt or "%(message)s"))
        c.ColoredFormatter = ColoredFormatter
        sys.modules["colorlog"] = c


def load_reference_models():
    """llm_mixed_q / llm_mixed_q.models as bare namespace packages (models/__init__.py pulls in a BERT file
    that does not import under transformers 5.x), then the two model sub-packages by their real names."""
    _stub_third_party()
    for name, path in (("llm_mixed_q", SRC), ("llm_mixed_q.models", SRC / "models")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [str(path)]
            sys.modules[name] = m
    opt = importlib.import_module("llm_mixed_q.models.opt_quantized.modeling_opt")
    optc = importlib.import_module("llm_mixed_q.models.opt_quantized.configuration_opt")
    llama = importlib.import_module("llm_mixed_q.models.llama_quantized.modeling_llama")
    llamac = importlib.import_module("llm_mixed_q.models.llama_quantized.configuration_llama")
    return opt, optc, llama, llamac


def bfp_default(x_w, w_w, b_w=None):
    """[default] section in the shipped TOMLs' form (configs/quantization/bfp_6bit.toml:5-16)"""
    b_w = w_w if b_w is None else b_w
    return dict(name="block_fp", bypass=False, is_ptq=True,
                data_in_width=x_w, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
                weight_width=w_w, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
                bias_width=b_w, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])


def mixed_opt_config():
    """per-layer mixed widths the way the search writes them (search.py save_best -> TOML with model_layer_i
    sections): layer 0 overrides some nodes, layer 1 falls back to [default]; exponent_bias None -> 127."""
    d = bfp_default(6, 6)
    for k in ("data_in_exponent_bias", "weight_exponent_bias", "bias_exponent_bias"):
        d[k] = "NA"                                  # the TOML spelling of None (utils/config_load.py)
    def node(xw, ww):
        n = dict(d)
        n.update(data_in_width=xw, weight_width=ww, bias_width=ww)
        return n
    return {"default": d,
            "model_layer_0": {"self_attn": {"q_proj": node(4, 5), "k_proj": node(5, 3), "v_proj": node(6, 4),
                                            "out_proj": node(3, 4), "bmm_0": node(5, 4), "bmm_1": node(4, 6)},
                              "fc1": node(4, 4), "fc2": node(5, 2)}}


def mixed_llama_config():
    d = bfp_default(6, 6)
    def node(xw, ww):
        n = dict(d)
        n.update(data_in_width=xw, weight_width=ww, bias_width=ww)
        return n
    return {"default": d,
            "model_layer_1": {"self_attn": {"q_proj": node(4, 5), "k_proj": node(5, 3), "v_proj": node(6, 4),
                                            "o_proj": node(3, 4), "rotary_positional_encoding": node(5, 5),
                                            "matmul_0": node(5, 4), "matmul_1": node(4, 6)},
                              "mlp": {"gate_proj": node(4, 4), "up_proj": node(5, 3), "down_proj": node(4, 5)}}}


def _jsonable(o):
    if isinstance(o, dict):
        return {k: _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    return o


def run_opt(opt, optc, tag, qcfg, arrays, meta, hidden=64, ffn=256, layers=2, heads=4, vocab=128, T=24, B=2, seed=0):
    torch.manual_seed(seed)
    cfg = optc.OPTQuantizedConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, ffn_dim=ffn,
                                  max_position_embeddings=64, num_attention_heads=heads, dropout=0.0,
                                  word_embed_proj_dim=hidden, quant_config=json.loads(json.dumps(qcfg)))
    model = opt.OPTQuantizedForCausalLM(cfg).eval()
    # random weights with some spread (HF init is N(0, 0.02) with zero biases: give biases and norms life)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (0.08 if "embed" not in n else 0.5))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ids = torch.randint(3, vocab, (B, T), generator=g)
    with torch.no_grad():
        out = model(input_ids=ids, labels=ids)
    for k, v in sd.items():
        arrays[f"{tag}/w/{k}"] = v.numpy()
    arrays[f"{tag}/input_ids"] = ids.numpy()
    arrays[f"{tag}/logits"] = out.logits.numpy()
    meta[tag] = dict(family="opt", hidden_size=hidden, ffn_dim=ffn, num_layers=layers, num_heads=heads, vocab_size=vocab,
                     max_positions=64, loss=float(out.loss), quant_config=_jsonable(qcfg),
                     parsed_quant_config=_jsonable({k: v for k, v in cfg.quant_config.items()}))
    print(f"{tag}: loss {float(out.loss):.6f}")


def run_llama(llama, llamac, tag, qcfg, arrays, meta, hidden=64, inter=128, layers=2, heads=4, vocab=128, T=24, B=2, seed=0):
    torch.manual_seed(seed)
    cfg = llamac.LlamaQuantizedConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter,
                                      num_hidden_layers=layers, num_attention_heads=heads, max_position_embeddings=64,
                                      quant_config=json.loads(json.dumps(qcfg)))
    model = llama.LlamaQuantizedForCausalLM(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (0.08 if "embed" not in n else 0.5))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ids = torch.randint(3, vocab, (B, T), generator=g)
    with torch.no_grad():
        out = model(input_ids=ids, labels=ids)
    for k, v in sd.items():
        arrays[f"{tag}/w/{k}"] = v.numpy()
    arrays[f"{tag}/input_ids"] = ids.numpy()
    arrays[f"{tag}/logits"] = out.logits.numpy()
    meta[tag] = dict(family="llama", hidden_size=hidden, intermediate_size=inter, num_layers=layers, num_heads=heads,
                     vocab_size=vocab, max_positions=64, rms_eps=float(cfg.rms_norm_eps), loss=float(out.loss),
                     quant_config=_jsonable(qcfg),
                     parsed_quant_config=_jsonable({k: v for k, v in cfg.quant_config.items()}))
    print(f"{tag}: loss {float(out.loss):.6f}")


def seeded_weights(recipe, seed):
    """the weights of a `wide` case from its recipe [(name, shape, std, mean)] -- numpy's PCG64 stream (stable across
    versions and machines), so that the fixture holds the recipe instead of 0.4-1.6 GB of weights.  tests/ and
    oracle/np_models.py regenerate them with the same lines (np_models.weights_from_recipe)."""
    rng = np.random.default_rng(seed)
    return {name: (rng.standard_normal(tuple(shape), dtype=np.float32) * np.float32(std) + np.float32(mean))
            for name, shape, std, mean in recipe}


def run_wide(mod, cmod, family, tag, qcfg, arrays, meta, hidden, inner, heads, vocab=512, T=128, B=2, layers=2, seed=0):
    """2 decoder layers at a real model's width (OPT-1.3B: 2048 / 8192 / 32 heads; Llama-7B: 4096 / 11008 / 32 heads):
    the shapes whose Linear layers take the 256 x 256-tile int8 GEMM and whose heads take the one-pass attention kernel"""
    torch.manual_seed(seed)
    if family == "opt":
        cfg = cmod.OPTQuantizedConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, ffn_dim=inner,
                                      max_position_embeddings=T, num_attention_heads=heads, dropout=0.0,
                                      word_embed_proj_dim=hidden, quant_config=json.loads(json.dumps(qcfg)))
        model = mod.OPTQuantizedForCausalLM(cfg).eval()
    else:
        cfg = cmod.LlamaQuantizedConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inner, num_hidden_layers=layers,
                                        num_attention_heads=heads, max_position_embeddings=T,
                                        quant_config=json.loads(json.dumps(qcfg)))
        model = mod.LlamaQuantizedForCausalLM(cfg).eval()
    recipe = []
    for n, p in model.named_parameters():
        if p.ndim == 1:
            recipe.append([n, list(p.shape), 0.05, 1.0 if "norm" in n and n.endswith("weight") else 0.0])
        else:
            recipe.append([n, list(p.shape), 0.5 if "embed" in n else 0.02, 0.0])
    w = seeded_weights(recipe, seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(w[n]))
    # (tied / derived entries of the state dict -- lm_head.weight where tied -- follow from the named parameters)
    sd_extra = [k for k in model.state_dict() if k not in w]
    ids = torch.from_numpy(np.random.default_rng(seed + 1).integers(3, vocab, size=(B, T)))
    # the first decoder layer's attention output (its first 128 channels): everything up to there is free of the activation
    # function, whose last-bit differences between implementations move a W6 rounding here and there at these sizes
    layer0 = (model.model.decoder.layers if family == "opt" else model.model.layers)[0]
    caps = {}
    layer0.self_attn.register_forward_hook(lambda mod, i, o: caps.__setitem__("attn0", (o[0] if isinstance(o, tuple) else o).detach()))
    with torch.no_grad():
        out = model(input_ids=ids, labels=ids)
    arrays[f"{tag}/attn0"] = caps["attn0"].reshape(B, T, hidden)[:, :, :128].numpy().copy()
    arrays[f"{tag}/input_ids"] = ids.numpy()
    arrays[f"{tag}/logits"] = out.logits.numpy()
    meta[tag] = dict(family=family, hidden_size=hidden, num_layers=layers, num_heads=heads, vocab_size=vocab, max_positions=T,
                     loss=float(out.loss), quant_config=_jsonable(qcfg), weight_recipe=recipe, weight_seed=seed,
                     state_dict_extra=sd_extra,
                     parsed_quant_config=_jsonable({k: v for k, v in cfg.quant_config.items()}))
    meta[tag].update(dict(ffn_dim=inner) if family == "opt" else dict(intermediate_size=inner, rms_eps=float(cfg.rms_norm_eps)))
    print(f"{tag}: loss {float(out.loss):.6f}  max|logit| {float(out.logits.abs().max()):.3f}")


def main_wide():
    torch.set_num_threads(8)
    opt, optc, llama, llamac = load_reference_models()
    arrays, meta = {}, {}
    run_wide(opt, optc, "opt", "opt1p3b_width_w6a6", {"default": bfp_default(6, 6)}, arrays, meta, 2048, 8192, 32, seed=200)
    run_wide(llama, llamac, "llama", "llama7b_width_w6a6", {"default": bfp_default(6, 6)}, arrays, meta, 4096, 11008, 32, seed=210)
    np.savez_compressed(OUT / "models_wide.npz", **arrays)
    (OUT / "models_wide.json").write_text(json.dumps(meta, indent=1))
    print(f"wide models: {len(meta)} cases, {sum(a.nbytes for a in arrays.values()) / 1e6:.2f} MB raw")


def mixed_wide_opt_config():
    """BASELINE config 4's kind at OPT-1.3B width: [default] W4A4 (configs/quantization/bfp_4bit.toml) and per-layer overrides
    as a search result writes them (experiments/emnlp/configs/search/opt_1.3b_sst2.toml:24-37: data_in 6 / 5 / 4 / 3, weight and
    bias 5 / 4 / 3 / 2)"""
    d = bfp_default(4, 4)
    def node(xw, ww):
        n = dict(d)
        n.update(data_in_width=xw, weight_width=ww, bias_width=ww)
        return n
    return {"default": d,
            "model_layer_0": {"self_attn": {"q_proj": node(6, 5), "k_proj": node(5, 4), "v_proj": node(4, 3), "out_proj": node(3, 5),
                                            "bmm_0": node(6, 5), "bmm_1": node(5, 4)},
                              "fc1": node(5, 2), "fc2": node(6, 3)}}


def run_wide_sampled(mod, cmod, family, tag, qcfg, arrays, meta, hidden, inner, heads, T, vocab=512, layers=2, seed=0, rows=64):
    """a `wide` case at T = 2048 (B = 1): the loss, `rows` sampled logit rows and the same rows of the first layer's attention
    output (first 128 channels) instead of the full tensors"""
    full_a, full_m = {}, {}
    run_wide(mod, cmod, family, tag, qcfg, full_a, full_m, hidden, inner, heads, vocab=vocab, T=T, B=1, layers=layers, seed=seed)
    pick = np.sort(np.random.default_rng(seed + 2).choice(T, size=rows, replace=False))
    pick[:2] = (0, T - 1)
    pick = np.sort(pick)
    arrays[f"{tag}/rows"] = pick.astype(np.int64)
    arrays[f"{tag}/input_ids"] = full_a[f"{tag}/input_ids"]
    arrays[f"{tag}/logits_rows"] = full_a[f"{tag}/logits"][0, pick].copy()
    arrays[f"{tag}/attn0_rows"] = full_a[f"{tag}/attn0"][0, pick].copy()
    meta[tag] = dict(full_m[tag], sampled_rows=rows)


def main_wide2():
    """models_wide2.{npz,json}: (1) OPT-1.3B width under W4A4 with mixed per-layer widths (2 x 128 tokens, full logits);
    (2), (3) one T = 2048 case per family at OPT-1.3B / Llama-7B width under W6A6 (sampled rows)"""
    torch.set_num_threads(8)
    opt, optc, llama, llamac = load_reference_models()
    arrays, meta = {}, {}
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    if not only or "mixed" in only:
        run_wide(opt, optc, "opt", "opt1p3b_width_w4a4_mixed", mixed_wide_opt_config(), arrays, meta, 2048, 8192, 32, seed=220)
    if not only or "opt2048" in only:
        run_wide_sampled(opt, optc, "opt", "opt1p3b_width_w6a6_t2048", {"default": bfp_default(6, 6)}, arrays, meta, 2048, 8192, 32, T=2048, seed=230)
    if not only or "llama2048" in only:
        run_wide_sampled(llama, llamac, "llama", "llama7b_width_w6a6_t2048", {"default": bfp_default(6, 6)}, arrays, meta, 4096, 11008, 32, T=2048, seed=240)
    out_npz, out_json = OUT / "models_wide2.npz", OUT / "models_wide2.json"
    if only and out_npz.exists():                       # (one case at a time: merge into what is there)
        old = np.load(out_npz)
        arrays = {**{k: old[k] for k in old.files}, **arrays}
        meta = {**json.loads(out_json.read_text()), **meta}
    np.savez_compressed(out_npz, **arrays)
    out_json.write_text(json.dumps(meta, indent=1))
    print(f"wide2 models: {len(meta)} cases, {sum(a.nbytes for a in arrays.values()) / 1e6:.2f} MB raw")


def main():
    if "--wide2" in sys.argv:
        return main_wide2()
    if "--wide" in sys.argv:
        return main_wide()
    torch.set_num_threads(4)
    opt, optc, llama, llamac = load_reference_models()
    arrays, meta = {}, {}
    run_opt(opt, optc, "opt_w6a6", {"default": bfp_default(6, 6)}, arrays, meta, seed=0)
    run_opt(opt, optc, "opt_w4a4", {"default": bfp_default(4, 4)}, arrays, meta, seed=10)
    run_opt(opt, optc, "opt_mixed", mixed_opt_config(), arrays, meta, seed=20)
    # a K % 128 == 0 model so that the reference's outputs check the int8 GEMM path directly
    run_opt(opt, optc, "opt_w6a6_k128", {"default": bfp_default(6, 6)}, arrays, meta, hidden=128, ffn=256, heads=2, seed=30)
    run_llama(llama, llamac, "llama_w6a6", {"default": bfp_default(6, 6)}, arrays, meta, seed=40)
    run_llama(llama, llamac, "llama_w4a4", {"default": bfp_default(4, 4)}, arrays, meta, seed=50)
    run_llama(llama, llamac, "llama_mixed", mixed_llama_config(), arrays, meta, seed=60)
    run_llama(llama, llamac, "llama_w6a6_k128", {"default": bfp_default(6, 6)}, arrays, meta, hidden=128, inter=256, heads=2, seed=70)
    # 48 tokens, head_dim 64: shapes the one-pass attention kernel takes (T % 16 == 0, head_dim % 32 == 0), so that the
    # reference's own logits pin that route too (uniform and mixed per-layer widths)
    run_opt(opt, optc, "opt_w6a6_t48", {"default": bfp_default(6, 6)}, arrays, meta, hidden=128, ffn=256, heads=2, T=48, seed=80)
    run_opt(opt, optc, "opt_mixed_t48", mixed_opt_config(), arrays, meta, hidden=128, ffn=256, heads=2, T=48, seed=90)
    run_llama(llama, llamac, "llama_w6a6_t48", {"default": bfp_default(6, 6)}, arrays, meta, hidden=128, inter=256, heads=2, T=48, seed=100)
    run_llama(llama, llamac, "llama_mixed_t48", mixed_llama_config(), arrays, meta, hidden=128, inter=256, heads=2, T=48, seed=110)
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / "models.npz", **arrays)
    (OUT / "models.json").write_text(json.dumps(meta, indent=1))
    print(f"models: {len(meta)} cases, {sum(a.nbytes for a in arrays.values()) / 1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
