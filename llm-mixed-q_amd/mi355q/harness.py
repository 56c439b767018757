"""Minimal OPT-style and Llama-style decoder harnesses over the registry API, and the perplexity loop.

NOT a port of the reference's HF model files (those are callers of the path and out of scope,
SURVEY.md section 2 rows 8-9): this is just enough model to drive the hot path the way
`OPTQuantizedDecoderLayer` does -- q/k/v/out_proj, fc1, fc2 through `get_quantized_cls("linear")`,
the two attention products through `get_quantized_func("bmm")`, 3-D inputs into the attention
projections, 2-D into the MLP, an unquantised lm_head (reference modeling_opt.py:143-330, 333-441,
934-1109) -- so that loss / perplexity parity of the HIP path can be checked on model-shaped random
weights (no checkpoint or Wikitext2 copy exists in this environment).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .quantize import fp32_linear, gated_mlp, get_quantized_cls, get_quantized_func, grouped_linear, relu_mlp
from .quantize.model_quant_config import parse_llama_quantized_config, parse_opt_quantized_config


def _attention_consumer(c1: dict, proj, hs, B: int):
    """(width, exponent width, exponent bias) of the out-projection's activation quantiser when the attention pass may write that
    layer's operand itself (config["mi355q_fused_attention_output"]; ops.bfp_attention(consumer=...)): one batch element, heads not
    sharded, the Linear on the per-block route -- else None"""
    if not c1.get("mi355q_fused_attention_output", False) or hs is not None or B != 1:
        return None
    ok = getattr(proj, "accepts_tiled_input", None)
    return proj.consumer_quantiser() if ok is not None and ok() else None


def _project_attention_output(o, proj, out, B, T, width, residual):
    """out_proj / o_proj behind the attention function: on its tiled operand (ops.TiledBf16) where the pass wrote one"""
    if isinstance(o, ops.TiledBf16):
        return proj.forward_tiled(o, (B, T), residual=residual)
    return out(o.transpose(1, 2).reshape(B, T, width))


@dataclass
class TinyOPTConfig:
    vocab_size: int = 512
    hidden_size: int = 256
    ffn_dim: int = 1024
    num_layers: int = 2
    num_heads: int = 4
    max_positions: int = 128
    init_std: float = 0.02        # HF OPT initializer range (modeling_opt.py:472-477)


def expand_quant_config(config: dict, num_layers: int) -> dict:
    """TOML-level dict -> per-layer node configs, as the reference's `parse_opt_quantized_config`
    (quant_config_opt.py:61-113: [default], optional [linear] / [bmm], [model_layer], [model_layer_<i>]).  A bare
    [default] body (a dict with a "name" key) is accepted as shorthand for {"default": body}."""
    if "default" not in config and "name" in config:
        config = {"default": dict(config)}
    return parse_opt_quantized_config(config, num_layers)


class _Attention(nn.Module):
    def __init__(self, cfg: TinyOPTConfig, qc: dict):
        super().__init__()
        self.h, self.nh, self.hd = cfg.hidden_size, cfg.num_heads, cfg.hidden_size // cfg.num_heads
        self.scaling = self.hd ** -0.5
        self.qc = qc
        for name in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self, name, get_quantized_cls("linear", qc[name])(self.h, self.h, bias=True, config=qc[name]))

    def forward(self, x, mask, norm=None, residual=None):
        """`residual`: returns residual + attention (config["mi355q_fused_residual"] of out_proj: the add in out_proj's stores where
        that layer's route has it, Linear.forward_residual)"""
        out_ = lambda o: self.out_proj(o) if residual is None else self.out_proj.forward_residual(o, residual)
        # head-sharded (sharded.shard_model(heads=True)): q / k / v arrive as this rank's heads only, the core runs on those, and
        # ONE all-gather of its output stands in front of out_proj
        hs = getattr(self, "mi355q_head_shard", None)
        if hs is None:
            out = out_
        else:
            from .sharded import gather_heads
            out = lambda o: out_(gather_heads(o, hs[0], hs[1]))
        B, T, _ = x.shape
        nh = self.nh if hs is None else self.q_proj.local.out_features // self.hd
        shape = lambda t: t.view(B, T, nh, self.hd).transpose(1, 2).contiguous().view(B * nh, T, self.hd)
        c1 = self.qc["bmm_1"]
        if c1["name"] == "block_fp" and c1.get("mi355q_fused_attention", False):
            # both products, the mask and the softmax in one pass per 16 queries (the harness' mask is the causal one); the
            # kernel reads the [heads, T, hd] views of the projections in place: no `_shape(...).contiguous()` copies
            heads = lambda t: t.view(B, T, nh, self.hd).transpose(1, 2)
            if c1.get("mi355q_grouped_linear", False):       # q / k / v projections: one quantisation, one GEMM launch
                qp, kp, vp = grouped_linear(x, (self.q_proj, self.k_proj, self.v_proj), norm=norm)
            else:
                qp, kp, vp = self.q_proj(x), self.k_proj(x), self.v_proj(x)
            # (q * scaling, modeling_opt.py:231: formed where the attention pass packs its Q fragments -- q_scale)
            o = get_quantized_func("attention", c1)(heads(qp), heads(kp), heads(vp), self.qc["bmm_0"], c1, causal=True,
                                                    q_scale=self.scaling, consumer=_attention_consumer(c1, self.out_proj, hs, B))
            return _project_attention_output(o, self.out_proj, out, B, T, nh * self.hd, residual)
        q = shape(self.q_proj(x) * self.scaling)
        k, v = shape(self.k_proj(x)), shape(self.v_proj(x))
        w = get_quantized_func("bmm", self.qc["bmm_0"])(q, k.transpose(1, 2), config=self.qc["bmm_0"])
        if c1["name"] in ("block_fp", "block_minifloat") and c1.get("mi355q_fused_softmax", False):
            # mask add, clamp and softmax folded into the product kernel (the harness' mask is the causal one)
            o = get_quantized_func("softmax_bmm", c1)(w, v, config=c1, causal=True)
        else:
            w = w.view(B, nh, T, T) + mask
            w = torch.max(w, w.new_full((), torch.finfo(w.dtype).min)).view(B * nh, T, T)
            p = F.softmax(w, dim=-1)
            o = get_quantized_func("bmm", c1)(p, v, config=c1)
        o = o.view(B, nh, T, self.hd).transpose(1, 2).reshape(B, T, nh * self.hd)
        return out(o)


class _DecoderLayer(nn.Module):
    def __init__(self, cfg: TinyOPTConfig, qc: dict):
        super().__init__()
        self.self_attn = _Attention(cfg, qc["self_attn"])
        self.self_attn_layer_norm = nn.LayerNorm(cfg.hidden_size)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size)
        self.fc1 = get_quantized_cls("linear", qc["fc1"])(cfg.hidden_size, cfg.ffn_dim, bias=True, config=qc["fc1"])
        self.fc2 = get_quantized_cls("linear", qc["fc2"])(cfg.ffn_dim, cfg.hidden_size, bias=True, config=qc["fc2"])

    def forward(self, x, mask):
        c1 = self.self_attn.qc["bmm_1"]
        fused_norm = (self.fc1.config.get("mi355q_fused_norm", False) and c1.get("mi355q_grouped_linear", False)
                      and c1.get("mi355q_fused_attention", False) and c1["name"] == "block_fp")
        ln = lambda m: (m.weight, m.bias, m.eps)
        fres = self.fc2.config.get("mi355q_fused_residual", False)      # the residual adds in out_proj's / fc2's stores
        if fused_norm:      # the LayerNorms are applied by the quantiser of the projections they feed (grouped_linear(norm=...))
            x = self.self_attn(x, mask, norm=ln(self.self_attn_layer_norm), residual=x) if fres else x + self.self_attn(x, mask, norm=ln(self.self_attn_layer_norm))
        else:
            x = self.self_attn(self.self_attn_layer_norm(x), mask, residual=x) if fres else x + self.self_attn(self.self_attn_layer_norm(x), mask)
        shape = x.shape
        h = x.reshape(-1, shape[-1])                       # the MLP sees a 2-D activation (modeling_opt.py:412)
        if fused_norm and self.fc2.config.get("mi355q_fused_activation", False):
            # round 6: fc1's product with relu and fc2's quantiser in its store epilogue (relu_mlp); None when the layers do not
            # qualify (fc1 not on the row-scale int8 route, fc2 not on the per-block route ...): the launches below, as before
            hr = h.contiguous()
            y = relu_mlp(h, self.fc1, self.fc2, norm=ln(self.final_layer_norm), residual=hr if fres else None)
            if y is not None:
                return (y if fres else h + y).view(shape)
        if fused_norm:
            f1 = grouped_linear(h, (self.fc1,), norm=ln(self.final_layer_norm))[0]
        else:
            f1 = None
        if self.fc2.config.get("mi355q_fused_activation", False):    # relu read by fc2's x quantiser (Linear.forward_after)
            f = f1 if fused_norm else self.fc1(self.final_layer_norm(h))
            h = self.fc2.forward_after(f, "relu", residual=h.contiguous()) if fres else h + self.fc2.forward_after(f, "relu")
        elif fused_norm:
            h = h + self.fc2(F.relu(f1))
        else:
            h = h + self.fc2(F.relu(self.fc1(self.final_layer_norm(h))))
        return h.view(shape)


class TinyOPTForCausalLM(nn.Module):
    def __init__(self, cfg: TinyOPTConfig, quant_config: dict):
        super().__init__()
        self.cfg = cfg
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.embed_positions = nn.Embedding(cfg.max_positions, cfg.hidden_size)
        self.layers = nn.ModuleList(_DecoderLayer(cfg, quant_config[f"model_layer_{i}"]) for i in range(cfg.num_layers))
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size)
        self.lm_head = nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)     # not quantised (modeling_opt.py:942-944)
        self.mi355q_lm_head = "split"
        self.apply(self._init)

    def _init(self, m):
        if isinstance(m, nn.Linear):
            m.weight.data.normal_(0.0, self.cfg.init_std)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.Embedding):
            m.weight.data.normal_(0.0, self.cfg.init_std)

    @torch.no_grad()
    def load_reference_state_dict(self, sd: dict):
        """load a state dict with the reference's (HF OPT) names: `model.decoder.` prefix, learned positions stored
        with an offset of 2 (modeling_opt.py:115-140)"""
        own = {}
        for k, v in sd.items():
            k = k.removeprefix("model.decoder.")
            v = torch.as_tensor(v)
            own[k] = v[2:2 + self.cfg.max_positions] if k == "embed_positions.weight" else v
        self.load_state_dict(own, strict=True)
        return self

    def reference_state_dict(self) -> dict:
        """this model's parameters under the reference's names (inverse of load_reference_state_dict)"""
        out = {}
        for k, v in self.state_dict().items():
            v = v.detach()
            if k == "embed_positions.weight":
                v = torch.cat([torch.zeros(2, v.shape[1], dtype=v.dtype, device=v.device), v])
            out[k if k.startswith("lm_head") else "model.decoder." + k] = v
        return out

    def forward(self, input_ids, labels=None):
        B, T = input_ids.shape
        pos = torch.arange(T, device=input_ids.device)
        x = self.embed_tokens(input_ids) + self.embed_positions(pos)[None]
        mask = torch.full((T, T), torch.finfo(x.dtype).min, device=x.device).triu(1)[None, None]
        for layer in self.layers:
            x = layer(x, mask)
        # (unquantised, modeling_opt.py:942-944: fp32-equivalent on the bf16 MFMA -- quantized_modules.linear.fp32_linear; "vendor" = F.linear)
        logits = fp32_linear(self.final_layer_norm(x), self.lm_head, self.mi355q_lm_head)
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1))
        return logits, loss


@dataclass
class TinyLlamaConfig:
    vocab_size: int = 512
    hidden_size: int = 256
    intermediate_size: int = 512
    num_layers: int = 2
    num_heads: int = 4
    max_positions: int = 128
    rms_eps: float = 1e-6
    init_std: float = 0.02


def expand_llama_quant_config(config: dict, num_layers: int) -> dict:
    """the same through `parse_llama_quantized_config` (quant_config_llama.py:70-130)"""
    if "default" not in config and "name" in config:
        config = {"default": dict(config)}
    return parse_llama_quantized_config(config, num_layers)


class _RMSNorm(nn.Module):
    def __init__(self, n, eps):
        super().__init__()
        self.weight, self.eps = nn.Parameter(torch.ones(n)), eps

    def forward(self, x):
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        return self.weight * (x * torch.rsqrt(v + self.eps)).to(x.dtype)


class _LlamaAttention(nn.Module):
    """Llama-style attention over the registry API the way the reference's LlamaQuantizedAttention drives it
    (modeling_llama.py:289-344): bias-free projections, rotary embedding through
    get_quantized_func("rotary_positional_encoding"), 4-D products through get_quantized_func("matmul")."""

    def __init__(self, cfg: TinyLlamaConfig, qc: dict):
        super().__init__()
        self.h, self.nh, self.hd = cfg.hidden_size, cfg.num_heads, cfg.hidden_size // cfg.num_heads
        self.qc = qc
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(self, name, get_quantized_cls("linear", qc[name])(self.h, self.h, bias=False, config=qc[name]))
        inv = 1.0 / (10000.0 ** (torch.arange(0, self.hd, 2).float() / self.hd))
        t = torch.arange(cfg.max_positions).float()
        emb = torch.cat([torch.outer(t, inv)] * 2, dim=-1)
        self.register_buffer("cos", emb.cos()[None, None], persistent=False)       # [1, 1, pos, hd]
        self.register_buffer("sin", emb.sin()[None, None], persistent=False)

    def forward(self, x, mask, position_ids, norm=None, residual=None):
        out_ = lambda o: self.o_proj(o) if residual is None else self.o_proj.forward_residual(o, residual)
        hs = getattr(self, "mi355q_head_shard", None)         # (head-sharded: see _Attention.forward)
        if hs is None:
            out = out_
        else:
            from .sharded import gather_heads
            out = lambda o: out_(gather_heads(o, hs[0], hs[1]))
        B, T, _ = x.shape
        nh = self.nh if hs is None else self.q_proj.local.out_features // self.hd
        shape = lambda t: t.view(B, T, nh, self.hd).transpose(1, 2)
        if self.qc["matmul_1"].get("mi355q_grouped_linear", False):
            q, k, v = (shape(t) for t in grouped_linear(x, (self.q_proj, self.k_proj, self.v_proj), norm=norm))
        else:
            if norm is not None:
                # (the layer asked for the fused norm but this module's own knobs do not group the projections -- the knobs
                #  ride per node: the norm is applied here, the RMSNorm expression of the reference, never skipped)
                weight, eps = norm
                var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
                x = weight * (x * torch.rsqrt(var + eps)).to(x.dtype)
            q, k, v = shape(self.q_proj(x)), shape(self.k_proj(x)), shape(self.v_proj(x))
        rc = self.qc["rotary_positional_encoding"]
        c1 = self.qc["matmul_1"]
        fused = c1["name"] == "block_fp" and c1.get("mi355q_fused_attention", False)
        if fused and c1.get("mi355q_fused_rotary", False):
            # (the rotary embedding applied where the attention pass loads q and k: attention_block_fp(rope=...))
            o = get_quantized_func("attention", c1)(q, k, v, self.qc["matmul_0"], c1, causal=True, scale_div=math.sqrt(self.hd),
                                                    rope=(self.cos[:, :, :T], self.sin[:, :, :T], position_ids, rc),
                                                    consumer=_attention_consumer(c1, self.o_proj, hs, B))
            return _project_attention_output(o, self.o_proj, out, B, T, nh * self.hd, residual)
        q, k = get_quantized_func("rotary_positional_encoding", rc)(q, k, self.cos[:, :, :T], self.sin[:, :, :T],
                                                                   position_ids, config=rc)
        if fused:
            o = get_quantized_func("attention", c1)(q, k, v, self.qc["matmul_0"], c1, causal=True, scale_div=math.sqrt(self.hd),
                                                    consumer=_attention_consumer(c1, self.o_proj, hs, B))
            return _project_attention_output(o, self.o_proj, out, B, T, nh * self.hd, residual)
        w = get_quantized_func("matmul", self.qc["matmul_0"])(q, k.transpose(2, 3), config=self.qc["matmul_0"])
        if c1["name"] in ("block_fp", "block_minifloat") and c1.get("mi355q_fused_softmax", False):
            o = get_quantized_func("softmax_matmul", c1)(w / math.sqrt(self.hd), v, config=c1, causal=True)
        else:
            w = w / math.sqrt(self.hd) + mask
            w = torch.max(w, w.new_full((), torch.finfo(w.dtype).min))
            p = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
            o = get_quantized_func("matmul", c1)(p, v, config=c1)
        return out(o.transpose(1, 2).reshape(B, T, nh * self.hd))


class _LlamaLayer(nn.Module):
    def __init__(self, cfg: TinyLlamaConfig, qc: dict):
        super().__init__()
        self.self_attn = _LlamaAttention(cfg, qc["self_attn"])
        self.input_layernorm = _RMSNorm(cfg.hidden_size, cfg.rms_eps)
        self.post_attention_layernorm = _RMSNorm(cfg.hidden_size, cfg.rms_eps)
        lin = lambda name, i, o: get_quantized_cls("linear", qc["mlp"][name])(i, o, bias=False, config=qc["mlp"][name])
        self.gate_proj = lin("gate_proj", cfg.hidden_size, cfg.intermediate_size)
        self.up_proj = lin("up_proj", cfg.hidden_size, cfg.intermediate_size)
        self.down_proj = lin("down_proj", cfg.intermediate_size, cfg.hidden_size)

    def forward(self, x, mask, position_ids):
        gc = self.gate_proj.config
        fused_norm = gc.get("mi355q_grouped_linear", False) and gc.get("mi355q_fused_norm", False)
        fres = self.down_proj.config.get("mi355q_fused_residual", False)      # the residual adds in o_proj's / down_proj's stores
        if fused_norm:      # the norms are applied by the quantiser of the projections they feed (grouped_linear(norm=...))
            n1, n2 = self.input_layernorm, self.post_attention_layernorm
            x = (self.self_attn(x, mask, position_ids, norm=(n1.weight, n1.eps), residual=x) if fres
                 else x + self.self_attn(x, mask, position_ids, norm=(n1.weight, n1.eps)))
            if self.down_proj.config.get("mi355q_fused_activation", False):
                # round 6: gate / up interleaved in ONE product whose epilogue writes down_proj's quantised operand (gated_mlp); None
                # when the layers do not qualify: the grouped launch + the quantiser that reads silu(gate) * up, as before
                y = gated_mlp(x, self.gate_proj, self.up_proj, self.down_proj, norm=(n2.weight, n2.eps), residual=x if fres else None)
                if y is not None:
                    return y if fres else x + y
            gate, up = grouped_linear(x, (self.gate_proj, self.up_proj), norm=(n2.weight, n2.eps))
            if self.down_proj.config.get("mi355q_fused_activation", False):
                return self.down_proj.forward_after(gate, "silu_mul", up, residual=x) if fres else x + self.down_proj.forward_after(gate, "silu_mul", up)
            return x + self.down_proj(F.silu(gate) * up)
        x = self.self_attn(self.input_layernorm(x), mask, position_ids, residual=x) if fres else x + self.self_attn(self.input_layernorm(x), mask, position_ids)
        h = self.post_attention_layernorm(x)
        if self.gate_proj.config.get("mi355q_grouped_linear", False):
            gate, up = grouped_linear(h, (self.gate_proj, self.up_proj))
        else:
            gate, up = self.gate_proj(h), self.up_proj(h)
        if self.down_proj.config.get("mi355q_fused_activation", False):    # silu(gate) * up read by down_proj's x quantiser
            return self.down_proj.forward_after(gate, "silu_mul", up, residual=x) if fres else x + self.down_proj.forward_after(gate, "silu_mul", up)
        return x + self.down_proj(F.silu(gate) * up)                      # (modeling_llama.py:208-240)


class TinyLlamaForCausalLM(nn.Module):
    def __init__(self, cfg: TinyLlamaConfig, quant_config: dict):
        super().__init__()
        self.cfg = cfg
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList(_LlamaLayer(cfg, quant_config[f"model_layer_{i}"]) for i in range(cfg.num_layers))
        self.norm = _RMSNorm(cfg.hidden_size, cfg.rms_eps)
        self.lm_head = nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
        self.mi355q_lm_head = "split"
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(0.0, cfg.init_std)

    @torch.no_grad()
    def load_reference_state_dict(self, sd: dict):
        """load a state dict with the reference's (HF Llama) names: `model.` prefix, `layers.i.mlp.*` projections,
        `rotary_emb.inv_freq` buffers skipped (the tables are rebuilt from the same formula)"""
        own = {}
        for k, v in sd.items():
            k = k.removeprefix("model.").replace(".mlp.", ".")
            if k.endswith("rotary_emb.inv_freq"):
                continue
            own[k] = torch.as_tensor(v)
        self.load_state_dict(own, strict=True)
        return self

    def reference_state_dict(self) -> dict:
        out = {}
        for k, v in self.state_dict().items():
            for proj in ("gate_proj", "up_proj", "down_proj"):
                k = k.replace("." + proj, ".mlp." + proj)
            out[k if k.startswith("lm_head") else "model." + k] = v.detach()
        return out

    def forward(self, input_ids, labels=None):
        B, T = input_ids.shape
        position_ids = torch.arange(T, device=input_ids.device)[None].expand(B, T)
        x = self.embed_tokens(input_ids)
        mask = torch.full((T, T), torch.finfo(x.dtype).min, device=x.device).triu(1)[None, None]
        for layer in self.layers:
            x = layer(x, mask, position_ids)
        # (unquantised, modeling_llama.py:772,866: fp32-equivalent on the bf16 MFMA -- quantized_modules.linear.fp32_linear; "vendor" = F.linear)
        logits = fp32_linear(self.norm(x), self.lm_head, self.mi355q_lm_head)
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1))
        return logits, loss


@torch.no_grad()
def eval_lm_perplexity(model, batches, device=None):
    """Reference eval/eval_lm.py:41-63: per batch loss * batch * seq_len summed, ppl = exp(sum / tokens)."""
    total, num_samples, seq_len, batch_size = 0.0, 0, None, None
    for input_ids in batches:
        if device is not None:
            input_ids = input_ids.to(device)
        batch_size, seq_len = input_ids.shape
        _, loss = model(input_ids, labels=input_ids)
        total += loss.item() * batch_size * seq_len
        num_samples += batch_size
    reduced = total / (seq_len * num_samples)
    try:
        ppl = math.exp(reduced)
    except OverflowError:
        ppl = float("inf")
    return {"loss": reduced, "perplexity": ppl, "num_samples": num_samples, "seq_len": seq_len, "batch_size": batch_size}
