"""Quantised matmul / bmm / rotary-embedding functions, `f(x, y, config)` and
`f(q, k, cos, sin, position_ids, config)` (reference quantized_functions/matmul.py:146-353,
rotary_positional_encoding.py:59-248; registry keys as quantized_functions/__init__.py:14-42).

Each operand is quantised along ITS OWN last dimension (matmul.py:166-195), so only x is blocked
along the contraction: these products are not int8 block dots.  block_fp products
whose blocks tile K and N go through the fused kernel (`ops.bfp_matmul`: x quantised in registers on
its way into bf16 MFMAs, one pass over x); everything else through the HIP fake-quant kernels and
the GPU's fp32 GEMM on exactly representable values.  Reference quirks kept: "log" maps to the block_log functions
(__init__.py:20,29); block_log leaves y unquantised (matmul.py:278-297); the block_fp rotary
function ignores `bypass` (rotary_positional_encoding.py:59-82)."""
from __future__ import annotations

import torch

from ..quantizers import QUANTIZER_MAP

_MATMUL = {"matmul": torch.matmul, "bmm": torch.bmm}


def _ops():
    from ... import ops
    return ops

_KEYS = {
    "block_fp": ("width", "exponent_width", "exponent_bias", "block_size"),
    "block_minifloat": ("width", "exponent_width", "exponent_bias_width", "block_size"),
    "block_log": ("width", "exponent_bias_width", "block_size"),
    "integer": ("width", "frac_width"),
    "minifloat_ieee": ("width", "exponent_width", "exponent_bias"),
    "minifloat_denorm": ("width", "exponent_width", "exponent_bias"),
    "log": ("width", "exponent_bias"),
}
_BLOCKED = ("block_fp", "block_minifloat", "block_log")


def _quantise_operand(t, arith, config, prefix):
    kw = {k: config[f"{prefix}_{k}"] for k in _KEYS[arith]}
    q = QUANTIZER_MAP[arith]
    if arith not in _BLOCKED:
        return q(t, **kw)
    many = t.ndim > 2                    # matmul.py:166-167, 187-195
    flat = torch.flatten(t, 0, -3) if many else t
    return torch.reshape(q(flat, **kw, skip_first_dim=many), t.shape)


def _fused_block_fp_matmul(x, y, config, style, softmax=False, mask=None, causal=False):
    """one pass over x: quantise inside the product kernel (ops.bfp_matmul) where shapes, block sizes and widths
    allow; None -> the caller takes the two-quantisers + GEMM route.  Autograd (QAT) stays on that route too.
    `softmax`: x holds scores whose row softmax is the quantiser's input (ops.bfp_matmul(..., softmax=True))."""
    from .. import quantizers as _q                                   # noqa: F401  (registry import order)
    from ... import ops
    if not (x.is_cuda and y.is_cuda) or x.dtype != torch.float32 or y.dtype != torch.float32:
        return None
    if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad):
        return None
    if x.ndim != y.ndim or x.ndim < 2 or x.shape[:-2] != y.shape[:-2] or x.shape[-1] != y.shape[-2]:
        return None                                                   # (broadcasting products: general route)
    x3 = x.reshape(-1, *x.shape[-2:]) if x.ndim != 3 else x
    y3 = y.reshape(-1, *y.shape[-2:]) if y.ndim != 3 else y
    many = x.ndim > 2
    for t, prefix in ((x3, "data_in"), (y3, "weight")):
        shape = t.shape if many else t.shape[-2:]
        if ops.resolve_blocking(list(shape), config[f"{prefix}_block_size"], many)[3:] != (1, 16):
            return None
    supported = ops.bfp_softmax_matmul_supported if softmax else ops.bfp_matmul_supported
    if not supported(x3, y3, config["data_in_width"], config["weight_width"]):
        return None
    out = ops.bfp_matmul(x3, y3, config["data_in_width"], config["data_in_exponent_width"], config["data_in_exponent_bias"],
                         config["weight_width"], config["weight_exponent_width"], config["weight_exponent_bias"], softmax=softmax,
                         mask=mask, causal=causal)
    return out.reshape(*x.shape[:-1], y.shape[-1])


def _fused_values_matmul(x, y, config, arith, softmax=False, mask=None, causal=False):
    """matmul_block_minifloat / matmul_block_log (matmul.py:199-249, 252-297) in the library's own product kernels
    (ops.values_matmul: y packed transposed once, x quantised in registers on its way into bf16 MFMAs -- one pass over x, no
    quantised copy of either operand; block_log's unquantised y as three exact bf16 planes).  The default route
    (config["mi355q_values_matmul"] = "fused"); None: shapes / settings it does not take -> the caller's next route."""
    from ... import ops
    if config.get("mi355q_values_matmul", "fused") != "fused" or not (x.is_cuda and y.is_cuda):
        return None
    if x.dtype != torch.float32 or y.dtype != torch.float32 or torch.is_grad_enabled() and (x.requires_grad or y.requires_grad):
        return None
    if x.ndim != y.ndim or x.ndim < 2 or x.shape[:-2] != y.shape[:-2] or x.shape[-1] != y.shape[-2]:
        return None
    many = x.ndim > 2
    x3 = x.reshape(-1, *x.shape[-2:]) if x.ndim != 3 else x
    y3 = y.reshape(-1, *y.shape[-2:]) if y.ndim != 3 else y
    for t, prefix in ((x3, "data_in"),) + (((y3, "weight"),) if arith == "block_minifloat" else ()):
        shape = t.shape if many else t.shape[-2:]
        if ops.resolve_blocking(list(shape), config[f"{prefix}_block_size"], many)[3:] != (1, 16):
            return None
    keys = _KEYS[arith][:-1]
    xp = tuple(config[f"data_in_{k}"] for k in keys)
    yp = tuple(config[f"weight_{k}"] for k in keys) if arith == "block_minifloat" else None
    if not ops.values_matmul_supported(x3, y3, arith, xp, yp, softmax):
        return None
    out = ops.values_matmul(x3, y3, arith, xp, yp, softmax=softmax, mask=mask, causal=causal)
    return out.reshape(*x.shape[:-1], y.shape[-1])


def _generic_matmul(x, y, config, arith, style):
    mm = _MATMUL[style]
    if config.get("bypass", False):
        _ops().count_vendor_gemm("matmul.bypass")
        return mm(x, y)
    # read y's keys first-to-last like the reference does (KeyError parity) even where unused
    for k in _KEYS[arith]:
        config[f"data_in_{k}"], config[f"weight_{k}"]
    if arith == "block_fp" and config.get("mi355q_fused_matmul", True):
        out = _fused_block_fp_matmul(x, y, config, style)
        if out is not None:
            return out
    if arith in ("block_minifloat", "block_log"):
        out = _fused_values_matmul(x, y, config, arith)
        if out is not None:
            return out
    xq = _quantise_operand(x, arith, config, "data_in")
    yq = y if arith == "block_log" else _quantise_operand(y, arith, config, "weight")
    _ops().count_vendor_gemm("matmul.generic")
    return mm(xq, yq)


def _mask_2d(mask, tq, tk):
    """an additive mask that is the same for every leading index, as the [T_q, T_k] tensor the HIP entry points read --
    [1, 1, T_q, T_k], [T_q, T_k], or a padding mask [1, 1, 1, T_k] expanded over the rows; None for anything else (per-
    batch masks, shapes that do not broadcast to [T_q, T_k]): the caller then takes the generic route"""
    # (fp32 additive masks only: a bool mask would become an additive 0 / 1 one under a plain cast, and a half-precision
    #  mask's finfo.min is not fp32's -- both take the generic route, which treats them as torch does)
    if mask.dtype != torch.float32 or mask.ndim < 2 or mask.numel() != mask.shape[-2] * mask.shape[-1]:
        return None
    m2 = mask.reshape(mask.shape[-2:])
    if m2.shape[0] not in (1, tq) or m2.shape[1] not in (1, tk):
        return None
    return m2.expand(tq, tk).contiguous()


def _make_softmax(style, arith="block_fp"):
    """`softmax_{matmul,bmm}_{block_fp,block_minifloat}(scores, y, config, mask=None, causal=False)` =
    `{matmul,bmm}_block_fp(softmax(max(scores + mask, finfo.min), dim=-1), y, config)`: what the reference's attention
    computes between its two products (modeling_opt.py:262-312, modeling_llama.py:318-344), as ONE call, so that neither
    the masked scores nor the probability tensor [heads, T, T] need exist.  `mask`: additive, [T_q, T_k]
    (or broadcastable to it over the leading dims); `causal=True` stands for the causal mask without reading one.  An
    addition to the registry (keys "softmax_matmul" / "softmax_bmm"); callers that keep the reference's steps are served
    as before."""
    def f(scores, y, config, mask=None, causal=False):
        m2 = None if mask is None else _mask_2d(mask, scores.shape[-2], scores.shape[-1])
        if (not config.get("bypass", False) and config.get("mi355q_fused_matmul", True) and (mask is None or m2 is not None)):
            for k in _KEYS[arith]:
                config[f"data_in_{k}"], config[f"weight_{k}"]
            if arith == "block_fp":
                out = _fused_block_fp_matmul(scores, y, config, style, softmax=True, mask=m2, causal=causal)
            else:
                out = _fused_values_matmul(scores, y, config, arith, softmax=True, mask=m2, causal=causal)
            if out is not None:
                return out
        w = scores
        if causal:
            tq, tk = scores.shape[-2], scores.shape[-1]
            w = w + torch.full((tq, tk), torch.finfo(w.dtype).min, device=w.device).triu(1 + tk - tq)
        if mask is not None:
            w = w + mask
        if causal or mask is not None:
            w = torch.max(w, w.new_full((), torch.finfo(w.dtype).min))
        p = torch.nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(scores.dtype)
        return _generic_matmul(p, y, config, arith, style)
    f.__name__ = f"softmax_{style}_{arith}"
    return f


def attention_block_fp(q, k, v, config_qk, config_pv, mask=None, causal=False, scale_div=None, rope=None, consumer=None, q_scale=None):
    """The reference's quantised attention core as one call (modeling_opt.py:246-312, modeling_llama.py:309-344):

        w = bmm_0(q, k^T)  [w = w / scale_div]  w = max(w + mask, finfo.min)  p = softmax(w, -1)  out = bmm_1(p, v)

    `q` [..., T_q, hd], `k`, `v` [..., T_k, hd] (k NOT transposed); `config_qk` / `config_pv` are the configs of the two
    products (OPT: bmm_0 / bmm_1, Llama: matmul_0 / matmul_1); `mask` additive [T_q, T_k] (or broadcastable to it over the
    leading dims), `causal=True` stands for the causal mask without reading one; `scale_div`: Llama's sqrt(head_dim).  On
    the HIP path neither scores nor probabilities are ever written (ops.bfp_attention); shapes it does not take, other
    arithmetics and autograd fall back to the same steps through the registry's own functions.  With
    `config_pv["mi355q_token_major_output"]` and q [1, heads, T, hd] the result is the [1, heads, T, hd] view of a
    contiguous [1, T, heads, hd] buffer (same values): the `transpose(1, 2).reshape(B, T, hidden)` that follows in both
    models is then free.  An addition to the registry (key "attention").
    `rope` = (cos, sin, position_ids, rotary config): q and k are the projections BEFORE the rotary embedding
    (modeling_llama.py:289-299), which the HIP pass applies as it loads them (ops.bfp_attention(rope=...): the turned q / k are
    never written); wherever that does not apply, the registry's rotary function runs first -- the same values either way.
    `consumer` = (width, exponent width, exponent bias) of the out-projection's data_in block_fp quantiser: where the HIP pass can
    (one batch element, head_dim 64 / 128, no additive mask) the result is `ops.TiledBf16` -- that Linear's quantised activations
    [T_q, heads x hd], for `Linear.forward_tiled` -- instead of the fp32 tensor; the caller checks which it got.
    `q_scale`: the core runs on q * q_scale (OPT: `self.q_proj(hidden_states) * self.scaling`, modeling_opt.py:231) -- multiplied where
    the HIP pass packs its Q fragments, by a torch kernel in front everywhere else: the same values."""
    from ... import ops
    if rope is not None:
        cos, sin, position_ids, rope_config = rope
        rope_fn = QUANTIZED_FUNC_MAP["rotary_positional_encoding"][rope_config["name"]]
    m2 = None if mask is None else _mask_2d(mask, q.shape[-2], k.shape[-2])
    both_fp = config_qk.get("name") == "block_fp" and config_pv.get("name") == "block_fp"
    fused_ok = (both_fp and not config_qk.get("bypass", False) and not config_pv.get("bypass", False)
                and config_qk.get("mi355q_fused_matmul", True) and config_pv.get("mi355q_fused_matmul", True)
                and (mask is None or m2 is not None) and q.ndim == k.ndim == v.ndim and q.ndim >= 3
                and q.shape[:-2] == k.shape[:-2] == v.shape[:-2]
                and not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)))
    if fused_ok:
        for c in (config_qk, config_pv):
            for key in _KEYS["block_fp"]:
                c[f"data_in_{key}"], c[f"weight_{key}"]
        nb, (tq, hd), tk = q.shape[:-2].numel(), q.shape[-2:], k.shape[-2]
        blocks_ok = all(ops.resolve_blocking(list(shape), bs, True)[3:] == (1, 16) for shape, bs in (
            ((nb, tq, hd), config_qk["data_in_block_size"]), ((nb, hd, tk), config_qk["weight_block_size"]),
            ((nb, tq, tk), config_pv["data_in_block_size"]), ((nb, tk, hd), config_pv["weight_block_size"])))
        widths = (config_qk["data_in_width"], config_qk["weight_width"], config_pv["data_in_width"], config_pv["weight_width"])
        if blocks_ok and ops.bfp_attention_supported(q, k, v, widths) and (not causal or tk >= tq):
            par = lambda c: (c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"], c["weight_width"],
                             c["weight_exponent_width"], c["weight_exponent_bias"])
            qs = q_scale if q_scale and rope is None and ops.bfp_attention_q_scale_supported(q, k) else None
            if q_scale and qs is None:
                q, q_scale = q * q_scale, None
            rope_in = None
            if rope is not None:
                cos_q, sin_q = (t.contiguous() for t in rope_tables(cos, sin, rope_config))
                pos_c = position_ids.contiguous()
                if (not (torch.is_grad_enabled() and (cos_q.requires_grad or sin_q.requires_grad))
                        and ops.bfp_attention_rope_supported(q, k, cos_q, sin_q, pos_c)):
                    rope_in = (cos_q, sin_q, pos_c)
                else:
                    q, k = rope_fn(q, k, cos, sin, position_ids, config=rope_config)
            return ops.bfp_attention(q, k, v, par(config_qk), par(config_pv), mask=m2,
                                     causal=causal, scale_div=scale_div,
                                     token_major=bool(config_pv.get("mi355q_token_major_output", False)), rope=rope_in,
                                     consumer=consumer if consumer is not None and ops.bfp_attention_consumer_supported(q, m2) else None,
                                     q_scale=qs)
    if q_scale:
        q = q * q_scale
    if rope is not None:
        q, k = rope_fn(q, k, cos, sin, position_ids, config=rope_config)
    style = "bmm" if q.ndim == 3 else "matmul"
    w = QUANTIZED_FUNC_MAP[style][config_qk["name"]](q, k.transpose(-1, -2), config=config_qk)
    if scale_div:
        w = w / scale_div
    if config_pv["name"] == "block_fp":
        return QUANTIZED_FUNC_MAP["softmax_" + style]["block_fp"](w, v, config_pv, mask=mask, causal=causal)
    if causal:
        tq, tk = w.shape[-2], w.shape[-1]
        w = w + torch.full((tq, tk), torch.finfo(w.dtype).min, device=w.device).triu(1 + tk - tq)
    if mask is not None:
        w = w + mask
    if causal or mask is not None:
        w = torch.max(w, w.new_full((), torch.finfo(w.dtype).min))
    p = torch.nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    return QUANTIZED_FUNC_MAP[style][config_pv["name"]](p, v, config=config_pv)


def _make(arith, style):
    def f(x, y, config):
        return _generic_matmul(x, y, config, arith, style)
    f.__name__ = f"{style}_{arith}"
    return f


matmul_integer, bmm_integer = _make("integer", "matmul"), _make("integer", "bmm")
matmul_minifloat_denorm, bmm_minifloat_denorm = _make("minifloat_denorm", "matmul"), _make("minifloat_denorm", "bmm")
matmul_minifloat_ieee, bmm_minifloat_ieee = _make("minifloat_ieee", "matmul"), _make("minifloat_ieee", "bmm")
matmul_log, bmm_log = _make("log", "matmul"), _make("log", "bmm")
matmul_block_fp, bmm_block_fp = _make("block_fp", "matmul"), _make("block_fp", "bmm")
softmax_matmul_block_fp, softmax_bmm_block_fp = _make_softmax("matmul"), _make_softmax("bmm")
softmax_matmul_block_minifloat = _make_softmax("matmul", "block_minifloat")
softmax_bmm_block_minifloat = _make_softmax("bmm", "block_minifloat")
matmul_block_minifloat, bmm_block_minifloat = _make("block_minifloat", "matmul"), _make("block_minifloat", "bmm")
matmul_block_log, bmm_block_log = _make("block_log", "matmul"), _make("block_log", "bmm")


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _rope_kernel_applies(q, k, cos_q, sin_q, position_ids) -> bool:
    return (q.is_cuda and k.is_cuda and q.dtype == k.dtype == cos_q.dtype == torch.float32 and q.ndim == 4 and k.ndim == 4
            and q.shape[0] == k.shape[0] and q.shape[2:] == k.shape[2:] and q.shape[3] % 8 == 0 and q.stride(3) == 1
            and k.stride(3) == 1 and all(s % 4 == 0 for s in q.stride()[:3] + k.stride()[:3])
            and q.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0
            and cos_q.ndim == 2 and cos_q.shape == sin_q.shape and cos_q.shape[1] == q.shape[3]
            and position_ids.dtype == torch.int64 and position_ids.shape == (q.shape[0], q.shape[2])
            and not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad)))


# cos / sin tables are constants of the model: their quantised images are kept (the reference re-quantises both on every
# call, rotary_positional_encoding.py:59-82 -- same values).  Keyed by storage, shape, strides, version counter and
# quantiser parameters; the record holds the source tensor, so its address cannot have been re-used.  Not consulted while
# a HIP graph is being recorded (a replay must not depend on a buffer the cache may drop).
_ROPE_TABLES = {}


def _quantised_table(t, quant, sig):
    from ... import ops
    if not t.is_cuda or t.requires_grad or ops._capturing():
        return quant(t)
    try:
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, sig)
    except RuntimeError:                      # (inference-mode tensors have no version counter)
        return quant(t)
    hit = _ROPE_TABLES.get(key)
    if hit is None:
        if len(_ROPE_TABLES) >= 32:
            _ROPE_TABLES.clear()
        hit = _ROPE_TABLES[key] = (t.detach(), quant(t).contiguous())
    return hit[1]


def _rope_tables(arith, honours_bypass, cos, sin, config):
    """the two tables as the reference's function quantises them (rotary_positional_encoding.py:59-82), [rows, D]"""
    if honours_bypass and config.get("bypass", False):
        quant = lambda t: t
    else:
        kw = {key: config[f"data_in_{key}"] for key in _KEYS[arith]}
        if arith in _BLOCKED:
            kw["skip_first_dim"] = False
        elif arith == "integer":
            kw["is_signed"] = True
        raw = lambda t: QUANTIZER_MAP[arith](t, **kw)
        sig = (arith,) + tuple((key, str(val)) for key, val in sorted(kw.items()))
        wants_grad = torch.is_grad_enabled() and (cos.requires_grad or sin.requires_grad)
        quant = raw if wants_grad else (lambda t: _quantised_table(t, raw, sig))
    return quant(cos.squeeze(1).squeeze(0)), quant(sin.squeeze(1).squeeze(0))


def rope_tables(cos, sin, config):
    """(cos_q, sin_q) [rows, D] of `get_quantized_func("rotary_positional_encoding", config)`: what that function multiplies q and k
    with -- for callers that hand the embedding to the attention pass (attention_block_fp(rope=...)) instead of calling it."""
    arith = config["name"]
    return _rope_tables(arith, arith != "block_fp", cos, sin, config)


def _make_rope(arith, honours_bypass=True):
    def f(q, k, cos, sin, position_ids, config):
        cos_q, sin_q = _rope_tables(arith, honours_bypass, cos, sin, config)
        if _rope_kernel_applies(q, k, cos_q, sin_q, position_ids):
            # one launch for q and k, position lookup included (ops.rope_apply), instead of ten elementwise kernels
            from ... import ops
            return ops.rope_apply(q, k, cos_q.contiguous(), sin_q.contiguous(), position_ids.contiguous())
        cos = cos_q[position_ids].unsqueeze(1)   # [bs, 1, seq, dim]
        sin = sin_q[position_ids].unsqueeze(1)
        return (q * cos) + (_rotate_half(q) * sin), (k * cos) + (_rotate_half(k) * sin)
    f.__name__ = f"apply_rotary_pos_emb_{arith}"
    return f


apply_rotary_pos_emb_block_fp = _make_rope("block_fp", honours_bypass=False)
apply_rotary_pos_emb_block_log = _make_rope("block_log")
apply_rotary_pos_emb_block_minifloat = _make_rope("block_minifloat")
apply_rotary_pos_emb_integer = _make_rope("integer")
apply_rotary_pos_emb_log = _make_rope("log")
apply_rotary_pos_emb_minifloat_denorm = _make_rope("minifloat_denorm")
apply_rotary_pos_emb_minifloat_ieee = _make_rope("minifloat_ieee")

QUANTIZED_FUNC_MAP = {
    "matmul": {
        "block_fp": matmul_block_fp, "block_log": matmul_block_log, "block_minifloat": matmul_block_minifloat,
        "integer": matmul_integer, "log": matmul_block_log, "minifloat_denorm": matmul_minifloat_denorm,
        "minifloat_ieee": matmul_minifloat_ieee,
    },
    "bmm": {
        "block_fp": bmm_block_fp, "block_log": bmm_block_log, "block_minifloat": bmm_block_minifloat,
        "integer": bmm_integer, "log": bmm_block_log, "minifloat_denorm": bmm_minifloat_denorm,
        "minifloat_ieee": bmm_minifloat_ieee,
    },
    "softmax_matmul": {"block_fp": softmax_matmul_block_fp, "block_minifloat": softmax_matmul_block_minifloat},
    "softmax_bmm": {"block_fp": softmax_bmm_block_fp, "block_minifloat": softmax_bmm_block_minifloat},
    "attention": {"block_fp": attention_block_fp},
    "rotary_positional_encoding": {
        "block_fp": apply_rotary_pos_emb_block_fp, "block_log": apply_rotary_pos_emb_block_log,
        "block_minifloat": apply_rotary_pos_emb_block_minifloat, "integer": apply_rotary_pos_emb_integer,
        "log": apply_rotary_pos_emb_log, "minifloat_denorm": apply_rotary_pos_emb_minifloat_denorm,
        "minifloat_ieee": apply_rotary_pos_emb_minifloat_ieee,
    },
}
