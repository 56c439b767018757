"""Quantiser registry -- same keys and call signatures as the reference's
`models/quantize/quantizers/__init__.py:8-16`; all seven run as HIP kernels (mi355q.ops), backward is
the reference's straight-through estimator."""
from __future__ import annotations

import torch

from ... import ops


class _STE(torch.autograd.Function):
    """forward: HIP kernel named by `kind`; backward: identity on x (block_fp.py:119-124,
    block_minifloat.py:102-112, block_log.py:96-97, integer.py:69-74)."""

    @staticmethod
    def forward(ctx, x, kind, args):
        return _FORWARD[kind](x, *args)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None, None


_FORWARD = {
    "block_fp": lambda x, w, ew, eb, bs, skip: ops.block_fp_quantize(x, w, ew, eb, bs, skip),
    "block_minifloat": lambda x, w, ew, ebw, bs, skip: ops.block_minifloat_quantize(x, w, ew, ebw, bs, skip),
    "block_log": lambda x, w, ebw, bs, skip: ops.block_log_quantize(x, w, ebw, bs, skip),
    "integer": lambda x, w, fw, signed: ops.integer_quantize(x, w, fw, signed),
    "minifloat_ieee": lambda x, w, ew, eb: ops.minifloat_quantize(x, w, ew, eb),
    "minifloat_denorm": lambda x, w, ew, eb: ops.minifloat_quantize(x, w, ew, eb, denorm=True),
    "log": lambda x, w, eb: ops.log_quantize(x, w, eb),
}


def block_fp_quantizer(x, width: int = 12, exponent_width: int = 8, exponent_bias: int = None,
                       block_size=[16], skip_first_dim: bool = True):
    """reference block_fp.py:127-153"""
    return _STE.apply(x, "block_fp", (width, exponent_width, exponent_bias, block_size, skip_first_dim))


def block_minifloat_quantizer(x, width: int, exponent_width: int, exponent_bias_width: int,
                              block_size=[16], skip_first_dim: bool = False):
    """reference block_minifloat.py:118-146"""
    return _STE.apply(x, "block_minifloat", (width, exponent_width, exponent_bias_width, block_size, skip_first_dim))


def block_log_quantizer(x, width: int, exponent_bias_width: int = None, block_size=[16],
                        skip_first_dim: bool = False):
    """reference block_log.py:111-134"""
    return _STE.apply(x, "block_log", (width, exponent_bias_width, block_size, skip_first_dim))


def integer_quantizer(x, width: int, frac_width: int, is_signed: bool = True):
    """reference integer.py:77-97 (RoPE tables in every shipped TOML); python ints pass through"""
    if isinstance(x, int):
        return x
    return _STE.apply(x, "integer", (width, frac_width, is_signed))


def minifloat_ieee_quantizer(x, width: int, exponent_width: int, exponent_bias: int = None):
    """reference minifloat.py:216-240"""
    return _STE.apply(x, "minifloat_ieee", (width, exponent_width, exponent_bias))


def minifloat_denorm_quantizer(x, width: int, exponent_width: int, exponent_bias: int = None):
    """reference minifloat.py:106-131"""
    return _STE.apply(x, "minifloat_denorm", (width, exponent_width, exponent_bias))


def log_quantizer(x, width: int, exponent_bias: int = None):
    """reference log.py:59-87 (backward: identity on x, log.py:66-69)"""
    return _STE.apply(x, "log", (width, exponent_bias))


QUANTIZER_MAP = {
    "block_fp": block_fp_quantizer,
    "block_log": block_log_quantizer,
    "block_minifloat": block_minifloat_quantizer,
    "integer": integer_quantizer,
    "log": log_quantizer,
    "minifloat_denorm": minifloat_denorm_quantizer,
    "minifloat_ieee": minifloat_ieee_quantizer,
}
