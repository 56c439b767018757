"""torch-CPU restatement of the reference's fake-quant path, op for op (TEST INFRASTRUCTURE).

Used for two things only: (1) `bench.py`'s `cpu_baseline` leg -- the reference's own Python cannot
travel to the GPU box, so the same torch elementwise chain + fp32 `F.linear` is timed on the
box's host cores; (2) a second, torch-kernel-level pin of the numpy oracle.
Parity status: PINNED against tests/golden (tests/test_torch_port.py).

Op order follows reference quantizers/block_fp.py:44-94 (abs-max -> zero fill -> sign -> +1e-9 ->
ceil(log2) -> clamp -> /2**e -> *2**mb -> round -> clamp -> rescale -> isclose mix) and
quantized_modules/linear.py:63-71.  Blocking is done with pad + reshape/permute instead of the
reference's F.unfold/F.fold; the arithmetic per block is the same.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .np_oracle import block_meta


def _to_blocks(x: torch.Tensor, meta):
    plane = x.reshape(meta.lead, meta.crop_rows, meta.crop_cols)
    plane = F.pad(plane, (0, meta.cols - meta.crop_cols, 0, meta.rows - meta.crop_rows))
    t = plane.reshape(meta.lead, meta.rows // meta.b0, meta.b0, meta.cols // meta.b1, meta.b1)
    if meta.b0 == 1:
        return t.reshape(meta.n_blocks, meta.b1)
    return t.permute(0, 1, 3, 2, 4).reshape(meta.n_blocks, meta.block_elems)


def _from_blocks(b: torch.Tensor, meta):
    t = b.reshape(meta.lead, meta.rows // meta.b0, meta.cols // meta.b1, meta.b0, meta.b1)
    if meta.b0 != 1:
        t = t.permute(0, 1, 3, 2, 4)
    plane = t.reshape(meta.lead, meta.rows, meta.cols)[:, :meta.crop_rows, :meta.crop_cols]
    return plane.reshape(meta.x_shape)


def block_fp_quantize(x: torch.Tensor, width: int, exponent_width: int = 8, exponent_bias=None,
                      block_size=(16,), skip_first_dim: bool = True) -> torch.Tensor:
    meta = block_meta(tuple(x.shape), block_size, skip_first_dim)
    blocked = _to_blocks(x, meta)
    bmax = blocked.abs().max(dim=1, keepdim=True)[0]
    if torch.all(bmax == 0):
        bmax = torch.ones_like(bmax)
    else:
        bmax[bmax == 0] = bmax[bmax != 0].min()
    mbits = width - 1
    if exponent_bias in (None, "none", "None"):
        exponent_bias = 2 ** (exponent_width - 1) - 1
    e_max, e_min = 2 ** exponent_width - 1 - exponent_bias, -exponent_bias
    sign = torch.sign(blocked + 1e-9)
    value = torch.abs(blocked) + 1e-9
    e = torch.ceil(torch.log2(bmax)).clamp(e_min, e_max)
    shift = 2 ** mbits
    m = torch.round(value / 2 ** e * shift).clamp(0, shift - 1)
    q = sign * (2 ** e) * (m / shift)
    out = _from_blocks(q, meta)
    close = torch.isclose(x, torch.tensor([0.0], dtype=x.dtype))
    return (~close) * out + close * x


def _fill_zero_block_maxima(bmax: torch.Tensor) -> torch.Tensor:
    """all-zero blocks take the smallest non-zero block maximum of the whole tensor (1 when every block is zero):
    block_fp.py:54-58, block_minifloat.py:52-55, block_log.py:50-53"""
    if torch.all(bmax == 0):
        return torch.ones_like(bmax)
    bmax[bmax == 0] = bmax[bmax != 0].min()
    return bmax


def block_minifloat_quantize(x: torch.Tensor, width: int, exponent_width: int, exponent_bias_width: int,
                             block_size=(16,), skip_first_dim: bool = True) -> torch.Tensor:
    """reference block_minifloat.py:22-74 -> minifloat.py:134-196, the same torch ops in the same order: shared bias
    = clamp(floor(log2(block max)), 0, 2^ebw - 1); per element an IEEE-style minifloat with exponent range
    [-bias, 2^ew - 1 - bias], subnormal where the clamped exponent equals -bias, |x| <= 1e-8 passed through."""
    meta = block_meta(tuple(x.shape), block_size, skip_first_dim)
    blocked = _to_blocks(x, meta)
    bmax = _fill_zero_block_maxima(blocked.abs().max(dim=1, keepdim=True)[0])
    bias = torch.floor(torch.log2(bmax)).clamp(0, 2 ** exponent_bias_width - 1)
    mbits = width - exponent_width - 1
    e_max, e_min = 2 ** exponent_width - 1 - bias, -bias
    shift = 2 ** mbits
    sign = torch.sign(blocked + 1e-9)
    value = torch.abs(blocked)
    e = torch.maximum(torch.minimum(torch.floor(torch.log2(value + 1e-9)), e_max), e_min)
    mant = value / 2 ** e
    normal = ~torch.isclose(e, -bias)
    sm = normal * torch.round(mant * shift - shift).clamp(0, shift - 1) + (~normal) * torch.round(mant * shift / 2).clamp(0, shift - 1)
    mant = normal * (1.0 + sm / shift) + (~normal) * (sm / shift * 2)
    close = torch.isclose(value, torch.tensor([0.0], dtype=value.dtype))
    q = (~close) * (sign * (2 ** e) * mant) + close * blocked
    return _from_blocks(q, meta)


def block_log_quantize(x: torch.Tensor, width: int, exponent_bias_width: int, block_size=(16,),
                       skip_first_dim: bool = True) -> torch.Tensor:
    """reference block_log.py:23-69 -> log.py:22-56: shared bias = clamp(2^(w-1) - 1 - ceil(log2(block max)), 0,
    2^ebw - 1); per element sign(x + 0.1 * 2^-bias) * 2^clamp(round(log2(|x| + 0.1 * 2^-bias)), -bias, 2^(w-1) - 1 - bias)."""
    meta = block_meta(tuple(x.shape), block_size, skip_first_dim)
    blocked = _to_blocks(x, meta)
    bmax = _fill_zero_block_maxima(blocked.abs().max(dim=1, keepdim=True)[0])
    ebits = width - 1
    bias = (2 ** ebits - 1 - torch.ceil(torch.log2(bmax))).clamp(0, 2 ** exponent_bias_width - 1)
    e_max, e_min = 2 ** ebits - 1 - bias, -bias
    min_pos = 2 ** e_min
    sign = torch.sign(blocked + min_pos * 0.1)
    value = torch.abs(blocked) + min_pos * 0.1
    e = torch.maximum(torch.minimum(torch.round(torch.log2(value)), e_max), e_min)
    return _from_blocks(sign * (2 ** e), meta)


def linear_ptq_step(x: torch.Tensor, w_q: torch.Tensor, b_q, cfg: dict) -> torch.Tensor:
    """steady-state PTQ LinearBlockFP forward: quantise x, F.linear against already-quantised W"""
    xq = block_fp_quantize(x, cfg["data_in_width"], cfg["data_in_exponent_width"],
                           cfg["data_in_exponent_bias"], cfg["data_in_block_size"], True)
    return F.linear(xq, w_q, b_q)
