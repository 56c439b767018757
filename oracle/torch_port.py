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


def linear_ptq_step(x: torch.Tensor, w_q: torch.Tensor, b_q, cfg: dict) -> torch.Tensor:
    """steady-state PTQ LinearBlockFP forward: quantise x, F.linear against already-quantised W"""
    xq = block_fp_quantize(x, cfg["data_in_width"], cfg["data_in_exponent_width"],
                           cfg["data_in_exponent_bias"], cfg["data_in_block_size"], True)
    return F.linear(xq, w_q, b_q)
