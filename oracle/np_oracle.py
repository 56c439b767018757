"""CPU oracle for the block-quantised linear/matmul hot path (numpy).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  The product path is the HIP library behind
``include/mi355q.h``.

Parity status: PINNED.  Every function here is checked bit-for-bit against golden
vectors produced by importing the reference quantisers in the build container
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``; test: ``tests/test_oracle_golden.py``).

This is a restatement, not a copy: the reference works on fp32 torch tensors through
~20 elementwise torch ops per quantiser; here each quantiser is written as an explicit
*encode* step that yields the integers the format actually stores (signed mantissa,
shared exponent / shared bias, element exponent code) and a *decode* step that
rebuilds the fp32 fake-quantised value with the same fp32 operations, in the same
order, as the reference.  All arithmetic that the reference does in fp32 is done in
``np.float32`` here.

``log2`` convention: the reference calls ``torch.log2`` on fp32.  The oracle models it
as the correctly rounded fp32 logarithm (float64 ``log2`` rounded once to fp32).  The
decisions the quantisers take from it (ceil / floor / round-half-even) agree with
torch-CPU on every golden vector, including the boundary sets around 2**k and
2**(k+1/2) (tests/golden/edge_*.npz).

Reference map (all under /root/reference/src/llm_mixed_q/models/quantize/):
  blocking / padding rules ............ quantizers/utils.py:42-83, 86-104, 127-144, 161-183, 211-237, 261-284
  block_fp ............................ quantizers/block_fp.py:21-96
  block_minifloat -> minifloat_ieee ... quantizers/block_minifloat.py:22-74, quantizers/minifloat.py:134-196
  block_log -> log .................... quantizers/block_log.py:23-69, quantizers/log.py:22-56
  PTQ linear .......................... quantized_modules/linear.py:59-76
  matmul / bmm wrappers ............... quantized_functions/matmul.py:146-297
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Sequence

import numpy as np

F32 = np.float32
_EPS9 = F32(1e-9)      # the "+1e-9" of block_fp.py:69,71 and minifloat.py:171,176 (added in fp32)
_ATOL = F32(1e-8)      # torch.isclose(x, 0) default atol, evaluated in fp32 (block_fp.py:93)
_TENTH = F32(0.1)      # log.py:47-48: min_pos * 0.1, scalar cast to fp32


# ----------------------------------------------------------------------------------
# correctly rounded fp32 log2 and exact powers of two
# ----------------------------------------------------------------------------------
def log2_f32(v: np.ndarray) -> np.ndarray:
    """fp32 log2, correctly rounded (float64 log2 rounded once)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log2(np.asarray(v, dtype=np.float64)).astype(F32)


def pow2_f32(e: np.ndarray) -> np.ndarray:
    """2**e in fp32 for integer-valued e: exact, subnormal below -126, inf above 127
    (what ``2 ** tensor`` gives in the reference for integer-valued fp32 exponents)."""
    with np.errstate(over="ignore"):
        return np.ldexp(np.float64(1.0), np.asarray(e).astype(np.int64)).astype(F32)


# ----------------------------------------------------------------------------------
# blocking (utils.py block()/unblock())
# ----------------------------------------------------------------------------------
def _right_align_block(x_shape: Sequence[int], block_shape: Sequence[int]) -> list[int]:
    """utils.py:42-66: right-align, missing leading dims and oversize dims take the
    whole axis."""
    nd = len(x_shape)
    bs = list(block_shape)
    bs = bs[-nd:] if len(bs) >= nd else [-1] * (nd - len(bs)) + bs
    return [x_shape[i] if (bs[i] == -1 or bs[i] > x_shape[i]) else bs[i] for i in range(nd)]


def _padded_dim(n: int, b: int) -> int:
    """utils.py:69-83: pad on the right up to a multiple of the block."""
    if b == -1 or n < b:
        return n
    return int(math.ceil(n / b)) * b


@dataclass
class BlockMeta:
    x_shape: tuple          # original shape
    lead: int               # number of independent leading slices (1, N or B)
    rows: int               # padded rows of the 2-D plane that is tiled
    cols: int               # padded cols
    b0: int                 # block rows
    b1: int                 # block cols
    crop_rows: int
    crop_cols: int

    @property
    def n_blocks(self) -> int:
        return self.lead * (self.rows // self.b0) * (self.cols // self.b1)

    @property
    def block_elems(self) -> int:
        return self.b0 * self.b1


def block_meta(x_shape: Sequence[int], block_size, skip_first_dim: bool) -> BlockMeta:
    """Resolve the reference's five blocking cases to one description:
    ``lead`` independent planes of ``rows x cols`` tiled by ``b0 x b1`` blocks."""
    if isinstance(block_size, int):
        block_size = [block_size]
    block_size = [int(b) for b in block_size]
    x_shape = tuple(int(s) for s in x_shape)
    nd = len(x_shape)
    if nd == 1:
        if skip_first_dim:
            raise AssertionError("skip_first_dim must be False for bias to be blocked")
        (b,) = _right_align_block(x_shape, block_size)
        return BlockMeta(x_shape, 1, 1, _padded_dim(x_shape[0], b), 1, b, 1, x_shape[0])
    if nd == 2:
        if skip_first_dim:                      # utils.py:127-144 activation [N, C]
            b = _right_align_block([1, x_shape[1]], block_size)
            return BlockMeta(x_shape, x_shape[0], 1, _padded_dim(x_shape[1], b[1]), 1, b[1],
                             1, x_shape[1])
        b = _right_align_block(x_shape, block_size)   # utils.py:161-183 weight [O, K]
        return BlockMeta(x_shape, 1, _padded_dim(x_shape[0], b[0]), _padded_dim(x_shape[1], b[1]),
                         b[0], b[1], x_shape[0], x_shape[1])
    if nd == 3:
        if not skip_first_dim:
            raise NotImplementedError("block 3d weight is not supported.")
        b = _right_align_block([1, x_shape[1], x_shape[2]], block_size)   # utils.py:211-237
        return BlockMeta(x_shape, x_shape[0], _padded_dim(x_shape[1], b[1]),
                         _padded_dim(x_shape[2], b[2]), b[1], b[2], x_shape[1], x_shape[2])
    raise RuntimeError(f"Unsupported x.ndim = {nd}")


def to_blocks(x: np.ndarray, meta: BlockMeta) -> np.ndarray:
    """-> [n_blocks, b0*b1] fp32; block index = lead-major, then block-row, then block-col
    (for a weight [O,K] with [1,B] blocks: o*(K/B)+kb)."""
    x = np.asarray(x, dtype=F32)
    plane = x.reshape(meta.lead, meta.crop_rows, meta.crop_cols)
    pad = np.zeros((meta.lead, meta.rows, meta.cols), dtype=F32)
    pad[:, :meta.crop_rows, :meta.crop_cols] = plane
    t = pad.reshape(meta.lead, meta.rows // meta.b0, meta.b0, meta.cols // meta.b1, meta.b1)
    return np.ascontiguousarray(t.transpose(0, 1, 3, 2, 4)).reshape(meta.n_blocks, meta.block_elems)


def from_blocks(blocks: np.ndarray, meta: BlockMeta) -> np.ndarray:
    t = blocks.reshape(meta.lead, meta.rows // meta.b0, meta.cols // meta.b1, meta.b0, meta.b1)
    pad = t.transpose(0, 1, 3, 2, 4).reshape(meta.lead, meta.rows, meta.cols)
    return np.ascontiguousarray(pad[:, :meta.crop_rows, :meta.crop_cols]).reshape(meta.x_shape)


def filled_block_max(blocks: np.ndarray) -> np.ndarray:
    """abs-max per block; all-zero blocks take the smallest non-zero block max of the
    WHOLE tensor, or 1 when every block is zero (block_fp.py:54-58 and twins)."""
    bmax = np.abs(blocks).max(axis=1)
    nz = bmax != 0
    if not nz.any():
        return np.ones_like(bmax)
    if not nz.all():
        bmax = bmax.copy()
        bmax[~nz] = bmax[nz].min()
    return bmax


def _passthrough_mix(close: np.ndarray, q: np.ndarray, x: np.ndarray) -> np.ndarray:
    """(~close)*q + close*x in fp32, as block_fp.py:94 / minifloat.py:194 write it."""
    with np.errstate(invalid="ignore"):
        return (~close).astype(F32) * q + close.astype(F32) * x


# ----------------------------------------------------------------------------------
# block floating point
# ----------------------------------------------------------------------------------
@dataclass
class BfpCode:
    meta: BlockMeta
    mant: np.ndarray     # int32 [n_blocks, block_elems], sign * integer mantissa
    exp: np.ndarray      # int32 [n_blocks], shared exponent (unbiased)
    mbits: int
    blocks: np.ndarray   # fp32 [n_blocks, block_elems] the padded input, block order

    def dequant_blocks(self) -> np.ndarray:
        """sign * 2**e * (m / 2**mbits) with the reference's fp32 operation order."""
        s = np.sign(self.mant).astype(F32)
        p2e = pow2_f32(self.exp)[:, None]
        mq = np.abs(self.mant).astype(F32) / F32(2 ** self.mbits)
        with np.errstate(invalid="ignore", over="ignore"):
            return (s * p2e) * mq


def bfp_encode(x, width: int, exponent_width: int = 8, exponent_bias=None,
               block_size=(16,), skip_first_dim: bool = True) -> BfpCode:
    """block_fp.py:44-79.  mant = sign(x+1e-9) * clamp(rne((|x|+1e-9)/2**e * 2**mb), 0, 2**mb-1),
    e = clamp(ceil(log2(block max)), -bias, 2**ew-1-bias)."""
    x = np.asarray(x, dtype=F32)
    meta = block_meta(x.shape, block_size, skip_first_dim)
    blocks = to_blocks(x, meta)
    bmax = filled_block_max(blocks)
    mbits = int(width) - 1
    if exponent_bias in (None, "none", "None"):
        exponent_bias = 2 ** (int(exponent_width) - 1) - 1
    e_max = 2 ** int(exponent_width) - 1 - int(exponent_bias)
    e_min = -int(exponent_bias)
    e = np.clip(np.ceil(log2_f32(bmax)), e_min, e_max).astype(F32)
    sign = np.sign(blocks + _EPS9)
    value = np.abs(blocks) + _EPS9
    p2e = pow2_f32(e)[:, None]
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        m = np.clip(np.rint((value / p2e) * F32(2 ** mbits)), 0, 2 ** mbits - 1)
    mant = (sign * m).astype(np.int32)
    return BfpCode(meta, mant, e.astype(np.int32), mbits, blocks)


def block_fp_quantize(x, width: int = 12, exponent_width: int = 8, exponent_bias=None,
                      block_size=(16,), skip_first_dim: bool = True) -> np.ndarray:
    """Fake-quantised fp32 tensor, == reference ``block_fp_quantizer`` (block_fp.py:127-153)."""
    x = np.asarray(x, dtype=F32)
    code = bfp_encode(x, width, exponent_width, exponent_bias, block_size, skip_first_dim)
    q = from_blocks(code.dequant_blocks(), code.meta)
    return _passthrough_mix(np.abs(x) <= _ATOL, q, x)


# ----------------------------------------------------------------------------------
# block minifloat
# ----------------------------------------------------------------------------------
@dataclass
class BmCode:
    meta: BlockMeta
    sign: np.ndarray      # int32 [n_blocks, be] in {-1,0,1}
    exp: np.ndarray       # int32 [n_blocks, be] element exponent (unbiased, already clamped)
    mant: np.ndarray      # int32 [n_blocks, be] fraction integer in [0, 2**mb-1]
    normal: np.ndarray    # bool  [n_blocks, be]
    bias: np.ndarray      # int32 [n_blocks] shared exponent bias
    mbits: int
    blocks: np.ndarray

    def dequant_blocks(self) -> np.ndarray:
        shift = F32(2 ** self.mbits)
        sm = self.mant.astype(F32)
        frac = np.where(self.normal, F32(1.0) + sm / shift, (sm / shift) * F32(2.0)).astype(F32)
        with np.errstate(invalid="ignore", over="ignore"):
            return (self.sign.astype(F32) * pow2_f32(self.exp)) * frac


def bm_encode(x, width: int, exponent_width: int, exponent_bias_width: int,
              block_size=(16,), skip_first_dim: bool = False) -> BmCode:
    """block_minifloat.py:44-65 + minifloat.py:158-191 with a per-block bias."""
    x = np.asarray(x, dtype=F32)
    meta = block_meta(x.shape, block_size, skip_first_dim)
    blocks = to_blocks(x, meta)
    bmax = filled_block_max(blocks)
    mbits = int(width) - int(exponent_width) - 1
    bias = np.clip(np.floor(log2_f32(bmax)), 0, 2 ** int(exponent_bias_width) - 1).astype(F32)
    e_max = (F32(2 ** int(exponent_width) - 1) - bias)[:, None]
    e_min = (-bias)[:, None]
    sign = np.sign(blocks + _EPS9)
    value = np.abs(blocks)
    e = np.clip(np.floor(log2_f32(value + _EPS9)), e_min, e_max).astype(F32)
    shift = F32(2 ** mbits)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        mant = value / pow2_f32(e)
        normal = e != e_min                     # ~isclose(e, -bias) on integer-valued fp32
        sm_n = np.clip(np.rint(mant * shift - shift), 0, 2 ** mbits - 1)
        sm_s = np.clip(np.rint(mant * shift / F32(2.0)), 0, 2 ** mbits - 1)
    sm = np.where(normal, sm_n, sm_s)
    return BmCode(meta, sign.astype(np.int32), e.astype(np.int32), sm.astype(np.int32),
                  normal, bias.astype(np.int32), mbits, blocks)


def block_minifloat_quantize(x, width: int, exponent_width: int, exponent_bias_width: int,
                             block_size=(16,), skip_first_dim: bool = False) -> np.ndarray:
    """== reference ``block_minifloat_quantizer`` (block_minifloat.py:118-146)."""
    x = np.asarray(x, dtype=F32)
    code = bm_encode(x, width, exponent_width, exponent_bias_width, block_size, skip_first_dim)
    qb = code.dequant_blocks()
    # the |x|<=1e-8 pass-through happens on the blocked tensor (minifloat.py:193-194);
    # padded zeros are cropped afterwards, so doing it on x is equivalent.
    q = from_blocks(qb, code.meta)
    return _passthrough_mix(np.abs(x) <= _ATOL, q, x)


# ----------------------------------------------------------------------------------
# block logarithmic
# ----------------------------------------------------------------------------------
@dataclass
class BlCode:
    meta: BlockMeta
    sign: np.ndarray      # int32 [n_blocks, be]
    exp: np.ndarray       # int32 [n_blocks, be] element exponent (unbiased)
    bias: np.ndarray      # int32 [n_blocks]
    blocks: np.ndarray

    def dequant_blocks(self) -> np.ndarray:
        with np.errstate(invalid="ignore", over="ignore"):
            return self.sign.astype(F32) * pow2_f32(self.exp)


def bl_encode(x, width: int, exponent_bias_width: int, block_size=(16,),
              skip_first_dim: bool = False) -> BlCode:
    """block_log.py:43-60 + log.py:38-56 with a per-block bias."""
    x = np.asarray(x, dtype=F32)
    meta = block_meta(x.shape, block_size, skip_first_dim)
    blocks = to_blocks(x, meta)
    bmax = filled_block_max(blocks)
    ebits = int(width) - 1
    top = F32(2 ** ebits - 1)
    bias = np.clip(top - np.ceil(log2_f32(bmax)), 0, 2 ** int(exponent_bias_width) - 1).astype(F32)
    e_max = (top - bias)[:, None]
    e_min = (-bias)[:, None]
    eps = (pow2_f32(-bias) * _TENTH)[:, None]
    sign = np.sign(blocks + eps)
    value = np.abs(blocks) + eps
    e = np.clip(np.rint(log2_f32(value)), e_min, e_max)
    return BlCode(meta, sign.astype(np.int32), e.astype(np.int32), bias.astype(np.int32), blocks)


def block_log_quantize(x, width: int, exponent_bias_width: int = None, block_size=(16,),
                       skip_first_dim: bool = False) -> np.ndarray:
    """== reference ``block_log_quantizer`` (block_log.py:111-134).  No pass-through:
    zeros come out as +2**-bias."""
    code = bl_encode(x, width, exponent_bias_width, block_size, skip_first_dim)
    return from_blocks(code.dequant_blocks(), code.meta)


# ----------------------------------------------------------------------------------
# element-wise minifloats and log (minifloat.py:21-86, 134-196; log.py:22-56): the un-blocked siblings, one fixed bias
# ----------------------------------------------------------------------------------
def _default_bias(exponent_bias, exponent_bits: int) -> int:
    return 2 ** (int(exponent_bits) - 1) - 1 if exponent_bias in (None, "none", "None") else int(exponent_bias)


def minifloat_ieee_quantize(x, width: int, exponent_width: int, exponent_bias=None) -> np.ndarray:
    """== reference ``minifloat_ieee_quantizer`` (minifloat.py:134-196): implicit leading one, subnormals at the lowest
    exponent, saturating, |x| <= 1e-8 passed through."""
    x = np.asarray(x, dtype=F32)
    mbits = int(width) - int(exponent_width) - 1
    bias = _default_bias(exponent_bias, exponent_width)
    e_max, e_min = F32(2 ** int(exponent_width) - 1 - bias), F32(-bias)
    sign = np.sign(x + _EPS9)
    value = np.abs(x)
    e = np.clip(np.floor(log2_f32(value + _EPS9)), e_min, e_max).astype(F32)
    shift = F32(2 ** mbits)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        mant = value / pow2_f32(e)
        normal = e != e_min
        sm = np.where(normal, np.clip(np.rint(mant * shift - shift), 0, 2 ** mbits - 1),
                      np.clip(np.rint(mant * shift / F32(2.0)), 0, 2 ** mbits - 1)).astype(F32)
        frac = np.where(normal, F32(1.0) + sm / shift, (sm / shift) * F32(2.0)).astype(F32)
        q = (sign * pow2_f32(e)) * frac
    return _passthrough_mix(value <= _ATOL, q.astype(F32), x)


def minifloat_denorm_quantize(x, width: int, exponent_width: int, exponent_bias=None) -> np.ndarray:
    """== reference ``minifloat_denorm_quantizer`` (minifloat.py:21-86): no implicit leading one, per-element exponent
    ceil(log2(|x| + 1e-9)), mantissa |x| / 2^e (WITHOUT the 1e-9) rounded to mantissa_bits."""
    x = np.asarray(x, dtype=F32)
    mbits = int(width) - int(exponent_width) - 1
    bias = _default_bias(exponent_bias, exponent_width)
    e_max, e_min = F32(2 ** int(exponent_width) - 1 - bias), F32(-bias)
    sign = np.sign(x + _EPS9)
    value = np.abs(x)
    e = np.clip(np.ceil(log2_f32(value + _EPS9)), e_min, e_max).astype(F32)
    shift = F32(2 ** mbits)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        sm = np.clip(np.rint((value / pow2_f32(e)) * shift), 0, 2 ** mbits - 1).astype(F32)
        q = (sign * pow2_f32(e)) * (sm / shift)
    return _passthrough_mix(value <= _ATOL, q.astype(F32), x)


def log_quantize(x, width: int, exponent_bias=None) -> np.ndarray:
    """== reference ``log_quantizer`` (log.py:22-56): sign * 2^clamp(rint(log2(|x| + 0.1 * 2^-bias))); cannot
    represent 0, no pass-through."""
    x = np.asarray(x, dtype=F32)
    ebits = int(width) - 1
    bias = _default_bias(exponent_bias, ebits)
    e_max, e_min = F32(2 ** ebits - 1 - bias), F32(-bias)
    eps = F32(np.float64(2.0) ** (-bias) * 0.1)       # (python float arithmetic, then the tensor's dtype: log.py:45-48)
    sign = np.sign(x + eps)
    value = np.abs(x) + eps
    e = np.clip(np.rint(log2_f32(value)), e_min, e_max)
    with np.errstate(over="ignore"):
        return (sign * pow2_f32(e)).astype(F32)


# ----------------------------------------------------------------------------------
# integer fixed point (RoPE tables in every shipped TOML; integer.py:25-58)
# ----------------------------------------------------------------------------------
def integer_quantize(x, width: int, frac_width: int, is_signed: bool = True) -> np.ndarray:
    x = np.asarray(x, dtype=F32)
    lo, hi = (-(2 ** (width - 1)), 2 ** (width - 1) - 1) if is_signed else (0, 2 ** width - 1)
    scale = F32(2.0 ** frac_width)
    return (np.clip(np.rint(x * scale), lo, hi) / scale).astype(F32)


# ----------------------------------------------------------------------------------
# PTQ linear and matmul on top of the quantisers
# ----------------------------------------------------------------------------------
def bfp_linear_int(x, w, bias, cfg: dict) -> np.ndarray:
    """The contraction the HIP GEMM performs, in exact arithmetic: for x [M,K], w [N,K]
    blocked [1,B] along K,
        y[m,n] = sum_kb 2**(ex[m,kb]+ew[n,kb]-mbx-mbw) * sum_j ix*iw  (+ b_q[n])
    (identity with F.linear on the fake-quantised operands: SURVEY 8a A7).  Integer block
    dots, float64 scaling/accumulation, one rounding to fp32 at the end."""
    x = np.asarray(x, dtype=F32)
    w = np.asarray(w, dtype=F32)
    M, K = x.shape
    N = w.shape[0]
    cx = bfp_encode(x, cfg["data_in_width"], cfg["data_in_exponent_width"],
                    cfg["data_in_exponent_bias"], cfg["data_in_block_size"], True)
    cw = bfp_encode(w, cfg["weight_width"], cfg["weight_exponent_width"],
                    cfg["weight_exponent_bias"], cfg["weight_block_size"], False)
    B = cx.meta.b1
    assert cw.meta.b0 == 1 and cw.meta.b1 == B and cx.meta.cols == cw.meta.cols
    nb = cx.meta.cols // B
    ix = cx.mant.reshape(M, nb, B).astype(np.int64)
    iw = cw.mant.reshape(N, nb, B).astype(np.int64)
    ex = cx.exp.reshape(M, nb).astype(np.int64) - cx.mbits
    ew = cw.exp.reshape(N, nb).astype(np.int64) - cw.mbits
    y = np.zeros((M, N), dtype=np.float64)
    for kb in range(nb):
        d = ix[:, kb, :] @ iw[:, kb, :].T                      # exact int64
        y += np.ldexp(d.astype(np.float64), ex[:, kb][:, None] + ew[:, kb][None, :])
    if bias is not None:
        bq = block_fp_quantize(np.asarray(bias, dtype=F32), cfg["bias_width"],
                               cfg["bias_exponent_width"], cfg["bias_exponent_bias"],
                               cfg["bias_block_size"], False)
        y += bq.astype(np.float64)[None, :]
    return y.astype(F32)


def _quantizer_for(name: str):
    return {"block_fp": block_fp_quantize, "block_minifloat": block_minifloat_quantize,
            "block_log": block_log_quantize}[name]


def _entry_kwargs(cfg: dict, prefix: str) -> dict:
    name = cfg["name"]
    if name == "block_fp":
        return dict(width=cfg[f"{prefix}_width"], exponent_width=cfg[f"{prefix}_exponent_width"],
                    exponent_bias=cfg[f"{prefix}_exponent_bias"], block_size=cfg[f"{prefix}_block_size"])
    if name == "block_minifloat":
        return dict(width=cfg[f"{prefix}_width"], exponent_width=cfg[f"{prefix}_exponent_width"],
                    exponent_bias_width=cfg[f"{prefix}_exponent_bias_width"],
                    block_size=cfg[f"{prefix}_block_size"])
    if name == "block_log":
        return dict(width=cfg[f"{prefix}_width"],
                    exponent_bias_width=cfg[f"{prefix}_exponent_bias_width"],
                    block_size=cfg[f"{prefix}_block_size"])
    raise KeyError(name)


def linear_ptq(x, w, bias, cfg: dict):
    """linear.py:63-71 steady state: returns (y, w_q, b_q) with y = x_q @ w_q.T + b_q in fp32
    (float64 accumulate, rounded once: the order-free value the fp32 GEMMs approximate)."""
    q = _quantizer_for(cfg["name"])
    xq = q(x, **_entry_kwargs(cfg, "data_in"), skip_first_dim=True)
    wq = q(w, **_entry_kwargs(cfg, "weight"), skip_first_dim=False)
    bq = None if bias is None else q(bias, **_entry_kwargs(cfg, "bias"), skip_first_dim=False)
    y = xq.astype(np.float64).reshape(-1, xq.shape[-1]) @ wq.astype(np.float64).T
    y = y.reshape(*xq.shape[:-1], wq.shape[0])
    if bq is not None:
        y = y + bq.astype(np.float64)
    return y.astype(F32), wq, bq


def matmul_quantized(x, y, cfg: dict) -> np.ndarray:
    """matmul.py:146-297: flatten leading dims, quantise x with data_in_* and y with
    weight_* params, skip_first_dim = (ndim > 2); block_log leaves y untouched."""
    x = np.asarray(x, dtype=F32)
    y = np.asarray(y, dtype=F32)
    if cfg.get("bypass", False):
        return np.matmul(x.astype(np.float64), y.astype(np.float64)).astype(F32)
    name = cfg["name"]
    q = _quantizer_for(name)

    def run(t, prefix):
        flat = t.reshape((-1,) + t.shape[-2:]) if t.ndim > 2 else t
        out = q(flat, **_entry_kwargs(cfg, prefix), skip_first_dim=t.ndim > 2)
        return out.reshape(t.shape)

    xq = run(x, "data_in")
    yq = y if name == "block_log" else run(y, "weight")
    return np.matmul(xq.astype(np.float64), yq.astype(np.float64)).astype(F32)
