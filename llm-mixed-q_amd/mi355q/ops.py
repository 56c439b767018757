"""Tensor-level entry points over the C ABI: shape resolution on the host, pointers and sizes
to the library, current HIP stream.  torch is plumbing here (device memory, streams)."""
from __future__ import annotations

import math
import os
from typing import Sequence

import torch

from . import _lib

_WORKSPACES: dict = {}


# ---------------------------------------------------------------------------------------
# host logic: the reference's block-shape rules (quantizers/utils.py:42-83, 261-284)
# ---------------------------------------------------------------------------------------
def _fit_block(shape: Sequence[int], block: Sequence[int]) -> list[int]:
    nd = len(shape)
    blk = list(block)[-nd:] if len(block) >= nd else [-1] * (nd - len(block)) + list(block)
    return [shape[i] if blk[i] == -1 or blk[i] > shape[i] else blk[i] for i in range(nd)]


def resolve_blocking(shape: Sequence[int], block_size, skip_first_dim: bool):
    """-> (lead, rows, cols, b0, b1): `lead` planes of rows x cols tiled by b0 x b1 blocks.
    Raises exactly where the reference's block() does (utils.py:267-284)."""
    if isinstance(block_size, int):
        block_size = [block_size]
    block_size = [int(b) for b in block_size]
    shape = [int(s) for s in shape]
    nd = len(shape)
    if nd == 1:
        assert skip_first_dim is False, "skip_first_dim must be False for bias to be blocked"
        (b,) = _fit_block(shape, block_size)
        return 1, 1, shape[0], 1, b
    if nd == 2:
        if skip_first_dim:
            b = _fit_block([1, shape[1]], block_size)
            return 1, shape[0], shape[1], 1, b[1]
        b = _fit_block(shape, block_size)
        return 1, shape[0], shape[1], b[0], b[1]
    if nd == 3:
        if not skip_first_dim:
            raise NotImplementedError("block 3d weight is not supported.")
        b = _fit_block([1, shape[1], shape[2]], block_size)
        return shape[0], shape[1], shape[2], b[1], b[2]
    raise RuntimeError(f"Unsupported x.ndim = {nd}")


def n_blocks(lead: int, rows: int, cols: int, b0: int, b1: int) -> int:
    return lead * math.ceil(rows / b0) * math.ceil(cols / b1)


# ---------------------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------------------
def _require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"mi355q.{what}: tensor is on '{t.device}'. The block-quantised path runs only as HIP "
            "kernels on an MI355X; there is no CPU fallback.")
    if t.dtype != torch.float32:
        raise TypeError(f"mi355q.{what}: fp32 tensors only (got {t.dtype}); the reference's BASELINE "
                        "configs are fp32 (SURVEY 8a quirk 11)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device) -> int:
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)                 # (no Stream object: this sits on every call's path)
    return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """`with torch.cuda.device(d)` costs ~5 us even when d is current; most callers never leave device 0"""
    __slots__ = ("dev", "ctx")

    def __init__(self, dev):
        self.dev, self.ctx = dev, None

    def __enter__(self):
        if self.dev.index is not None and self.dev.index != torch.cuda.current_device():
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _workspace(device) -> torch.Tensor:
    key = (device.index, _stream_ptr(device))
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = torch.zeros(_lib.WORKSPACE_BYTES // 4, dtype=torch.int32, device=device)
        _WORKSPACES[key] = ws
    return ws


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


BIAS_DEFAULT = -(2 ** 31)        # MI355Q_BIAS_DEFAULT: the reference's exponent_bias=None


def _default_bias(exponent_bias) -> int:
    """None -> the ABI's "default" sentinel; anything else, negative values included, is passed literally
    (block_fp.py:61-62 uses a given bias as it is)"""
    return BIAS_DEFAULT if exponent_bias in (None, "none", "None") else int(exponent_bias)


# ---------------------------------------------------------------------------------------
# quantisers
# ---------------------------------------------------------------------------------------
def block_fp_quantize(x: torch.Tensor, width: int, exponent_width: int, exponent_bias, block_size,
                      skip_first_dim: bool, *, want_fake: bool = True, want_packed: bool = False,
                      fast_zero_blocks: bool = False):
    """Returns y, or (y_or_None, mant int8 like x, exp uint8 [n_blocks]) when want_packed."""
    _require_device(x, "block_fp_quantize")
    lead, rows, cols, b0, b1 = resolve_blocking(x.shape, block_size, skip_first_dim)
    xc = x.contiguous()
    y = torch.empty_like(xc) if want_fake else None
    mant = torch.empty(xc.shape, dtype=torch.int8, device=x.device) if want_packed else None
    exp = (torch.empty(n_blocks(lead, rows, cols, b0, b1), dtype=torch.uint8, device=x.device)
           if want_packed else None)
    lib = _lib.load_library()
    with _on_device(x.device):
        rc = lib.mi355q_block_fp_quantize(
            _ptr(xc), _ptr(y), _ptr(mant), _ptr(exp), lead, rows, cols, b0, b1, int(width),
            int(exponent_width), _default_bias(exponent_bias),
            _lib.ZERO_BLOCK_FAST if fast_zero_blocks else _lib.ZERO_BLOCK_EXACT,
            _ptr(_workspace(x.device)), _stream_ptr(x.device))
    _lib.check(rc, "mi355q_block_fp_quantize")
    return (y, mant, exp) if want_packed else y


def block_fp_quantize_bf16(x: torch.Tensor, width: int, exponent_width: int, exponent_bias, block_size,
                           skip_first_dim: bool) -> torch.Tensor:
    """block_fp fake-quantisation straight to bf16 (exact for width <= 9): the operand of a bf16 MFMA GEMM on quantised
    values.  Blocks that are row vectors tiling the last dim go through one kernel (4 B read + 2 B written per element);
    anything else through the fp32 quantiser and a cast."""
    _require_device(x, "block_fp_quantize_bf16")
    assert int(width) <= 9, "a block_fp value wider than 9 bits is not exact in bf16"
    lead, rows, cols, b0, b1 = resolve_blocking(x.shape, block_size, skip_first_dim)
    if b0 != 1 or cols % b1 != 0 or b1 % 4 != 0:
        return block_fp_quantize(x, width, exponent_width, exponent_bias, block_size, skip_first_dim).to(torch.bfloat16)
    xc = x.contiguous()
    y = torch.empty(xc.shape, dtype=torch.bfloat16, device=x.device)
    lib = _lib.load_library()
    with _on_device(x.device):
        rc = lib.mi355q_block_fp_quantize_bf16(_ptr(xc), _ptr(y), lead, rows, cols, b0, b1, int(width), int(exponent_width),
                                               _default_bias(exponent_bias), _ptr(_workspace(x.device)), _stream_ptr(x.device))
    _lib.check(rc, "mi355q_block_fp_quantize_bf16")
    return y


def block_minifloat_quantize_bf16(x: torch.Tensor, width: int, exponent_width: int, exponent_bias_width: int, block_size,
                                  skip_first_dim: bool) -> torch.Tensor:
    """block_minifloat fake-quantisation straight to bf16 (<= 7 mantissa bits: exact), row-vector blocks tiling the last dim
    through one kernel (4 B read + 2 B written per element), anything else through the fp32 quantiser and a cast"""
    _require_device(x, "block_minifloat_quantize_bf16")
    assert int(width) - int(exponent_width) - 1 <= 7, "a minifloat with more than 7 mantissa bits is not exact in bf16"
    lead, rows, cols, b0, b1 = resolve_blocking(x.shape, block_size, skip_first_dim)
    xc = x.contiguous()
    # the kernel's vector path (launch_format, csrc/mi355q_quant.hip) takes row blocks of 4, 8, ..., 256 values that tile the
    # last dim, 16-byte aligned; its generic path has no bf16 output (T = 12 with block [1,16] fits the block to
    # b1 = 12 -> b1 / 4 = 3 -> MI355Q_E_UNSUPPORTED)
    if b0 != 1 or cols % b1 != 0 or b1 % 4 != 0 or (b1 // 4) not in (1, 2, 4, 8, 16, 32, 64) or xc.data_ptr() % 16 != 0:
        return block_minifloat_quantize(x, width, exponent_width, exponent_bias_width, block_size, skip_first_dim).to(torch.bfloat16)
    y = torch.empty(xc.shape, dtype=torch.bfloat16, device=x.device)
    lib = _lib.load_library()
    with _on_device(x.device):
        rc = lib.mi355q_block_minifloat_quantize_bf16(_ptr(xc), _ptr(y), lead, rows, cols, b0, b1, int(width), int(exponent_width),
                                                      int(exponent_bias_width), _ptr(_workspace(x.device)), _stream_ptr(x.device))
    _lib.check(rc, "mi355q_block_minifloat_quantize_bf16")
    return y


def block_log_quantize_bf16(x: torch.Tensor, width: int, exponent_bias_width: int, block_size, skip_first_dim: bool) -> torch.Tensor:
    """block_log fake-quantisation straight to bf16 (signed powers of two: exact), [1,16] row blocks through one kernel (+ the
    zero-block fix-up when the fill guess misses), anything else through the fp32 quantiser and a cast"""
    _require_device(x, "block_log_quantize_bf16")
    lead, rows, cols, b0, b1 = resolve_blocking(x.shape, block_size, skip_first_dim)
    if b0 != 1 or b1 != 16 or cols % 16 != 0:
        return block_log_quantize(x, width, exponent_bias_width, block_size, skip_first_dim).to(torch.bfloat16)
    xc = x.contiguous()
    y = torch.empty(xc.shape, dtype=torch.bfloat16, device=x.device)
    lib = _lib.load_library()
    with _on_device(x.device):
        rc = lib.mi355q_block_log_quantize_bf16(_ptr(xc), _ptr(y), lead, rows, cols, b0, b1, int(width), int(exponent_bias_width),
                                                _ptr(_workspace(x.device)), _stream_ptr(x.device))
    _lib.check(rc, "mi355q_block_log_quantize_bf16")
    return y


class _StreamCache:
    """Per-(device, stream, shape ...) device buffers whose raw pointers the library's kernels are launched with.  Least-
    recently-used eviction PER STREAM (never wholesale), and none at all for a stream that has recorded a HIP graph: the
    captured kernels keep the pointers, so a buffer a graph may replay into must outlive the cache's appetite (the caching
    allocator would hand a dropped buffer to the next tensor).  Key layout: (device index, stream handle, ...)."""
    _pinned_streams: set = set()

    def __init__(self, per_stream: int):
        from collections import OrderedDict
        self.per_stream, self._d = per_stream, OrderedDict()

    @classmethod
    def pin_stream(cls, device_index: int, stream_handle: int):
        cls._pinned_streams.add((device_index, stream_handle))

    def get(self, key):
        v = self._d.get(key)
        if v is not None:
            self._d.move_to_end(key)
        return v

    def put(self, key, value):
        sk = key[:2]
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self._pinned_streams.add(sk)
        self._d[key] = value
        if sk not in self._pinned_streams:
            mine = [k for k in self._d if k[:2] == sk]
            for k in mine[:max(0, len(mine) - self.per_stream)]:
                del self._d[k]
        return value

    def __len__(self):
        return len(self._d)

    def __contains__(self, key):
        return key in self._d

    def keys(self):
        return self._d.keys()


_BF16_TILED_BUFFERS = _StreamCache(32)


PRE_NONE, PRE_RELU, PRE_SILU_MUL, PRE_RMSNORM, PRE_LAYERNORM = 0, 1, 2, 3, 4   # include/mi355q.h: steps folded into a quantiser
PRE_OPS = {None: PRE_NONE, "relu": PRE_RELU, "silu_mul": PRE_SILU_MUL, "rmsnorm": PRE_RMSNORM, "layernorm": PRE_LAYERNORM}


def _pre_args(x, pre):
    """(op code, second input, epsilon, third input) of a `pre` = None | ("relu", None) | ("silu_mul", other) |
    ("rmsnorm", weight, eps) | ("layernorm", weight, eps, bias) request on x [rows, K]"""
    if pre is None:
        return PRE_NONE, None, 0.0, None
    op, other = pre[0], pre[1]
    code = PRE_OPS[op]
    if code == PRE_SILU_MUL:
        assert other is not None and other.shape == x.shape and other.dtype == torch.float32 and other.device == x.device
        return code, other.contiguous(), 0.0, None
    if code in (PRE_RMSNORM, PRE_LAYERNORM):
        vec = lambda t: t is not None and t.shape == (x.shape[1],) and t.dtype == torch.float32 and t.device == x.device
        assert vec(other)
        bias = pre[3] if code == PRE_LAYERNORM and len(pre) > 3 else None
        assert bias is None or vec(bias)
        return code, other.detach().contiguous(), float(pre[2]), None if bias is None else bias.detach().contiguous()
    return code, None, 0.0, None


def bfp_tiled_bytes(rows: int, row_bytes: int) -> int:
    """bytes of a tiled operand of `rows` rows of `row_bytes` bytes each (mi355q_bfp_tiled_bytes: rows padded to whole tiles)"""
    return int(_lib.load_library().mi355q_bfp_tiled_bytes(int(rows), int(row_bytes)))


def block_fp_quantize_bf16_tiled(x: torch.Tensor, width: int, exponent_width: int, exponent_bias, *, out_fake: torch.Tensor = None,
                                 reuse: bool = True, pre=None, out: torch.Tensor = None) -> torch.Tensor:
    """x [rows, K] fp32 ([1,16] blocks along K) -> bf16 in the tile order of `bf16_gemm_tiled` (a flat int8 buffer of
    mi355q_bfp_tiled_bytes(rows, 2 K) bytes).  `out_fake`: also write the fp32 fake-quantised values there (may be x
    itself).  `reuse`: the buffer is shared by calls with the same shape on the same stream (activations; consume it
    before quantising again), else freshly allocated (weights).  `out`: the operand's bytes go there instead (a contiguous int8
    tensor of exactly that many bytes: a rank's segment of an all-gather buffer, sharded.py)."""
    _require_device(x, "block_fp_quantize_bf16_tiled")
    assert x.ndim == 2 and x.shape[1] % 32 == 0 and int(width) <= 9 and x.is_contiguous()
    rows, K = x.shape
    lib = _lib.load_library()
    nbytes = lib.mi355q_bfp_tiled_bytes(rows, 2 * K)
    if out is not None:
        assert out.dtype == torch.int8 and out.numel() == nbytes and out.is_contiguous() and out.device == x.device
        yt = out.view(-1)
    elif reuse:
        key = (x.device.index, _stream_ptr(x.device), rows, K)
        yt = _BF16_TILED_BUFFERS.get(key)
        if yt is None:
            yt = _BF16_TILED_BUFFERS.put(key, torch.empty(nbytes, dtype=torch.int8, device=x.device))
    else:
        yt = torch.empty(nbytes, dtype=torch.int8, device=x.device)
    # (`pre`: quantise relu(x) / silu(x) * other / LlamaRMSNorm(x) instead of x, one pass -- ("rmsnorm", weight, eps))
    pre_op, other, eps, third = _pre_args(x, pre)
    assert pre_op != PRE_LAYERNORM and third is None, "the bf16 tiled quantiser applies relu, silu_mul and rmsnorm"
    with _on_device(x.device):
        rc = lib.mi355q_block_fp_quantize_bf16_tiled_norm(_ptr(x), _ptr(other), pre_op, eps, _ptr(out_fake), _ptr(yt), rows, K,
                                                          int(width), int(exponent_width), _default_bias(exponent_bias),
                                                          _ptr(_workspace(x.device)), _stream_ptr(x.device))
    _lib.check(rc, "mi355q_block_fp_quantize_bf16_tiled_norm")
    if out_fake is not None:
        _wrote_into(out_fake)
    return yt


def block_minifloat_quantize_bf16_tiled(x: torch.Tensor, width: int, exponent_width: int, exponent_bias_width: int) -> torch.Tensor:
    """x [rows, K] fp32 -> the block_minifloat values ([1,16] blocks along K) as tiled bf16, the operand of `bf16_gemm_tiled`
    (exact for <= 7 mantissa bits); the buffer is shared by calls of the same shape on the same stream, like the block_fp
    one's (consume it before quantising again)."""
    _require_device(x, "block_minifloat_quantize_bf16_tiled")
    assert x.ndim == 2 and x.shape[1] % 32 == 0 and x.is_contiguous() and int(width) - int(exponent_width) - 1 <= 7
    rows, K = x.shape
    lib = _lib.load_library()
    key = (x.device.index, _stream_ptr(x.device), rows, K)
    yt = _BF16_TILED_BUFFERS.get(key)
    if yt is None:
        yt = _BF16_TILED_BUFFERS.put(key, torch.empty(lib.mi355q_bfp_tiled_bytes(rows, 2 * K), dtype=torch.int8, device=x.device))
    with _on_device(x.device):
        rc = lib.mi355q_block_minifloat_quantize_bf16_tiled(_ptr(x), _ptr(yt), rows, K, int(width), int(exponent_width),
                                                            int(exponent_bias_width), _ptr(_workspace(x.device)),
                                                            _stream_ptr(x.device))
    _lib.check(rc, "mi355q_block_minifloat_quantize_bf16_tiled")
    return yt


# ---------------------------------------------------------------------------------------
# true width-bit weight storage (include/mi355q.h: mi355q_bfp_pack_bits / mi355q_bfp_expand)
# ---------------------------------------------------------------------------------------
def bfp_pack_bits(mant: torch.Tensor, width: int) -> torch.Tensor:
    """canonical int8 mantissas [rows, K] -> [rows, K * width / 8] uint8 (dense width-bit two's complement)"""
    assert mant.is_cuda and mant.dtype == torch.int8 and mant.ndim == 2 and mant.is_contiguous() and mant.shape[1] % 16 == 0
    rows, K = mant.shape
    out = torch.empty(rows, K * int(width) // 8, dtype=torch.uint8, device=mant.device)
    with _on_device(mant.device):
        rc = _lib.load_library().mi355q_bfp_pack_bits(_ptr(mant), _ptr(out), rows, K, int(width), _stream_ptr(mant.device))
    _lib.check(rc, "mi355q_bfp_pack_bits")
    return out


_EXPAND_SCRATCH: dict = {}


def _expand_scratch(device, nbytes, tag):
    key = (device.index, _stream_ptr(device), tag)
    buf = _EXPAND_SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _EXPAND_SCRATCH[key] = torch.empty(nbytes, dtype=torch.int8, device=device)
    return buf


class PackedWeights:
    """A weight operand at rest: width-bit mantissas + one code byte per block (width + 0.5 bits per value) and the few
    per-row words of its flavour; `expand()` streams it into a scratch operand shared by every layer on the stream
    (consume it -- run the GEMM -- before the next layer expands into the same `slot`; the members of a grouped launch take
    slots 0, 1, 2)."""

    def __init__(self, packed, codes, rows, K, width, exp_bias, *, rowflag=None, rowscale=None, rowexp=None, sparse=None):
        self.packed, self.codes = packed, codes
        self.rows, self.K, self.width, self.exp_bias = int(rows), int(K), int(width), int(exp_bias)
        self.rowflag, self.rowscale, self.rowexp, self.sparse = rowflag, rowscale, rowexp, sparse
        self.row_scale_flavour = rowscale is not None

    def bits_per_value(self) -> float:
        n = self.packed.numel() + self.codes.numel()
        for t in (self.rowflag, self.rowscale, self.rowexp, self.sparse):
            n += 0 if t is None else t.numel() * t.element_size()
        return 8.0 * n / (self.rows * self.K)

    def expand(self, slot: int = 0):
        lib = _lib.load_library()
        dev = self.packed.device
        off = self.exp_bias + self.width - 1
        if self.row_scale_flavour:                      # int8 row-scale operand
            tiled = _expand_scratch(dev, lib.mi355q_bfp_tiled_bytes(self.rows, self.K), f"i8{slot or ''}")
            exp = _expand_scratch(dev, self.rows * (self.K // 16), f"exp{slot or ''}").view(torch.uint8)
            with _on_device(dev):
                rc = lib.mi355q_bfp_expand(_ptr(self.packed), _ptr(self.codes), _ptr(tiled), self.rows, self.K, self.width, 0, off,
                                           _ptr(self.rowexp), _ptr(exp), _stream_ptr(dev))
            _lib.check(rc, "mi355q_bfp_expand")
            return AlignedOperand(self.rows, self.K, None, tiled, exp, self.rowflag, self.rowscale, self.sparse,
                                  self.width - 1, self.exp_bias, row_aligned=True, bucket_cap=0)
        tiled = _expand_scratch(dev, lib.mi355q_bfp_tiled_bytes(self.rows, 2 * self.K), "bf16")
        with _on_device(dev):
            rc = lib.mi355q_bfp_expand(_ptr(self.packed), _ptr(self.codes), _ptr(tiled), self.rows, self.K, self.width, 1, off,
                                       0, 0, _stream_ptr(dev))
        _lib.check(rc, "mi355q_bfp_expand")
        return tiled


def pack_row_aligned_weights(wm: torch.Tensor, we: torch.Tensor, wa: "AlignedOperand", width: int, exp_bias: int) -> PackedWeights:
    """the at-rest form of a row-aligned weight operand (one-off, at pack time): packed canonical mantissas; per block its
    left shift onto the row's exponent, 0xFF for the exception blocks (listed in wa.sparse), 0 for all-zero blocks"""
    rows, K = wm.shape
    nkb = K // 16
    eo = wa.exp.view(rows, nkb).to(torch.int16)
    shift = we.view(rows, nkb).to(torch.int16) - eo
    zero_blk = (wm.view(rows, nkb, 16) == 0).all(-1)
    shift[zero_blk] = 0
    _, ent = row_list_entries(wa.sparse, rows, wa.list_cap)
    if len(ent):
        e = torch.from_numpy(ent[:, :2].astype("int64")).to(wm.device)
        shift[e[:, 0], e[:, 1]] = 255
    assert int(((shift < 0) | ((shift > 7) & (shift != 255))).sum()) == 0, "row-aligned operand with a block outside its window"
    rowexp = eo[:, 0].to(torch.uint8).contiguous()
    return PackedWeights(bfp_pack_bits(wm, width), shift.to(torch.uint8).contiguous(), rows, K, width, exp_bias,
                         rowflag=wa.rowflag, rowscale=wa.gscale, rowexp=rowexp, sparse=wa.sparse)


def pack_block_exponent_weights(wm: torch.Tensor, we: torch.Tensor, width: int, exp_bias: int) -> PackedWeights:
    """the at-rest form of a per-block-exponent (bf16 flavour) weight operand: packed mantissas + biased exponents"""
    rows, K = wm.shape
    return PackedWeights(bfp_pack_bits(wm, width), we.contiguous().view(rows, K // 16), rows, K, width, exp_bias)


def bf16_tile(x: torch.Tensor) -> torch.Tensor:
    """already-quantised fp32 values [rows, K] -> tiled bf16 (a cast into the tile order; exact for widths <= 9)"""
    _require_device(x, "bf16_tile")
    assert x.ndim == 2 and x.shape[1] % 32 == 0 and x.is_contiguous()
    rows, K = x.shape
    lib = _lib.load_library()
    yt = torch.empty(lib.mi355q_bfp_tiled_bytes(rows, 2 * K), dtype=torch.int8, device=x.device)
    with _on_device(x.device):
        rc = lib.mi355q_bf16_tile(_ptr(x), _ptr(yt), rows, K, _stream_ptr(x.device))
    _lib.check(rc, "mi355q_bf16_tile")
    return yt


def fp32_split_tile(x: torch.Tensor, role: int) -> torch.Tensor:
    """fp32 [rows, K] -> the tiled bf16 operand [rows, 6 K] of an fp32-equivalent product: three bf16 parts per value, the six column
    segments of mi355q_split.hip (role 0 = left operand / activations, 1 = right operand / weights).  `fp32_gemm_split` multiplies two of them."""
    _require_device(x, "fp32_split_tile")
    assert x.ndim == 2 and x.dtype == torch.float32 and x.shape[1] % 32 == 0 and x.is_contiguous() and role in (0, 1)
    rows, K = x.shape
    lib = _lib.load_library()
    yt = torch.empty(lib.mi355q_bfp_tiled_bytes(rows, 12 * K), dtype=torch.int8, device=x.device)
    with _on_device(x.device):
        rc = lib.mi355q_fp32_split_tile(_ptr(x), _ptr(yt), rows, K, int(role), _stream_ptr(x.device))
    _lib.check(rc, "mi355q_fp32_split_tile")
    return yt


def fp32_gemm_split(x6: torch.Tensor, w6: torch.Tensor, M: int, N: int, K: int, bias=None, out: torch.Tensor = None):
    """y[M, N] = x . w^T (+ bias) of two fp32 matrices given as their split operands (fp32_split_tile, roles 0 and 1): one launch of the bf16
    tile GEMM over 6 K, fp32 accumulation -- the unquantised lm_head (modeling_llama.py:866) on the MFMA instead of a vendor fp32 GEMM."""
    return bf16_gemm_tiled(x6, w6, M, N, 6 * K, bias=bias, out=out)


def bf16_gemm_tiled(xt: torch.Tensor, wt: torch.Tensor, M: int, N: int, K: int, bias=None, out: torch.Tensor = None,
                    segments: int = 1, residual: torch.Tensor = None):
    """y[M, N] = x . w^T (+ bias) on tiled bf16 operands (block_fp_quantize_bf16_tiled), fp32 accumulation and output:
    the tile GEMM's bf16 arithmetic -- operands whose blocks keep their own exponents.  `segments` > 1: xt is [segments,
    bytes] -- column segment s of x as its own tiled operand, rank-major as an all-gather leaves them (sharded.py).
    `residual` [M, N] fp32: y = (x . w^T + bias) + residual, the decoder layer's residual add in the store (the same bits as
    the separate add; mi355q_bf16_gemm_tiled_res)."""
    if not (xt.is_cuda and wt.is_cuda):
        raise RuntimeError("mi355q.bf16_gemm_tiled: operands must be on a HIP device; there is no CPU fallback")
    given = out is not None
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=xt.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    ldy = out.stride(0) if M > 1 else max(N, out.stride(0))
    with _on_device(xt.device):
        if segments > 1:
            assert xt.ndim == 2 and xt.shape[0] == segments and xt.is_contiguous()
            # (each segment is a whole tiled operand of M rows x K / segments values: ADVICE r4)
            assert xt.shape[1] * xt.element_size() == _lib.load_library().mi355q_bfp_tiled_bytes(M, 2 * K // segments), (xt.shape, M, K, segments)
            rc = _lib.load_library().mi355q_bf16_gemm_tiled_seg(_ptr(xt), _ptr(wt), _ptr(bias), _ptr(out), M, N, K, ldy, int(segments),
                                                               xt.stride(0) * xt.element_size(), _stream_ptr(xt.device))
        elif residual is not None:
            assert residual.dtype == torch.float32 and residual.shape == (M, N) and residual.stride(1) == 1 and residual.device == out.device
            rc = _lib.load_library().mi355q_bf16_gemm_tiled_res(_ptr(xt), _ptr(wt), _ptr(bias), _ptr(residual),
                                                               residual.stride(0) if M > 1 else max(N, residual.stride(0)),
                                                               _ptr(out), M, N, K, ldy, _stream_ptr(xt.device))
        else:
            rc = _lib.load_library().mi355q_bf16_gemm_tiled(_ptr(xt), _ptr(wt), _ptr(bias), _ptr(out), M, N, K, ldy,
                                                           _stream_ptr(xt.device))
    _lib.check(rc, "mi355q_bf16_gemm_tiled_seg" if segments > 1 else "mi355q_bf16_gemm_tiled")
    if given:
        _wrote_into(out)
    return out


def block_minifloat_quantize(x: torch.Tensor, width: int, exponent_width: int, exponent_bias_width: int,
                             block_size, skip_first_dim: bool, *, want_bias: bool = False):
    _require_device(x, "block_minifloat_quantize")
    lead, rows, cols, b0, b1 = resolve_blocking(x.shape, block_size, skip_first_dim)
    xc = x.contiguous()
    y = torch.empty_like(xc)
    bias = (torch.empty(n_blocks(lead, rows, cols, b0, b1), dtype=torch.uint8, device=x.device)
            if want_bias else None)
    lib = _lib.load_library()
    with _on_device(x.device):
        rc = lib.mi355q_block_minifloat_quantize(
            _ptr(xc), _ptr(y), _ptr(bias), lead, rows, cols, b0, b1, int(width), int(exponent_width),
            int(exponent_bias_width), _lib.ZERO_BLOCK_EXACT, _ptr(_workspace(x.device)),
            _stream_ptr(x.device))
    _lib.check(rc, "mi355q_block_minifloat_quantize")
    return (y, bias) if want_bias else y


def block_log_quantize(x: torch.Tensor, width: int, exponent_bias_width: int, block_size,
                       skip_first_dim: bool, *, want_bias: bool = False):
    _require_device(x, "block_log_quantize")
    lead, rows, cols, b0, b1 = resolve_blocking(x.shape, block_size, skip_first_dim)
    xc = x.contiguous()
    y = torch.empty_like(xc)
    bias = (torch.empty(n_blocks(lead, rows, cols, b0, b1), dtype=torch.uint8, device=x.device)
            if want_bias else None)
    lib = _lib.load_library()
    with _on_device(x.device):
        rc = lib.mi355q_block_log_quantize(
            _ptr(xc), _ptr(y), _ptr(bias), lead, rows, cols, b0, b1, int(width),
            int(exponent_bias_width), _lib.ZERO_BLOCK_EXACT, _ptr(_workspace(x.device)),
            _stream_ptr(x.device))
    _lib.check(rc, "mi355q_block_log_quantize")
    return (y, bias) if want_bias else y


def minifloat_quantize(x: torch.Tensor, width: int, exponent_width: int, exponent_bias=None, *, denorm: bool = False) -> torch.Tensor:
    """minifloat_ieee_quantizer / minifloat_denorm_quantizer of the reference (minifloat.py:134-196 / 21-86), element-wise"""
    _require_device(x, "minifloat_quantize")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    with _on_device(x.device):
        rc = _lib.load_library().mi355q_minifloat_quantize(_ptr(xc), _ptr(y), xc.numel(), int(width), int(exponent_width),
                                                           _default_bias(exponent_bias), int(bool(denorm)), _stream_ptr(x.device))
    _lib.check(rc, "mi355q_minifloat_quantize")
    return y


def log_quantize(x: torch.Tensor, width: int, exponent_bias=None) -> torch.Tensor:
    """log_quantizer of the reference (log.py:22-56), element-wise"""
    _require_device(x, "log_quantize")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    with _on_device(x.device):
        rc = _lib.load_library().mi355q_log_quantize(_ptr(xc), _ptr(y), xc.numel(), int(width), _default_bias(exponent_bias),
                                                     _stream_ptr(x.device))
    _lib.check(rc, "mi355q_log_quantize")
    return y


def integer_quantize(x: torch.Tensor, width: int, frac_width: int, is_signed: bool = True) -> torch.Tensor:
    _require_device(x, "integer_quantize")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    lib = _lib.load_library()
    with _on_device(x.device):
        rc = lib.mi355q_integer_quantize(_ptr(xc), _ptr(y), xc.numel(), int(width), int(frac_width),
                                         int(bool(is_signed)), _stream_ptr(x.device))
    _lib.check(rc, "mi355q_integer_quantize")
    return y


# ---------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------
def bfp_gemm(xm: torch.Tensor, xe: torch.Tensor, wm: torch.Tensor, we: torch.Tensor, bias,
             x_mbits: int, x_exp_bias: int, w_mbits: int, w_exp_bias: int, out: torch.Tensor = None):
    """y[M,N] = block-fp product of packed x [M,K] and packed w [N,K] (+ bias).  `out` may be a
    column slice view of a wider row-major buffer (its stride(0) becomes ldy)."""
    if not (xm.is_cuda and wm.is_cuda):
        raise RuntimeError("mi355q.bfp_gemm: operands must be on a HIP device; there is no CPU fallback")
    M, K = xm.shape
    N = wm.shape[0]
    assert wm.shape[1] == K and xm.dtype == torch.int8 and wm.dtype == torch.int8
    assert xe.numel() == M * (K // 16) and we.numel() == N * (K // 16)
    assert xm.is_contiguous() and wm.is_contiguous() and xe.is_contiguous() and we.is_contiguous()
    given = out is not None
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=xm.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    ldy = out.stride(0) if M > 1 else max(N, out.stride(0))
    lib = _lib.load_library()
    with _on_device(xm.device):
        rc = lib.mi355q_bfp_gemm(_ptr(xm), _ptr(xe), _ptr(wm), _ptr(we), _ptr(bias), _ptr(out), M, N, K, ldy,
                                 int(x_mbits), int(x_exp_bias), int(w_mbits), int(w_exp_bias),
                                 _stream_ptr(xm.device))
    _lib.check(rc, "mi355q_bfp_gemm")
    if given:
        _wrote_into(out)
    return out


class AlignedOperand:
    """A packed block-fp operand rewritten for the row-scale GEMM (include/mi355q.h, mi355q_bfp_align_rows):
    exponent-aligned tiled mantissas, effective exponents, per-row flags and fp32 row scales, and the bucketed list of
    exception blocks (blocks outside their row's exponent window, kept aside exactly)."""

    def __init__(self, rows, K, mant, tiled, exp, rowflag, gscale, sparse, mbits, exp_bias, row_aligned=True,
                 bucket_cap=None):
        assert row_aligned, "the 256-value-group flavour was removed in round 5"
        self.rows, self.K = int(rows), int(K)
        self.mant, self.tiled = mant, tiled          # row-major (may be None) / tiled (what the GEMM reads)
        self.exp, self.rowflag, self.gscale, self.sparse = exp, rowflag, gscale, sparse
        self.mbits, self.exp_bias = int(mbits), int(exp_bias)
        self.row_aligned = True                      # one exponent per ROW (mi355q_bfp_align_rows)
        # entries the list holds per 256-row bucket
        self.unaligned = bucket_cap is not None and int(bucket_cap) < 0   # row format, own exponents
        self.list_cap = int(bucket_cap) if bucket_cap and int(bucket_cap) > 0 else ROW_BUCKET_CAP

    def c_struct(self):
        cs = getattr(self, "_cs", None)
        if cs is None:                                   # (operands are immutable once built)
            cs = self._cs = _lib.BfpOperand(_ptr(self.tiled), _ptr(self.exp), _ptr(self.rowflag), _ptr(self.gscale),
                                            _ptr(self.sparse), self.list_cap, self.mbits, self.exp_bias,
                                            2 if self.unaligned else int(self.row_aligned))
            self._cs_addr = __import__("ctypes").addressof(cs)
        return cs


ROW_ALIGN_MAX_K = 16384
ROW_BUCKET_ROWS, ROW_BUCKET_CAP, ROW_BUCKET_CAP_MAX, ROW_NO_ALIGN = 256, 120, 1016, -1
# exception entries per 256 rows of a fused-quantised ACTIVATION operand.  120: the GEMM adds them from LDS (at most
# ~96 per tile together with the weights'); larger: the row post-pass adds them after the GEMM, without a tile limit.
import os as _os
ACTIVATION_BUCKET_CAP = int(_os.environ.get("MI355Q_X_BUCKET_CAP", ROW_BUCKET_CAP))


def row_align_supported(K: int) -> bool:
    """the row-aligned fast GEMM needs K % 128 == 0 and K <= MI355Q_ROW_ALIGN_MAX_K"""
    return K % 128 == 0 and 0 < K <= ROW_ALIGN_MAX_K


def _new_row_list(device, rows, bucket_cap=0):
    n = _lib.load_library().mi355q_bfp_row_list_bytes(rows, int(bucket_cap)) // 4
    return torch.zeros(n, dtype=torch.int32, device=device)


ROW_TILE_ENTRIES_FAST = 48      # entries of one tile (x bucket + w bucket) the row-scale GEMM adds back from spare LDS
ROW_TILE_ENTRIES_SLOW = 96      # ... from the stage area behind the K loop (gathers exposed: +10-30 us per launch)


def gemm_tile_rows(M: int, N: int) -> int:
    """rows of the tile the row-scale GEMM takes for an M x N output (the launch rule of mi355q_gemm_v8.hip: 256 x 256
    unless 128 x 256 fills the 256 compute units better).  A 128-row tile sees about half of a 256-row bucket's entries."""
    tn = -(-N // 256)
    t256, t128 = -(-M // 256) * tn, -(-M // 128) * tn
    return 128 if -(-t128 // 256) * 0.82 < -(-t256 // 256) * 1.0 else 256


def gemm_rounds(M: int, N: int, count: int = 1) -> float:
    """rounds over the 256 compute units (in units of a 256 x 256 tile's time) that `count` M x N outputs of one launch of
    the row-scale GEMM cost by its launch rule: whole rounds of 256-row tiles, or of 128-row tiles at 0.82 each"""
    tn = -(-N // 256) * count
    t256, t128 = -(-M // 256) * tn, -(-M // 128) * tn
    return min(-(-t256 // 256) * 1.0, -(-t128 // 256) * 0.82)


def grouped_launch_plan(M: int, N: int, count: int):
    """how to spread `count` (2 or 3) equally shaped products over launches of the grouped tile GEMM: the split with the
    fewest rounds over the chip, 0.1 round charged per launch.  2048 x 4096 x 3 (Llama-7B q / k / v): one launch is 384
    tiles = two rounds with the second half empty; (2, 1) is one full round of 256-row tiles + one of 128-row tiles."""
    plans = {2: [(2,), (1, 1)], 3: [(3,), (2, 1), (1, 1, 1)]}.get(count, [(count,)])
    return min(plans, key=lambda p: sum(gemm_rounds(M, N, g) + 0.1 for g in p))


def row_list_fill(lst, rows, bucket_cap=None):
    """(rows that overflowed their bucket, entries in the fullest bucket) of a row-aligned operand's list (host read)"""
    nb = (rows + ROW_BUCKET_ROWS - 1) // ROW_BUCKET_ROWS
    words = 8 + 8 * (bucket_cap or ROW_BUCKET_CAP)
    head = lst[: 8 + nb * words].detach().cpu()
    counts = head[8::words][:nb]
    return int(head[0]), int(counts.max()) if nb else 0


def row_list_entries(lst, rows, bucket_cap=None):
    """decode a row-aligned operand's bucketed exception list -> (overflowed rows, [n, 8] int32 entries)"""
    import numpy as np
    lst = lst.detach().cpu().numpy() if hasattr(lst, "detach") else np.asarray(lst)
    cap = bucket_cap or ROW_BUCKET_CAP
    words = 8 + 8 * cap
    out = []
    for b in range((rows + ROW_BUCKET_ROWS - 1) // ROW_BUCKET_ROWS):
        bk = lst[8 + b * words: 8 + (b + 1) * words]
        n = min(int(bk[0]), cap)
        ent = bk[8:8 + 8 * n].reshape(n, 8)
        out.append(ent[ent[:, 0] >= 0])
    return int(lst[0]), (np.concatenate(out) if out else np.zeros((0, 8), np.int32))


def bfp_align_rows(mant: torch.Tensor, exp: torch.Tensor, mbits: int, exp_bias: int, with_list: bool = True,
                   bucket_cap: int = 0) -> AlignedOperand:
    """Rewrite a packed [rows, K] operand into the ROW-aligned, tiled format (one exponent and one fp32 scale
    per row; include/mi355q.h, mi355q_bfp_align_rows).  K % 64 == 0, K <= ROW_ALIGN_MAX_K."""
    if not mant.is_cuda:
        raise RuntimeError("mi355q.bfp_align_rows: operands must be on a HIP device; there is no CPU fallback")
    rows, K = mant.shape
    assert mant.dtype == torch.int8 and exp.dtype == torch.uint8 and mant.is_contiguous() and exp.is_contiguous()
    assert exp.numel() == rows * (K // 16)
    lib = _lib.load_library()
    eo = torch.empty_like(exp)
    tiled = torch.zeros(lib.mi355q_bfp_tiled_bytes(rows, K), dtype=torch.int8, device=mant.device)
    flag = torch.empty(rows, dtype=torch.uint8, device=mant.device)
    rscale = torch.zeros(lib.mi355q_bfp_rows_pad(rows), dtype=torch.float32, device=mant.device)
    sparse = _new_row_list(mant.device, rows, bucket_cap) if with_list else None
    with _on_device(mant.device):
        rc = lib.mi355q_bfp_align_rows(_ptr(mant), _ptr(exp), _ptr(tiled), _ptr(eo), _ptr(flag), _ptr(rscale),
                                       _ptr(sparse), int(exp_bias) + int(mbits), rows, K, int(bucket_cap),
                                       _stream_ptr(mant.device))
    _lib.check(rc, "mi355q_bfp_align_rows")
    return AlignedOperand(rows, K, None, tiled, eo, flag, rscale, sparse, mbits, exp_bias, row_aligned=True,
                          bucket_cap=bucket_cap)


# ---- W4A4 / W5A5 on the MX scaled matrix instruction (include/mi355q.h, "MX scaled matrix instruction"; csrc/mi355q_mx.hip)
class MxOperand:
    """block_fp values of <= 5 bits as FP6 e2m3 codes + one E8M0 scale per 32 values, in the MX product's tile order.  `bad`:
    one int32 word the quantiser raises when some 32-group's two blocks lie too far apart for one scale (the product launch
    then forms the exact product from the fp32 tensors: slow -- callers read the word now and then and leave the route)"""
    __slots__ = ("rows", "K", "c16", "c8", "sc", "bad", "width", "exponent_width", "exponent_bias", "source", "_wbad_of", "_buf")


_MX_BUFFERS = _StreamCache(16)


def mx_supported(K: int, x_width: int, w_width: int) -> bool:
    return K % 128 == 0 and 0 < K <= ROW_ALIGN_MAX_K and 2 <= int(x_width) <= 5 and 2 <= int(w_width) <= 5


def block_fp_quantize_mx(x: torch.Tensor, width: int, exponent_width: int, exponent_bias, reuse: bool = True,
                         keep_source: bool = True) -> MxOperand:
    """x [rows, K] fp32 ([1,16] blocks along K) -> MxOperand.  `reuse`: the planes are shared by calls of the same shape on the
    same stream (activations: consume before quantising again), else freshly allocated (weights).  The operand keeps `x`
    (`source`): the product launch reads it if a flag is raised -- an ACTIVATION operand's; a weights' operand is built with
    keep_source = False (mx_gemm is handed the fake-quantised weights themselves, and a kept alias of the fp32 storage would
    defeat release_fp32_weight())."""
    _require_device(x, "block_fp_quantize_mx")
    assert x.ndim == 2 and x.is_contiguous() and x.dtype == torch.float32 and x.shape[1] % 128 == 0 and 2 <= int(width) <= 5
    rows, K = x.shape
    lib = _lib.load_library()
    sp = _stream_ptr(x.device)
    key = (x.device.index, sp, "mx", rows, K)
    buf = _MX_BUFFERS.get(key) if reuse else None
    if buf is None:
        buf = dict(c16=torch.empty(lib.mi355q_mx_plane_bytes(rows, K, 0), dtype=torch.uint8, device=x.device),
                   c8=torch.empty(lib.mi355q_mx_plane_bytes(rows, K, 1), dtype=torch.uint8, device=x.device),
                   sc=torch.empty(lib.mi355q_mx_plane_bytes(rows, K, 2), dtype=torch.uint8, device=x.device),
                   bad=torch.zeros(8, dtype=torch.int32, device=x.device), calls=0)
        if reuse:
            _MX_BUFFERS.put(key, buf)
    op = MxOperand()
    # (flag words: call n raises word 2 (n & 1) and clears the other call's; word 2 (n & 1) + 1 holds the weights' flag for the product)
    par = buf["calls"] & 1
    buf["calls"] += 1
    op.rows, op.K, op.c16, op.c8, op.sc = rows, K, buf["c16"], buf["c8"], buf["sc"]
    op.bad, nxt = buf["bad"][2 * par:2 * par + 2], buf["bad"][2 * (1 - par):2 * (1 - par) + 1]
    op._wbad_of = buf.get("wbad_of")
    op._buf = buf
    op.width, op.exponent_width, op.exponent_bias, op.source = int(width), int(exponent_width), _default_bias(exponent_bias), (x if keep_source else None)
    with _on_device(x.device):
        rc = lib.mi355q_block_fp_quantize_mx(_ptr(x), _ptr(op.c16), _ptr(op.c8), _ptr(op.sc), _ptr(op.bad), _ptr(nxt) if not _capturing() else None, rows, K, int(width),
                                            int(exponent_width), op.exponent_bias, sp)
    _lib.check(rc, "mi355q_block_fp_quantize_mx")
    return op


def mx_gemm(x: MxOperand, w: MxOperand, w_fp32: torch.Tensor, bias=None, out: torch.Tensor = None) -> torch.Tensor:
    """y [M, N] = x_q . w_q^T (+ bias) on the MX scaled MFMA (fp32 accumulation); `w_fp32`: the fake-quantised weights [N, K]
    (read only if a flag word is raised).  x.bad / w.bad: word 0 of each."""
    M, K, N = x.rows, x.K, w.rows
    assert w.K == K and w_fp32.shape == (N, K) and w_fp32.is_contiguous() and w_fp32.dtype == torch.float32
    assert x.source is not None, "mx_gemm: the activation operand must keep its fp32 source (the exact in-launch fallback reads it)"
    given = out is not None
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.c16.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    ldy = out.stride(0) if M > 1 else max(N, out.stride(0))
    # the two flag words side by side: x's own word and a copy slot for w's (weights are static: copied once per pairing)
    if getattr(x, "_wbad_of", None) is not w:
        x._buf["bad"][1::2].copy_(w.bad[0:1].expand(4))         # (both parities' slots)
        x._buf["wbad_of"] = x._wbad_of = w
    lib = _lib.load_library()
    with _on_device(x.c16.device):
        rc = lib.mi355q_mx_gemm(_ptr(x.c16), _ptr(x.c8), _ptr(x.sc), _ptr(w.c16), _ptr(w.c8), _ptr(w.sc), _ptr(x.bad), _ptr(x.source),
                                _ptr(w_fp32), _ptr(bias), _ptr(out), M, N, K, ldy, x.width, x.exponent_width, x.exponent_bias,
                                _stream_ptr(x.c16.device))
    _lib.check(rc, "mi355q_mx_gemm")
    if given:
        _wrote_into(out)
    return out


# ---- vendor-library GEMM calls: the decline routes keep the reference's semantics through torch's own
#      GEMMs (F.linear, torch.matmul / bmm: bypass mode, QAT on fp32, shapes / arithmetics no kernel of this library takes).
#      Never the default route of a quantised layer -- this counter says how often a forward took them, and where.
VENDOR_GEMM_CALLS: dict = {}


def count_vendor_gemm(site: str) -> None:
    VENDOR_GEMM_CALLS[site] = VENDOR_GEMM_CALLS.get(site, 0) + 1


def vendor_gemm_calls(reset: bool = False) -> dict:
    """{site: calls} since the last reset; sites: "linear.bypass", "linear.ptq_fallthrough", "linear.qat_fp32",
    "matmul.bypass", "matmul.generic" (quantized_functions), and whatever a harness adds (the unquantised lm_head)"""
    out = dict(VENDOR_GEMM_CALLS)
    if reset:
        VENDOR_GEMM_CALLS.clear()
    return out


def _wrote_into(t: torch.Tensor) -> None:
    """a kernel of this library has just written into a caller-provided tensor through its raw pointer: move its version
    counter like an in-place torch op would (version-keyed caches -- the quantised-activation reuse below, autograd's
    saved-tensor checks -- must see the write)"""
    torch.autograd.graph.increment_version(t)


def _capturing() -> bool:
    """the current stream is recording a HIP graph (torch.cuda.graph): no host reads, no state that alternates per call"""
    return torch.cuda.is_current_stream_capturing()


class _ActivationBuffers:
    """Reusable device buffers of the fused activation path, keyed by (device, stream, rows, K).  Two
    exception lists alternate: each quantise call fills one and zeroes the other's count for the next call."""
    _cache = _StreamCache(32)

    @classmethod
    def get(cls, device, rows, K, row_aligned=True, sp=None, bucket_cap=0):
        key = (device.index, _stream_ptr(device) if sp is None else sp, rows, K, row_aligned, bucket_cap)
        buf = cls._cache.get(key)
        if buf is None:
            lib = _lib.load_library()
            buf = dict(
                tiled=torch.zeros(lib.mi355q_bfp_tiled_bytes(rows, K), dtype=torch.int8, device=device),
                exp=torch.empty(rows * (K // 16), dtype=torch.uint8, device=device),
                flag=torch.empty(rows, dtype=torch.uint8, device=device),
                gscale=torch.zeros(lib.mi355q_bfp_rows_pad(rows), dtype=torch.float32, device=device),
                sparse=[_new_row_list(device, rows, bucket_cap) for _ in range(2)] if bucket_cap >= 0 else [None, None],
                calls=0)
            cls._cache.put(key, buf)
        return buf


# Several layers often take the SAME activation (q / k / v projections of an attention block, gate / up of a gated MLP):
# the fused quantiser then runs once and the later calls get the operand that is still sitting in the shape's buffers.
# A hit needs the same storage pointer, shape and strides, the same version counter (any in-place write bumps it, through
# views too) and the same quantiser parameters.  The record keeps the tensor alive, so its address cannot have been handed
# to another allocation in between; every quantise call of that shape overwrites the record, so what the record names is
# always what the buffers hold.  Cost: one activation per shape stays allocated until the next call of that shape.
REUSE_QUANTISED_INPUT = os.environ.get("MI355Q_REUSE_INPUT", "1") != "0"    # (timing loops over ONE tensor switch it off)


def _recorded_operand(buf, x, sig):
    last = buf.get("last")
    if last is None or not REUSE_QUANTISED_INPUT:
        return None
    kept, version, lsig, operand = last
    try:
        hit = (x.data_ptr() == kept.data_ptr() and x.shape == kept.shape and x.stride() == kept.stride()
               and x._version == version and lsig == sig)
    except RuntimeError:                 # (inference-mode tensors have no version counter: never reused)
        hit = False
    return operand if hit else None


def _record_operand(buf, x, sig, operand):
    try:
        buf["last"] = (x.detach(), x._version, sig, operand)
    except RuntimeError:
        buf["last"] = None


def block_fp_quantize_aligned_rows(x: torch.Tensor, width: int, exponent_width: int, exponent_bias,
                                   bucket_cap: int = None, pre=None, segments: bool = False) -> AlignedOperand:
    """Fused activation path, ROW-aligned flavour: x [rows, K] fp32 -> quantise ([1,16] blocks) + pack +
    row-align + tile in one kernel (K % 64 == 0, K <= ROW_ALIGN_MAX_K).  Buffers are reused per shape and
    stream like block_fp_quantize_aligned's.  `bucket_cap`: exception entries per 256 rows (default
    ACTIVATION_BUCKET_CAP); anything but 120 makes the GEMM add x's exceptions in its row post-pass; ROW_NO_ALIGN (-1):
    no alignment at all, every block keeps its exponent and the GEMM takes its blockwise-exact kernel.
    `segments`: x is [P, rows, K / P] -- the rank-major result of the all-gather over out_features shards
    (sharded.ShardedRows): row r of the [rows, K] tensor is the concatenation of x[0, r], x[1, r], ...; read in place
    (mi355q_block_fp_quantize_aligned_rows_seg), the same operand as from the re-assembled tensor."""
    bucket_cap = ACTIVATION_BUCKET_CAP if bucket_cap is None else int(bucket_cap)
    _require_device(x, "block_fp_quantize_aligned_rows")
    if segments:
        assert x.ndim == 3 and x.is_contiguous() and x.shape[2] % 4 == 0 and (pre is None or pre[0] == "relu")
        nseg, rows, seg_len = x.shape
        K = nseg * seg_len
    else:
        assert x.ndim == 2
        rows, K = x.shape
        nseg, seg_len = 1, 0
    assert K % 64 == 0 and K <= ROW_ALIGN_MAX_K
    sp = _stream_ptr(x.device)
    buf = _ActivationBuffers.get(x.device, rows, K, row_aligned=True, sp=sp, bucket_cap=bucket_cap)
    bias = _default_bias(exponent_bias)
    # (the record is only good inside the capture sequence -- or the eager stretch -- it was made in: a hit while a graph
    #  is being recorded on a record from the warm-up would leave the quantiser out of the graph)
    sig = (int(width), int(exponent_width), bias, bucket_cap, _lib.load_library().mi355q_stream_capture_id(sp), nseg)
    again = _recorded_operand(buf, x, sig) if pre is None else None
    if again is not None:
        return again
    xc = x.contiguous()
    # (`pre`: quantise relu(x) / silu(x) * other / norm(x) instead of x)
    pre_op, other, eps, third = (PRE_OPS[pre[0]] if pre else PRE_NONE, None, 0.0, None) if segments else _pre_args(xc, pre)
    lib = _lib.load_library()
    if _capturing() and bucket_cap >= 0:
        # HIP-graph capture: a replayed node always sees the pointers it was captured with, so the two alternating lists
        # (each call fills one and clears the other's counts for the next call) cannot work -- the node gets a list of its
        # own, zeroed by a fill node in front of it on every replay
        cur, nxt = _new_row_list(x.device, rows, bucket_cap), None
    else:
        cur, nxt = buf["sparse"][buf["calls"] & 1], buf["sparse"][(buf["calls"] + 1) & 1]
        buf["calls"] += 1
    with _on_device(x.device):
        rc = lib.mi355q_block_fp_quantize_aligned_rows_seg(_ptr(xc), _ptr(other), _ptr(third), pre_op, eps, _ptr(buf["tiled"]),
                                                           _ptr(buf["exp"]), _ptr(buf["flag"]), _ptr(buf["gscale"]), _ptr(cur),
                                                           _ptr(nxt), rows, K, seg_len, rows * seg_len, int(width),
                                                           int(exponent_width), bias, bucket_cap, sp)
    _lib.check(rc, "mi355q_block_fp_quantize_aligned_rows_seg")
    eb = 2 ** (int(exponent_width) - 1) - 1 if bias == BIAS_DEFAULT else bias
    operand = AlignedOperand(rows, K, None, buf["tiled"], buf["exp"], buf["flag"], buf["gscale"], cur,
                             int(width) - 1, eb, row_aligned=True, bucket_cap=bucket_cap)
    if pre is None:
        _record_operand(buf, x, sig, operand)
    else:
        buf["last"] = None                         # (the buffers hold another tensor's operand now)
    return operand


class ColumnClasses:
    """the split of a layer's in_features into two classes of [1,16] block columns for the mixed contraction (bfp_gemm_mixed):
    `blocks0` / `blocks1` -- the block columns of class 0 (row-aligned int8) / class 1 (tiled bf16), in the order the operands
    hold them; `cols0` / `cols1` -- the same as element indices (what index_select takes to split the weights); `cmap` -- the
    uint16 word per block column the class-aware quantiser reads (position | class << 15)."""

    def __init__(self, K: int, blocks1, device):
        nb = K // 16
        b1 = torch.as_tensor(sorted(int(b) for b in blocks1), dtype=torch.long)
        mask = torch.ones(nb, dtype=torch.bool)
        mask[b1] = False
        b0 = torch.nonzero(mask).flatten()
        self.K, self.n0, self.n1 = K, int(b0.numel()), int(b1.numel())
        assert self.n0 % 8 == 0 and self.n1 % 8 == 0 and self.n1 >= 8 and self.n0 >= 16, "whole pairs of 64-byte K-steps in both classes"
        cmap = torch.zeros(nb, dtype=torch.int32)
        cmap[b0] = torch.arange(self.n0, dtype=torch.int32)
        cmap[b1] = torch.arange(self.n1, dtype=torch.int32) | 0x8000
        ar = torch.arange(16)
        self.blocks0, self.blocks1 = b0.to(device), b1.to(device)
        self.cols0 = (b0[:, None] * 16 + ar[None]).reshape(-1).to(device)
        self.cols1 = (b1[:, None] * 16 + ar[None]).reshape(-1).to(device)
        self.cmap = cmap.to(torch.uint16).to(device)

    @property
    def K0(self):
        return 16 * self.n0

    @property
    def K1(self):
        return 16 * self.n1


_CLASS_BUFFERS = _StreamCache(16)


def block_fp_quantize_classes(x: torch.Tensor, classes: ColumnClasses, width: int, exponent_width: int, exponent_bias,
                              bucket_cap: int = None):
    """x [rows, K] fp32 -> (class-0 AlignedOperand [rows, K0], class-1 tiled bf16 operand [rows, K1]) in ONE pass
    (include/mi355q.h, mi355q_block_fp_quantize_classes): the activation side of bfp_gemm_mixed.  Buffers are shared by calls of
    the same shape on the same stream (consume before quantising again)."""
    bucket_cap = ACTIVATION_BUCKET_CAP if bucket_cap is None else int(bucket_cap)
    _require_device(x, "block_fp_quantize_classes")
    assert x.ndim == 2 and x.dtype == torch.float32 and x.shape[1] == classes.K
    rows, K = x.shape
    xc = x.contiguous()
    lib = _lib.load_library()
    sp = _stream_ptr(x.device)
    key = (x.device.index, sp, "cls", rows, K, classes.n0, bucket_cap)
    buf = _CLASS_BUFFERS.get(key)
    if buf is None:
        buf = dict(tiled=torch.zeros(lib.mi355q_bfp_tiled_bytes(rows, classes.K0), dtype=torch.int8, device=x.device),
                   exp=torch.zeros(rows * classes.n0, dtype=torch.uint8, device=x.device),
                   flag=torch.zeros(rows, dtype=torch.uint8, device=x.device),
                   gscale=torch.zeros(lib.mi355q_bfp_rows_pad(rows), dtype=torch.float32, device=x.device),
                   x1=torch.zeros(lib.mi355q_bfp_tiled_bytes(rows, 2 * classes.K1), dtype=torch.int8, device=x.device),
                   sparse=[_new_row_list(x.device, rows, bucket_cap) for _ in range(2)], calls=0)
        _CLASS_BUFFERS.put(key, buf)
    if _capturing():
        cur, nxt = _new_row_list(x.device, rows, bucket_cap), None          # (a captured node keeps a list of its own: block_fp_quantize_aligned_rows)
    else:
        cur, nxt = buf["sparse"][buf["calls"] & 1], buf["sparse"][(buf["calls"] + 1) & 1]
        buf["calls"] += 1
    bias = _default_bias(exponent_bias)
    with _on_device(x.device):
        rc = lib.mi355q_block_fp_quantize_classes(_ptr(xc), _ptr(classes.cmap), classes.n0, classes.n1, _ptr(buf["tiled"]), _ptr(buf["exp"]),
                                                  _ptr(buf["flag"]), _ptr(buf["gscale"]), _ptr(cur), _ptr(nxt), _ptr(buf["x1"]), rows, K,
                                                  int(width), int(exponent_width), bias, bucket_cap, sp)
    _lib.check(rc, "mi355q_block_fp_quantize_classes")
    eb = 2 ** (int(exponent_width) - 1) - 1 if bias == BIAS_DEFAULT else bias
    x0 = AlignedOperand(rows, classes.K0, None, buf["tiled"], buf["exp"], buf["flag"], buf["gscale"], cur, int(width) - 1, eb,
                        row_aligned=True, bucket_cap=bucket_cap)
    return x0, buf["x1"]


def bfp_gemm_aligned(x: AlignedOperand, w: AlignedOperand, bias=None, out: torch.Tensor = None, residual: torch.Tensor = None):
    """bfp_gemm on operands rewritten by bfp_align.  `residual` [M, N] fp32: y = (x . w^T + bias) + residual -- in the product's stores
    on the one-launch route of row-aligned operands (mi355q_bfp_gemm_aligned_res: the same bits as the separate add), as a torch add
    behind the product elsewhere."""
    M, K, N = x.rows, x.K, w.rows
    assert w.K == K
    given = out is not None
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.tiled.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    ldy = out.stride(0) if M > 1 else max(N, out.stride(0))
    lib = _lib.load_library()
    sp = _stream_ptr(x.tiled.device)
    x.c_struct(), w.c_struct()
    rc = _lib.E_UNSUPPORTED
    with _on_device(x.tiled.device):
        if residual is not None:
            assert residual.dtype == torch.float32 and residual.shape == (M, N) and residual.stride(1) == 1 and residual.device == out.device
            if residual.stride(0) % 4 == 0 and residual.data_ptr() % 16 == 0:
                rc = lib.mi355q_bfp_gemm_aligned_res(x._cs_addr, w._cs_addr, _ptr(bias), _ptr(residual),
                                                     residual.stride(0) if M > 1 else max(N, residual.stride(0)), _ptr(out), M, N, K, ldy, sp)
        fused = rc != _lib.E_UNSUPPORTED
        if not fused:
            rc = lib.mi355q_bfp_gemm_aligned(x._cs_addr, w._cs_addr, _ptr(bias), _ptr(out), M, N, K, ldy, sp)
    _lib.check(rc, "mi355q_bfp_gemm_aligned")
    if residual is not None and not fused:
        out.add_(residual)
    if given:
        _wrote_into(out)
    return out


def bfp_gemm_mixed(x0: AlignedOperand, w0: AlignedOperand, x1: torch.Tensor, w1: torch.Tensor, K1: int, bias=None,
                   out: torch.Tensor = None):
    """y = x . w^T + bias with the contraction in two column classes, ONE launch (include/mi355q.h, mi355q_bfp_gemm_mixed): class 0
    = the row-aligned int8 operands x0 [M, K0], w0 [N, K0] (int8 MFMA, exception lists), class 1 = the tiled bf16 operands x1
    [M, K1], w1 [N, K1] (block_fp_quantize_bf16_tiled / bf16_tile: every block its own exponent, bf16 MFMA) -- activations with
    outlier channels keep three quarters and more of their MFMA work at the int8 rate.  None when the library does not take the
    shapes (callers use the per-block route)."""
    M, K0, N = x0.rows, x0.K, w0.rows
    assert w0.K == K0 and x1.dtype == torch.int8 and w1.dtype == torch.int8
    given = out is not None
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x0.tiled.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    ldy = out.stride(0) if M > 1 else max(N, out.stride(0))
    lib = _lib.load_library()
    x0.c_struct(), w0.c_struct()
    with _on_device(x0.tiled.device):
        rc = lib.mi355q_bfp_gemm_mixed(x0._cs_addr, w0._cs_addr, _ptr(x1), _ptr(w1), _ptr(bias), _ptr(out), M, N, K0, int(K1), ldy,
                                       _stream_ptr(x0.tiled.device))
    if rc == _lib.E_UNSUPPORTED:
        return None
    _lib.check(rc, "mi355q_bfp_gemm_mixed")
    if given:
        _wrote_into(out)
    return out


def interleave_gate_up(wg: AlignedOperand, wu: AlignedOperand):
    """gate_proj's and up_proj's row-aligned operands [I, K] as ONE operand [2 I, K] with their rows interleaved in chunks of 16
    (rows 32 c .. 32 c + 15 = gate rows 16 c .., rows 32 c + 16 .. = the same rows of up): what bfp_gemm_aligned_gated multiplies.
    Tiled mantissas are copied piece row by piece row, exponents / flags / scales row chunk by row chunk, the exception lists
    are re-bucketed on the host with their row numbers mapped (a one-off per pair of layers).  None when the pair does not
    qualify (I % 128, different formats, overflowed lists, a re-bucketed bucket beyond its 120 entries)."""
    import numpy as np
    I, K = wg.rows, wg.K
    if (wu.rows != I or wu.K != K or I % 128 or K % 64 or wg.mbits != wu.mbits or wg.exp_bias != wu.exp_bias
            or wg.list_cap != ROW_BUCKET_CAP or wu.list_cap != ROW_BUCKET_CAP or wg.unaligned or wu.unaligned):
        return None
    dev = wg.tiled.device
    nb = K // 16
    pr = K * 16                                               # bytes of one piece row (16 rows x K)
    il = lambda a, b, unit: torch.stack((a.reshape(I // 16, unit), b.reshape(I // 16, unit)), dim=1).reshape(-1).contiguous()
    tiled = il(wg.tiled[: I * K], wu.tiled[: I * K], pr)
    exp = il(wg.exp.reshape(-1)[: I * nb], wu.exp.reshape(-1)[: I * nb], 16 * nb)
    flag = il(wg.rowflag[:I], wu.rowflag[:I], 16)
    gscale = il(wg.gscale[:I], wu.gscale[:I], 16)
    og, eg = row_list_entries(wg.sparse, I)
    ou, eu = row_list_entries(wu.sparse, I)
    if og or ou:
        return None
    eg, eu = eg.copy(), eu.copy()
    eg[:, 0] = (eg[:, 0] // 16) * 32 + eg[:, 0] % 16
    eu[:, 0] = (eu[:, 0] // 16) * 32 + 16 + eu[:, 0] % 16
    ent = np.concatenate([eg, eu]) if len(eg) + len(eu) else np.zeros((0, 8), np.int32)
    words = 8 + 8 * ROW_BUCKET_CAP
    nbk = (2 * I + ROW_BUCKET_ROWS - 1) // ROW_BUCKET_ROWS
    lst = np.zeros(8 + nbk * words, np.int32)
    for b in range(nbk):
        mine = ent[(ent[:, 0] // ROW_BUCKET_ROWS) == b]
        mine = mine[np.lexsort((mine[:, 1], mine[:, 0]))]
        if len(mine) > ROW_BUCKET_CAP:
            return None
        base = 8 + b * words
        lst[base] = len(mine)
        lst[base + 8: base + 8 + 8 * len(mine)] = mine.reshape(-1)
    sparse = torch.from_numpy(lst).to(dev)
    return AlignedOperand(2 * I, K, None, tiled, exp, flag, gscale, sparse, wg.mbits, wg.exp_bias, row_aligned=True, bucket_cap=ROW_BUCKET_CAP)


_GATED_BUFFERS = _StreamCache(8)


def bfp_gemm_aligned_gated(x: AlignedOperand, w_gu: AlignedOperand, q_width: int, q_exponent_width: int, q_exponent_bias, bias_gu=None):
    """x [M, K] against the interleaved gate / up operand (interleave_gate_up) with silu(gate) * up and the CONSUMER's block_fp
    quantiser in the store epilogue (include/mi355q.h, mi355q_bfp_gemm_aligned_gated): returns the tiled bf16 operand [M, I] that
    bf16_gemm_tiled multiplies against down_proj's weights -- the buffer is shared by calls of the same shape on the same stream
    (consume before the next call).  None when the library does not take the shapes."""
    M, K, I = x.rows, x.K, w_gu.rows // 2
    assert w_gu.K == K
    lib = _lib.load_library()
    dev = x.tiled.device
    sp = _stream_ptr(dev)
    key = (dev.index, sp, "gated", M, I)
    buf = _GATED_BUFFERS.get(key)
    if buf is None:
        # (the fp32 scratch is touched on the kernel's slow paths only; the pages of an untouched allocation cost nothing)
        buf = dict(out=torch.zeros(lib.mi355q_bfp_tiled_bytes(M, 2 * I), dtype=torch.int8, device=dev),
                   scratch=torch.empty(M, 2 * I, dtype=torch.float32, device=dev))
        _GATED_BUFFERS.put(key, buf)
    x.c_struct(), w_gu.c_struct()
    with _on_device(dev):
        rc = lib.mi355q_bfp_gemm_aligned_gated(x._cs_addr, w_gu._cs_addr, _ptr(bias_gu), _ptr(buf["scratch"]), _ptr(buf["out"]), M, I, K,
                                               int(q_width), int(q_exponent_width), _default_bias(q_exponent_bias), sp)
    if rc == _lib.E_UNSUPPORTED:
        return None
    _lib.check(rc, "mi355q_bfp_gemm_aligned_gated")
    return buf["out"]


def bfp_gemm_aligned_relu(x: AlignedOperand, w: AlignedOperand, q_width: int, q_exponent_width: int, q_exponent_bias, bias=None):
    """relu(x . w^T + bias) quantised with the CONSUMER's block_fp quantiser in the product's store epilogue (include/mi355q.h,
    mi355q_bfp_gemm_aligned_relu; OPT's fc1 in front of fc2): returns the tiled bf16 operand [M, N] that bf16_gemm_tiled multiplies
    against the consumer's weights (buffer shared per shape and stream).  None when the library does not take the shapes."""
    M, K, N = x.rows, x.K, w.rows
    assert w.K == K
    lib = _lib.load_library()
    dev = x.tiled.device
    sp = _stream_ptr(dev)
    key = (dev.index, sp, "relu", M, N)
    buf = _GATED_BUFFERS.get(key)
    if buf is None:
        buf = dict(out=torch.zeros(lib.mi355q_bfp_tiled_bytes(M, 2 * N), dtype=torch.int8, device=dev),
                   scratch=torch.empty(M, N, dtype=torch.float32, device=dev))
        _GATED_BUFFERS.put(key, buf)
    x.c_struct(), w.c_struct()
    with _on_device(dev):
        rc = lib.mi355q_bfp_gemm_aligned_relu(x._cs_addr, w._cs_addr, _ptr(bias), _ptr(buf["scratch"]), _ptr(buf["out"]), M, N, K,
                                              int(q_width), int(q_exponent_width), _default_bias(q_exponent_bias), sp)
    if rc == _lib.E_UNSUPPORTED:
        return None
    _lib.check(rc, "mi355q_bfp_gemm_aligned_relu")
    return buf["out"]


def bfp_gemm_aligned_multi(x: AlignedOperand, ws, biases=None, outs=None):
    """[x . w^T + bias for w in ws] in ONE launch of the tile GEMM (equally shaped row-aligned weight operands: q / k / v,
    gate / up); returns None when the library does not take the group (callers then use bfp_gemm_aligned per weight).
    `outs`: where to store the products (contiguous [M, N] fp32 each; None entries are allocated)."""
    import ctypes
    M, K, N = x.rows, x.K, ws[0].rows
    n = len(ws)
    if not (1 <= n <= 3) or any(w.K != K or w.rows != N for w in ws):
        return None
    outs = [o if o is not None and o.shape == (M, N) and o.dtype == torch.float32 and o.is_contiguous()
            else torch.empty(M, N, dtype=torch.float32, device=x.tiled.device) for o in (outs or [None] * n)]
    x.c_struct()
    for w in ws:
        w.c_struct()
    wp = (ctypes.c_void_p * n)(*[w._cs_addr for w in ws])
    bp = (ctypes.c_void_p * n)(*[(_ptr(b) if b is not None else None) for b in (biases or [None] * n)])
    yp = (ctypes.c_void_p * n)(*[_ptr(o) for o in outs])
    with _on_device(x.tiled.device):
        rc = _lib.load_library().mi355q_bfp_gemm_aligned_multi(x._cs_addr, ctypes.addressof(wp), ctypes.addressof(bp), ctypes.addressof(yp),
                                                              n, M, N, K, N, _stream_ptr(x.tiled.device))
    if rc == _lib.E_UNSUPPORTED:
        return None
    _lib.check(rc, "mi355q_bfp_gemm_aligned_multi")
    return outs


_MATMUL_WS = _StreamCache(8)


def bfp_matmul_supported(x: torch.Tensor, y: torch.Tensor, x_width: int, y_width: int) -> bool:
    """shapes / widths the fused quantise + matmul kernel takes (include/mi355q.h, mi355q_bfp_matmul)"""
    return (x.is_cuda and y.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and x.ndim == 3
            and y.ndim == 3 and x.shape[0] == y.shape[0] and x.shape[2] == y.shape[1] and x.shape[0] <= 65535
            and x.shape[2] % 16 == 0 and y.shape[2] % 16 == 0 and x.shape[2] > 0 and 2 <= int(x_width) <= 9
            and 2 <= int(y_width) <= 9)


def bfp_softmax_matmul_supported(x, y, x_width, y_width) -> bool:
    return bfp_matmul_supported(x, y, x_width, y_width) and x.shape[2] > 192 and y.shape[2] <= 128


def bfp_matmul(x: torch.Tensor, y: torch.Tensor, x_width: int, x_exponent_width: int, x_exponent_bias, y_width: int,
               y_exponent_width: int, y_exponent_bias, *, softmax: bool = False, mask: torch.Tensor = None,
               causal: bool = False) -> torch.Tensor:
    """out[b] = Qx(x[b]) @ Qy(y[b]) for x [B, M, K], y [B, K, N] fp32: block_fp [1,16] blocks along each operand's last
    dim (reference matmul.py:146-196), x quantised on its way into the MFMAs (one pass over x, no fake-quantised copy)"""
    _require_device(x, "bfp_matmul")
    assert bfp_matmul_supported(x, y, x_width, y_width)
    B, M, K = x.shape
    N = y.shape[2]
    xc, yc = x.contiguous(), y.contiguous()
    out = torch.empty(B, M, N, dtype=torch.float32, device=x.device)
    lib = _lib.load_library()
    sp = _stream_ptr(x.device)
    key = (x.device.index, sp, B, K, N)
    ws = _MATMUL_WS.get(key)
    if ws is None:
        ws = _MATMUL_WS.put(key, torch.empty(lib.mi355q_bfp_matmul_workspace_bytes(B, K, N), dtype=torch.uint8, device=x.device))
    args = (B, M, K, N, int(x_width), int(x_exponent_width), _default_bias(x_exponent_bias), int(y_width), int(y_exponent_width),
            _default_bias(y_exponent_bias), sp)
    with _on_device(x.device):
        if softmax:
            if mask is not None:
                assert mask.shape == (M, K) and mask.dtype == torch.float32 and mask.is_contiguous() and mask.device == x.device
            rc = lib.mi355q_bfp_softmax_matmul(_ptr(xc), _ptr(mask), int(bool(causal)), _ptr(yc), _ptr(out), _ptr(ws), *args)
        else:
            rc = lib.mi355q_bfp_matmul(_ptr(xc), _ptr(yc), _ptr(out), _ptr(ws), *args)
    _lib.check(rc, "mi355q_bfp_softmax_matmul" if softmax else "mi355q_bfp_matmul")
    return out


def values_matmul_supported(x: torch.Tensor, y: torch.Tensor, arith: str, x_params, y_params=None, softmax: bool = False) -> bool:
    """shapes / settings the fused block_minifloat / block_log products take (include/mi355q.h, ABI 19).
    x_params / y_params: (width, exponent_width, exponent_bias_width) for block_minifloat, (width, exponent_bias_width) for
    block_log (y is not quantised there)"""
    if not (x.is_cuda and y.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and x.ndim == 3 and y.ndim == 3
            and x.shape[0] == y.shape[0] and x.shape[2] == y.shape[1] and x.shape[0] <= 65535 and x.shape[2] % 16 == 0
            and y.shape[2] % 16 == 0 and x.shape[2] > 0):
        return False
    if softmax and not (arith == "block_minifloat" and x.shape[2] > 192 and y.shape[2] <= 128):
        return False
    if arith == "block_minifloat":
        return all(1 <= int(ew) <= 8 and 1 <= int(ebw) <= 8 and 0 <= int(w) - int(ew) - 1 <= 7 for w, ew, ebw in (x_params, y_params))
    if arith == "block_log":
        return 2 <= int(x_params[0]) <= 9 and 1 <= int(x_params[1]) <= 8
    return False


def values_matmul(x: torch.Tensor, y: torch.Tensor, arith: str, x_params, y_params=None, *, softmax: bool = False,
                  mask: torch.Tensor = None, causal: bool = False) -> torch.Tensor:
    """out[b] = Qx(x[b]) @ Qy(y[b]) for block_minifloat (reference matmul.py:199-249), out[b] = Qx(x[b]) @ y[b] for block_log
    (matmul.py:252-297: y is not quantised): x [B, M, K], y [B, K, N] fp32, [1,16] blocks along each operand's last dim, x
    quantised on its way into the MFMAs -- bfp_matmul's two kernels with the other block quantisers."""
    _require_device(x, "values_matmul")
    assert values_matmul_supported(x, y, arith, x_params, y_params, softmax)
    B, M, K = x.shape
    N = y.shape[2]
    xc, yc = x.contiguous(), y.contiguous()
    out = torch.empty(B, M, N, dtype=torch.float32, device=x.device)
    lib = _lib.load_library()
    sp = _stream_ptr(x.device)
    log = arith == "block_log"
    key = (x.device.index, sp, B, K, N, log)
    ws = _MATMUL_WS.get(key)
    if ws is None:
        nbytes = (lib.mi355q_block_log_matmul_workspace_bytes if log else lib.mi355q_bfp_matmul_workspace_bytes)(B, K, N)
        ws = _MATMUL_WS.put(key, torch.empty(nbytes, dtype=torch.uint8, device=x.device))
    with _on_device(x.device):
        if log:
            name = "mi355q_block_log_matmul"
            rc = lib.mi355q_block_log_matmul(_ptr(xc), _ptr(yc), _ptr(out), _ptr(ws), B, M, K, N, int(x_params[0]), int(x_params[1]), sp)
        else:
            args = (B, M, K, N, *(int(v) for v in x_params), *(int(v) for v in y_params), sp)
            if softmax:
                if mask is not None:
                    assert mask.shape == (M, K) and mask.dtype == torch.float32 and mask.is_contiguous() and mask.device == x.device
                name = "mi355q_block_minifloat_softmax_matmul"
                rc = lib.mi355q_block_minifloat_softmax_matmul(_ptr(xc), _ptr(mask), int(bool(causal)), _ptr(yc), _ptr(out), _ptr(ws), *args)
            else:
                name = "mi355q_block_minifloat_matmul"
                rc = lib.mi355q_block_minifloat_matmul(_ptr(xc), _ptr(yc), _ptr(out), _ptr(ws), *args)
    _lib.check(rc, name)
    return out


_ATTN_WS = _StreamCache(8)
ATTENTION_MAX_KEYS, ATTENTION_MAX_HEAD_DIM = 1 << 20, 128      # (beyond 2048 keys: the streaming kernel, scores formed twice)


def _as_heads_view(t: torch.Tensor):
    """[..., T, D] -> (tensor3 [B, T, D], batch stride, row stride) without copying when the leading dims fold into one
    stride (always for one leading dim; for [B, H, T, D] views of [B, T, H, D] projections whenever B == 1) and the
    innermost stride is 1 with 16-byte aligned rows; a contiguous copy otherwise"""
    lead = t.shape[:-2]
    T, D = t.shape[-2:]
    if t.stride(-1) == 1 and t.stride(-2) % 4 == 0 and t.data_ptr() % 16 == 0:
        dims = [(n, st) for n, st in zip(lead, t.stride()[:-2]) if n != 1]
        ok, bstride = True, (dims[-1][1] if dims else T * D)
        for (n0, s0), (n1, s1) in zip(dims[:-1], dims[1:]):
            ok = ok and s0 == n1 * s1
        if ok and bstride % 4 == 0:
            nb = 1
            for n in lead:
                nb *= n
            return t.as_strided((nb, T, D), (bstride, t.stride(-2), 1)), bstride, t.stride(-2)
    c = t.reshape(-1, T, D).contiguous()
    return c, T * D, D


def bfp_attention_supported(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, widths) -> bool:
    """shapes / widths the one-pass attention kernel takes (include/mi355q.h, mi355q_bfp_attention): q [..., M, D],
    k / v [..., T, D] with equal leading dims"""
    return (q.is_cuda and k.is_cuda and v.is_cuda and q.dtype == k.dtype == v.dtype == torch.float32 and q.ndim >= 3
            and k.ndim == q.ndim and v.shape == k.shape and q.shape[:-2] == k.shape[:-2] and q.shape[-1] == k.shape[-1]
            and 0 < q.shape[:-2].numel() <= 65535 and 0 < k.shape[-2] <= ATTENTION_MAX_KEYS and k.shape[-2] % 16 == 0
            and q.shape[-2] > 0 and q.shape[-1] % 32 == 0 and 0 < q.shape[-1] <= ATTENTION_MAX_HEAD_DIM
            and all(2 <= int(w) <= 9 for w in widths))


def bfp_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, qk_params, pv_params, *, mask: torch.Tensor = None,
                  causal: bool = False, scale_div: float = None, token_major: bool = False, rope=None, consumer=None,
                  q_scale: float = None):
    """out[b] = Qc(softmax(max(Qa(q[b]) @ Qb(k[b]^T) [/ scale_div] + mask, finfo.min))) @ Qd(v[b]) for q [..., M, D], k and v
    [..., T, D] fp32 (k untransposed; strided head views are read in place), block_fp [1,16] blocks along each operand's
    last dim as the reference's two products apply them (matmul.py:146-196); neither scores nor probabilities are
    written.  qk_params / pv_params: (x width, x exponent width, x exponent bias, y width, y exponent width, y exponent
    bias) of bmm_0 / bmm_1.  Returns a contiguous tensor of q's shape -- or, with `token_major` and q [1, H, M, D], the
    [1, H, M, D] view of a contiguous [1, M, H, D] buffer: what the models' `attn_output.transpose(1, 2).reshape(B, T, H * D)`
    (modeling_opt.py:318-322, modeling_llama.py:349-350) then takes without a copy.
    `rope` = (cos_q, sin_q [rows, D] quantised tables, position_ids int64 [batch, M]) for q, k [batch, heads, M, D]: the rotary
    embedding of q and k (modeling_llama.py:289-299) applied as the pass loads them (bfp_attention_rope_supported; the same bits as
    rope_apply first).
    `consumer` = (width, exponent width, exponent bias) of the out-projection's data_in quantiser, q [1, H, M, D], no mask: the result is
    NOT an fp32 tensor but `TiledBf16` -- that Linear's quantised activations [M, H D] as the tiled bf16 operand of its per-block
    product, written by the kernels' store epilogue (bfp_attention_consumer_supported; the same bits as block_fp_quantize_bf16_tiled
    of the fp32 output, which is never written).
    `q_scale`: q * q_scale (OPT's `q_proj(x) * scaling`, modeling_opt.py:231) formed as the Q fragments are packed instead of by a torch
    kernel in front (bfp_attention_q_scale_supported; the same bits)."""
    import ctypes
    _require_device(q, "bfp_attention")
    assert not q_scale or bfp_attention_q_scale_supported(q, k, rope)
    assert bfp_attention_supported(q, k, v, (qk_params[0], qk_params[3], pv_params[0], pv_params[3]))
    M, D = q.shape[-2:]
    T = k.shape[-2]
    q3, qsb, qsm = _as_heads_view(q)
    k3, ksb, kst = _as_heads_view(k)
    v3, vsb, vst = _as_heads_view(v)
    B = q3.shape[0]
    if token_major and q.ndim == 4 and q.shape[0] == 1:
        H = q.shape[1]
        out = None if consumer is not None else torch.empty(1, M, H, D, dtype=torch.float32, device=q.device).permute(0, 2, 1, 3)
        osb, osm = D, H * D
    else:
        out = None if consumer is not None else torch.empty(*q.shape, dtype=torch.float32, device=q.device)
        osb, osm = M * D, D
    lib = _lib.load_library()
    tiled = pc = None
    if consumer is not None:
        assert bfp_attention_consumer_supported(q, mask)
        tiled = torch.empty(lib.mi355q_bfp_tiled_bytes(M, 2 * B * D), dtype=torch.int8, device=q.device)
        pc = (ctypes.c_int32 * 3)(int(consumer[0]), int(consumer[1]), _default_bias(consumer[2]))
    sp = _stream_ptr(q.device)
    key = (q.device.index, sp, B, T, D)
    ws = _ATTN_WS.get(key)
    if ws is None:
        ws = _ATTN_WS.put(key, torch.empty(lib.mi355q_bfp_attention_workspace_bytes(B, T, D), dtype=torch.uint8, device=q.device))
    if mask is not None:
        assert mask.shape == (M, T) and mask.dtype == torch.float32 and mask.is_contiguous() and mask.device == q.device
    pa = (ctypes.c_int32 * 6)(*[_default_bias(p) if i % 3 == 2 else int(p) for i, p in enumerate(qk_params)])
    pb = (ctypes.c_int32 * 6)(*[_default_bias(p) if i % 3 == 2 else int(p) for i, p in enumerate(pv_params)])
    strides = (ctypes.c_int64 * 8)(qsb, qsm, ksb, kst, vsb, vst, osb, osm)
    cos_q = sin_q = pos = None
    rows = heads = 0
    if rope is not None:
        cos_q, sin_q, pos = rope
        assert bfp_attention_rope_supported(q, k, cos_q, sin_q, pos)
        rows, heads = cos_q.shape[0], q.shape[1]
    with _on_device(q.device):
        rc = lib.mi355q_bfp_attention_fused(_ptr(q3), _ptr(k3), _ptr(v3), _ptr(mask), int(bool(causal)), float(q_scale) if q_scale else 0.0,
                                            float(scale_div) if scale_div else 0.0, _ptr(out), _ptr(tiled), ctypes.addressof(pc) if pc else None,
                                            _ptr(ws), B, M, T, D, ctypes.addressof(pa), ctypes.addressof(pb), ctypes.addressof(strides),
                                            _ptr(cos_q), _ptr(sin_q), _ptr(pos), rows, max(1, heads), sp)
    _lib.check(rc, "mi355q_bfp_attention_fused")
    return out if tiled is None else TiledBf16(tiled, M, B * D)


class TiledBf16:
    """quantised activations [rows, cols] as the tiled bf16 operand of the per-block product (block_fp_quantize_bf16_tiled's output with
    its shape): what a producer hands a Linear that runs on that route (`Linear.forward_tiled`)"""

    def __init__(self, buf: torch.Tensor, rows: int, cols: int):
        self.buf, self.rows, self.cols = buf, int(rows), int(cols)


def bfp_attention_q_scale_supported(q, k, rope=None) -> bool:
    """what q_scale takes (mi355q_bfp_attention_fused): head_dim 64 / 128, no more queries than keys, no rotary embedding"""
    return rope is None and q.shape[-1] in (64, 128) and q.shape[-2] <= k.shape[-2]


def bfp_attention_consumer_supported(q, mask) -> bool:
    """what the consumer's operand as the attention output takes (include/mi355q.h, mi355q_bfp_attention_fused): ONE batch element
    [1, H, M, D], head_dim 64 or 128, no additive mask (causal is fine)"""
    return q.ndim == 4 and q.shape[0] == 1 and q.shape[-1] in (64, 128) and mask is None


def bfp_attention_rope_supported(q, k, cos_q, sin_q, position_ids) -> bool:
    """what the rotary embedding on load takes (include/mi355q.h, mi355q_bfp_attention_rope): q, k [batch, heads, T, D] over the same
    positions, head_dim 64 or 128, contiguous fp32 tables [rows, D], int64 position_ids [batch, T]"""
    return (q.ndim == 4 and k.shape == q.shape and q.shape[-1] in (64, 128) and torch.is_tensor(cos_q) and torch.is_tensor(sin_q)
            and cos_q.dtype == sin_q.dtype == torch.float32 and cos_q.ndim == 2 and cos_q.shape == sin_q.shape and cos_q.shape[1] == q.shape[-1]
            and cos_q.is_contiguous() and sin_q.is_contiguous() and cos_q.device == q.device and sin_q.device == q.device
            and position_ids.dtype == torch.int64 and position_ids.shape == (q.shape[0], q.shape[2]) and position_ids.is_contiguous()
            and position_ids.device == q.device)


def attention_set_kernel(which: int) -> int:
    """0: the library picks by size, 1: resident-score kernel, 2: streaming kernel; returns the previous setting"""
    return _lib.load_library().mi355q_bfp_attention_set_kernel(int(which))


def attention_set_qpack(on: int) -> int:
    """1 (default): the pack launch in front of the attention kernels also leaves the quantised Q fragments where that pays (T <= 2048,
    64 <= M <= T, head_dim 128 or head_dim 64 with T > 1024); 0: q is always quantised inside the attention kernels (before round 6); 2: fragments wherever they fit
    (head_dim 64 / 128: tests).  The same bits; returns the previous setting."""
    return _lib.load_library().mi355q_bfp_attention_set_qpack(int(on))


def rope_apply(q: torch.Tensor, k: torch.Tensor, cos_q: torch.Tensor, sin_q: torch.Tensor, position_ids: torch.Tensor):
    """(q * cos[pos] + rotate_half(q) * sin[pos], the same for k) for q [B, Hq, T, D], k [B, Hk, T, D] fp32 (any strides
    with a unit innermost one), cos_q / sin_q [rows, D] already quantised, position_ids int64 [B, T]: one launch
    (include/mi355q.h, mi355q_rope_apply).  Outputs are contiguous."""
    import ctypes
    _require_device(q, "rope_apply")
    B, Hq, T, D = q.shape
    Hk = k.shape[1]
    assert k.shape == (B, Hk, T, D) and q.stride(3) == 1 and k.stride(3) == 1 and D % 8 == 0
    assert cos_q.shape == sin_q.shape and cos_q.shape[1] == D and cos_q.is_contiguous() and sin_q.is_contiguous()
    assert position_ids.dtype == torch.int64 and position_ids.shape == (B, T) and position_ids.is_contiguous()
    qo = torch.empty(B, Hq, T, D, dtype=torch.float32, device=q.device)
    ko = torch.empty(B, Hk, T, D, dtype=torch.float32, device=q.device)
    qs = (ctypes.c_int64 * 3)(q.stride(0), q.stride(1), q.stride(2))
    ks = (ctypes.c_int64 * 3)(k.stride(0), k.stride(1), k.stride(2))
    with _on_device(q.device):
        rc = _lib.load_library().mi355q_rope_apply(_ptr(q), _ptr(k), _ptr(cos_q), _ptr(sin_q), _ptr(position_ids), _ptr(qo), _ptr(ko),
                                                   B, Hq, Hk, T, D, cos_q.shape[0], ctypes.addressof(qs), ctypes.addressof(ks),
                                                   _stream_ptr(q.device))
    _lib.check(rc, "mi355q_rope_apply")
    return qo, ko


def set_gemm_variant(variant: int) -> int:
    return _lib.load_library().mi355q_bfp_gemm_set_variant(int(variant))


def gemm_timing(enable: bool) -> None:
    """Bracket the main GEMM kernel of every bfp_gemm_aligned call with HIP events (benchmarks)."""
    _lib.load_library().mi355q_gemm_timing_enable(int(bool(enable)))


def gemm_timing_read():
    """-> (count, avg_ms, min_ms) of the recorded main-kernel launches; clears the record."""
    import ctypes
    n, avg, mn = ctypes.c_int32(0), ctypes.c_float(0), ctypes.c_float(0)
    rc = _lib.load_library().mi355q_gemm_timing_read(ctypes.addressof(n), ctypes.addressof(avg), ctypes.addressof(mn))
    _lib.check(rc, "mi355q_gemm_timing_read")
    return n.value, avg.value, mn.value
