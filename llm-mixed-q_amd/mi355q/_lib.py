"""ctypes binding of include/mi355q.h.  This file is the reference-side FFI stub a maintainer
would add: plain pointers and sizes in, int status out."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_LIB = None

# name -> (restype, argtypes); must list every symbol include/mi355q.h declares
_i32, _i64, _u32, _vp = C.c_int32, C.c_int64, C.c_uint32, C.c_void_p
SIGNATURES = {
    "mi355q_abi_version": (C.c_int, []),
    "mi355q_error_string": (C.c_char_p, [C.c_int]),
    "mi355q_block_fp_quantize_bf16": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mi355q_block_minifloat_quantize_bf16": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mi355q_block_log_quantize_bf16": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mi355q_block_fp_quantize_bf16_tiled": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "mi355q_block_fp_quantize_bf16_tiled_pre": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "mi355q_block_fp_quantize_bf16_tiled_norm": (C.c_int, [_vp, _vp, _i32, C.c_float, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "mi355q_block_fp_quantize_aligned_rows_pre": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32,
                                                            _i32, _i32, _i32, _vp]),
    "mi355q_block_fp_quantize_aligned_rows_norm": (C.c_int, [_vp, _vp, _vp, _i32, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                                             _i64, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_block_fp_quantize_aligned_rows_seg": (C.c_int, [_vp, _vp, _vp, _i32, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                                            _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_block_minifloat_quantize_bf16_tiled": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "mi355q_bfp_packed_bytes": (C.c_size_t, [_i64, _i64, _i32]),
    "mi355q_bfp_pack_bits": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp]),
    "mi355q_bfp_expand": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mi355q_bf16_tile": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "mi355q_fp32_split_tile": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp]),
    "mi355q_bf16_gemm_tiled": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mi355q_bf16_gemm_tiled_res": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mi355q_bf16_gemm_tiled_seg": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i64, _vp]),
    "mi355q_block_fp_quantize": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                           _i32, _i32, _i32, _u32, _vp, _vp]),
    "mi355q_block_minifloat_quantize": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                                  _i32, _i32, _i32, _u32, _vp, _vp]),
    "mi355q_block_log_quantize": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                            _i32, _i32, _u32, _vp, _vp]),
    "mi355q_minifloat_quantize": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_log_quantize": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "mi355q_integer_quantize": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "mi355q_bfp_gemm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
                                  _i32, _i32, _i32, _i32, _vp]),
    "mi355q_bfp_rowflag_bytes": (C.c_size_t, [_i64, _i64]),
    "mi355q_bfp_rows_pad": (_i64, [_i64]),
    "mi355q_bfp_tiled_bytes": (C.c_size_t, [_i64, _i64]),
    "mi355q_bfp_row_list_bytes": (C.c_size_t, [_i64, _i32]),
    "mi355q_bfp_align_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _vp]),
    "mi355q_block_fp_quantize_aligned_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32,
                                                        _i32, _i32, _vp]),
    "mi355q_bfp_matmul_workspace_bytes": (C.c_size_t, [_i64, _i64, _i64]),
    "mi355q_bfp_softmax_matmul": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_bfp_matmul": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_block_minifloat_matmul": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_block_minifloat_softmax_matmul": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_block_log_matmul_workspace_bytes": (C.c_size_t, [_i64, _i64, _i64]),
    "mi355q_block_log_matmul": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp]),
    "mi355q_bfp_attention_set_kernel": (C.c_int, [C.c_int]),
    "mi355q_bfp_attention_set_qpack": (C.c_int, [C.c_int]),
    "mi355q_bfp_attention_workspace_bytes": (C.c_size_t, [_i64, _i64, _i64]),
    "mi355q_bfp_attention": (C.c_int, [_vp, _vp, _vp, _vp, _i32, C.c_float, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "mi355q_bfp_attention_strided": (C.c_int, [_vp, _vp, _vp, _vp, _i32, C.c_float, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mi355q_bfp_attention_rope": (C.c_int, [_vp, _vp, _vp, _vp, _i32, C.c_float, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _i64, _i32, _vp]),
    "mi355q_bfp_attention_fused": (C.c_int, [_vp, _vp, _vp, _vp, _i32, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                             _vp, _i64, _i32, _vp]),
    "mi355q_stream_capture_id": (C.c_uint64, [_vp]),
    "mi355q_rope_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "mi355q_bfp_gemm_aligned": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mi355q_bfp_gemm_aligned_res": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp]),
    "mi355q_bfp_gemm_aligned_multi": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _vp]),
    "mi355q_block_fp_quantize_classes": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_bfp_gemm_aligned_gated": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp]),
    "mi355q_bfp_gemm_aligned_relu": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp]),
    "mi355q_bfp_gemm_mixed": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    "mi355q_mx_plane_bytes": (C.c_size_t, [_i64, _i64, _i32]),
    "mi355q_block_fp_quantize_mx": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "mi355q_mx_gemm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp]),
    "mi355q_gemm_timing_enable": (C.c_int, [C.c_int]),
    "mi355q_gemm_timing_read": (C.c_int, [_vp, _vp, _vp]),
    "mi355q_bfp_gemm_set_variant": (C.c_int, [C.c_int]),
}


class BfpOperand(C.Structure):
    """struct mi355q_bfp_operand"""
    _fields_ = [("mant", _vp), ("exp", _vp), ("rowflag", _vp), ("gscale", _vp), ("list", _vp),
                ("list_cap", _i32), ("mbits", _i32), ("exp_bias", _i32), ("row_aligned", _i32)]


ABI_VERSION = 24
WORKSPACE_BYTES = 16384
ZERO_BLOCK_EXACT, ZERO_BLOCK_FAST = 0, 1


def library_path() -> Path:
    return Path(os.environ.get("MI355Q_LIBRARY", _HERE / "libmi355q.so"))


def load_library() -> C.CDLL:
    """Load libmi355q.so (built by `make -C llm-mixed-q_amd/csrc` or __graft_entry__.build()).
    Raises -- never falls back -- when the library is missing or its ABI version differs."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch first: the library must bind to the HIP runtime torch has loaded (a second copy of the runtime, loaded
    # before torch's, would see no device)
    import torch  # noqa: F401
    path = library_path()
    if not path.exists():
        raise RuntimeError(
            f"mi355q: HIP library not found at {path}. Build it with "
            "`make -C llm-mixed-q_amd/csrc` (hipcc --offload-arch=gfx950); there is no CPU fallback.")
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    got = lib.mi355q_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"mi355q: ABI version mismatch (library {got}, binding {ABI_VERSION})")
    _LIB = lib
    return lib


E_BADARG, E_UNSUPPORTED, E_ALIGN = -1, -2, -3          # include/mi355q.h


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load_library().mi355q_error_string(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
