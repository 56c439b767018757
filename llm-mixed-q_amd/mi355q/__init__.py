"""mi355q -- MI355X-native block-quantised linear / matmul path.

Host side of the path (Python on PyTorch-ROCm) over the C-ABI library ``libmi355q.so``
(include/mi355q.h).  ``mi355q.quantize`` mirrors the reference's
``llm_mixed_q.models.quantize`` package: same registries, same getters, same class and
function contracts, so the reference's OPT / Llama / BERT model files and search harness
can import it in place of the original.

There is no CPU fallback: every quantiser and GEMM call runs the HIP kernels or raises.
"""
from . import _lib  # noqa: F401  (import errors surface early and loudly)
from ._lib import library_path, load_library

__all__ = ["library_path", "load_library"]
