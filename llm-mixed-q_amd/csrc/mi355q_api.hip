// C-ABI entry points declared in include/mi355q.h: argument validation and dispatch only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <utility>
#include <vector>

#include "mi355q.h"
#include <map>
#include <mutex>

#include "mi355q_internal.h"
#include "mi355q_align_row.h"

using namespace mi355q;

namespace {
std::atomic<int> g_gemm_variant{0};

// optional HIP-event bracket around the main GEMM kernel (benchmarks).  Round 6: the events are offered to the launcher
// (g_kernel_events); the 256 x 256 tile kernel's launcher attaches them to its dispatch (hipExtLaunchKernelGGL) -- kernel start to kernel
// end, as the profiler's kernel trace has it -- and every other launcher leaves them: then two marker records bracket the launch as
// before (which also times the command processor's way from marker to dispatch to marker: 5-7 us on round 6's boxes)
struct GemmTiming {
    bool enabled = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    size_t used = 0;
    hipEvent_t begin(hipStream_t st) {
        if (!enabled || used >= 4096) return nullptr;
        if (used == pool.size()) {
            hipEvent_t a, b;
            // (device-scope release: a default event record releases to SYSTEM scope -- the stop marker then waits for the write-back of
            //  the kernel's 64 MiB of output to leave the L2s before it takes its timestamp, 5-7 us that are not the kernel's and that
            //  the step without events never pays; HIP documents this flag for "more precise timings of commands between events")
            if (hipEventCreateWithFlags(&a, hipEventReleaseToDevice) != hipSuccess || hipEventCreateWithFlags(&b, hipEventReleaseToDevice) != hipSuccess)
                return nullptr;
            pool.emplace_back(a, b);
        }
        (void)hipEventRecord(pool[used].first, st);
        mi355q::g_kernel_events = {pool[used].first, pool[used].second};
        return pool[used].second;
    }
    void end(hipEvent_t e, hipStream_t st) {
        if (e) {
            if (mi355q::g_kernel_events.start) (void)hipEventRecord(e, st);      // (no launcher took them: the marker pair)
            mi355q::g_kernel_events = {nullptr, nullptr};
            ++used;
        }
    }
} g_timing;

}  // namespace
namespace mi355q { KernelEvents g_kernel_events = {nullptr, nullptr}; }
namespace {
bool bad_shape(int64_t lead, int64_t rows, int64_t cols, int32_t b0, int32_t b1) {
    return lead < 0 || rows < 0 || cols < 0 || b0 < 1 || b1 < 1;
}

int fill_common(QuantArgs& a, const float* x, float* y, void* workspace, int64_t lead, int64_t rows,
                int64_t cols, int32_t b0, int32_t b1, uint32_t flags) {
    if (bad_shape(lead, rows, cols, b0, b1)) return MI355Q_E_BADARG;
    a = QuantArgs{};
    a.x = x;
    a.y = y;
    a.ws = static_cast<unsigned*>(workspace);
    a.lead = lead; a.rows = rows; a.cols = cols;
    a.b0 = b0; a.b1 = b1;
    a.n_elems = lead * rows * cols;
    a.nbr = (rows + b0 - 1) / b0;
    a.nbc = (cols + b1 - 1) / b1;
    a.n_blocks = lead * a.nbr * a.nbc;
    a.flags = flags;
    if (a.n_elems == 0) return 1 << 30;   // nothing to do (caller returns 0)
    if (x == nullptr) return MI355Q_E_BADARG;
    if ((flags & MI355Q_ZERO_BLOCK_FAST) == 0u && workspace == nullptr) return MI355Q_E_BADARG;
    return 0;
}
void set_mantissa(QuantArgs& a, int mbits) {
    a.shift = std::ldexp(1.0f, mbits);
    a.inv_shift = std::ldexp(1.0f, -mbits);
    a.mant_max = a.shift - 1.0f;
}
}  // namespace

extern "C" {

int mi355q_abi_version(void) { return MI355Q_ABI_VERSION; }

const char* mi355q_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case MI355Q_E_BADARG: return "mi355q: bad argument (null pointer, negative size, or width out of range)";
        case MI355Q_E_UNSUPPORTED: return "mi355q: configuration valid for the reference but not built here";
        case MI355Q_E_ALIGN: return "mi355q: pointer or leading-dimension alignment requirement violated";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "mi355q: unknown error";
    }
}

int mi355q_block_fp_quantize(const float* x, float* y, int8_t* mant, uint8_t* exp, int64_t lead, int64_t rows,
                             int64_t cols, int32_t b0, int32_t b1, int32_t width, int32_t exponent_width,
                             int32_t exponent_bias, uint32_t flags, void* workspace, void* stream) {
    QuantArgs a;
    const int rc = fill_common(a, x, y, workspace, lead, rows, cols, b0, b1, flags);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    if ((mant == nullptr) != (exp == nullptr)) return MI355Q_E_BADARG;
    if (y == nullptr && mant == nullptr) return MI355Q_E_BADARG;
    if (exponent_width < 1 || exponent_width > 8) return MI355Q_E_BADARG;
    if (width < 2 || width > (mant ? 8 : 25)) return MI355Q_E_BADARG;
    if (exponent_bias == MI355Q_BIAS_DEFAULT) exponent_bias = (1 << (exponent_width - 1)) - 1;
    a.mant = mant;
    a.code = exp;
    a.code_bias = exponent_bias;
    a.e_min = -exponent_bias;
    a.e_max = (1 << exponent_width) - 1 - exponent_bias;
    set_mantissa(a, width - 1);
    return launch_quant(a, 0, /*needs_fixup=*/mant != nullptr, static_cast<hipStream_t>(stream));
}

int mi355q_block_fp_quantize_bf16(const float* x, uint16_t* y, int64_t lead, int64_t rows, int64_t cols, int32_t b0,
                                  int32_t b1, int32_t width, int32_t exponent_width, int32_t exponent_bias,
                                  void* workspace, void* stream) {
    QuantArgs a;
    const int rc = fill_common(a, x, nullptr, workspace, lead, rows, cols, b0, b1, MI355Q_ZERO_BLOCK_FAST);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    if (y == nullptr) return MI355Q_E_BADARG;
    if (exponent_width < 1 || exponent_width > 8 || width < 2) return MI355Q_E_BADARG;
    if (width > 9) return MI355Q_E_UNSUPPORTED;            // a quantised value must fit bf16's 8 significant bits
    if (exponent_bias == MI355Q_BIAS_DEFAULT) exponent_bias = (1 << (exponent_width - 1)) - 1;
    a.ybf = y;
    a.code_bias = exponent_bias;
    a.e_min = -exponent_bias;
    a.e_max = (1 << exponent_width) - 1 - exponent_bias;
    set_mantissa(a, width - 1);
    return launch_quant(a, 0, /*needs_fixup=*/false, static_cast<hipStream_t>(stream));
}

static bool pre_op_ok(int32_t pre_op, const float* x2, bool rows_kernel = false) {
    const bool norm = pre_op == MI355Q_PRE_RMSNORM || pre_op == MI355Q_PRE_LAYERNORM;
    if (norm && !rows_kernel) return false;                              // (needs the whole row in one workgroup)
    if (pre_op != MI355Q_PRE_NONE && pre_op != MI355Q_PRE_RELU && pre_op != MI355Q_PRE_SILU_MUL && !norm) return false;
    return (pre_op != MI355Q_PRE_SILU_MUL && !norm) || (x2 != nullptr && reinterpret_cast<uintptr_t>(x2) % 16 == 0);
}

int mi355q_block_fp_quantize_bf16_tiled(const float* x, float* y, uint16_t* y_tiled, int64_t rows, int64_t K, int32_t width,
                                        int32_t exponent_width, int32_t exponent_bias, void* workspace, void* stream) {
    return mi355q_block_fp_quantize_bf16_tiled_pre(x, nullptr, MI355Q_PRE_NONE, y, y_tiled, rows, K, width, exponent_width,
                                                   exponent_bias, workspace, stream);
}

int mi355q_block_fp_quantize_bf16_tiled_pre(const float* x, const float* x2, int32_t pre_op, float* y, uint16_t* y_tiled,
                                            int64_t rows, int64_t K, int32_t width, int32_t exponent_width,
                                            int32_t exponent_bias, void* workspace, void* stream) {
    if (pre_op == MI355Q_PRE_RMSNORM) return MI355Q_E_BADARG;            // (mi355q_block_fp_quantize_bf16_tiled_norm)
    return mi355q_block_fp_quantize_bf16_tiled_norm(x, x2, pre_op, 0.f, y, y_tiled, rows, K, width, exponent_width, exponent_bias,
                                                    workspace, stream);
}

int mi355q_block_fp_quantize_bf16_tiled_norm(const float* x, const float* x2, int32_t pre_op, float eps, float* y, uint16_t* y_tiled,
                                             int64_t rows, int64_t K, int32_t width, int32_t exponent_width,
                                             int32_t exponent_bias, void* workspace, void* stream) {
    if (pre_op == MI355Q_PRE_LAYERNORM || !pre_op_ok(pre_op, x2, pre_op == MI355Q_PRE_RMSNORM) || !(eps >= 0.f)) return MI355Q_E_BADARG;
    QuantArgs a;
    const int rc = fill_common(a, x, y, workspace, 1, rows, K, 1, 16, MI355Q_ZERO_BLOCK_FAST);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    a.x2 = x2;
    a.pre_op = pre_op;
    a.pre_eps = eps;
    if (y_tiled == nullptr) return MI355Q_E_BADARG;
    if (exponent_width < 1 || exponent_width > 8 || width < 2) return MI355Q_E_BADARG;
    if (width > 9 || K % 32 != 0) return MI355Q_E_UNSUPPORTED;   // bf16's 8 significant bits; whole 64-byte K-steps
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y_tiled)) % 16) return MI355Q_E_ALIGN;
    if (exponent_bias == MI355Q_BIAS_DEFAULT) exponent_bias = (1 << (exponent_width - 1)) - 1;
    a.code_bias = exponent_bias;
    a.e_min = -exponent_bias;
    a.e_max = (1 << exponent_width) - 1 - exponent_bias;
    set_mantissa(a, width - 1);
    return launch_quant_bf16_tiled(a, y_tiled, static_cast<hipStream_t>(stream));
}

int mi355q_block_minifloat_quantize_bf16_tiled(const float* x, uint16_t* y_tiled, int64_t rows, int64_t K, int32_t width,
                                               int32_t exponent_width, int32_t exponent_bias_width, void* workspace,
                                               void* stream) {
    QuantArgs a;
    const int rc = fill_common(a, x, nullptr, workspace, 1, rows, K, 1, 16, MI355Q_ZERO_BLOCK_FAST);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    if (y_tiled == nullptr) return MI355Q_E_BADARG;
    const int mbits = width - exponent_width - 1;
    if (exponent_width < 1 || exponent_width > 8 || mbits < 0 || mbits > 23) return MI355Q_E_BADARG;
    if (exponent_bias_width < 1 || exponent_bias_width > 8) return MI355Q_E_BADARG;
    if (mbits > 7 || K % 32 != 0) return MI355Q_E_UNSUPPORTED;      // bf16's 8 significant bits; whole 64-byte K-steps
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y_tiled)) % 16) return MI355Q_E_ALIGN;
    a.span = (1 << exponent_width) - 1;
    a.bias_max = (1 << exponent_bias_width) - 1;
    set_mantissa(a, mbits);
    return launch_quant_bf16_tiled(a, y_tiled, static_cast<hipStream_t>(stream), false, 1);
}

size_t mi355q_bfp_packed_bytes(int64_t rows, int64_t K, int32_t width) {
    return (rows <= 0 || K <= 0 || width < 2 || width > 8) ? 0 : (size_t)rows * (size_t)(K / 16) * (size_t)width * 2;
}

int mi355q_bfp_pack_bits(const int8_t* mant, uint8_t* packed, int64_t rows, int64_t K, int32_t width, void* stream) {
    if (rows < 0 || K < 0 || width < 2 || width > 8) return MI355Q_E_BADARG;
    if (rows == 0 || K == 0) return 0;
    if (!mant || !packed) return MI355Q_E_BADARG;
    if (K % 16 != 0) return MI355Q_E_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(mant) % 16 || reinterpret_cast<uintptr_t>(packed) % 2) return MI355Q_E_ALIGN;
    return launch_bfp_pack_bits(mant, reinterpret_cast<uint16_t*>(packed), rows, K, width, static_cast<hipStream_t>(stream));
}

int mi355q_bfp_expand(const uint8_t* packed, const uint8_t* codes, void* out_tiled, int64_t rows, int64_t K, int32_t width,
                      int32_t mode, int32_t exp_offset, const uint8_t* row_exp, uint8_t* exp_out, void* stream) {
    if (rows < 0 || K < 0 || width < 2 || width > 8 || (mode != 0 && mode != 1)) return MI355Q_E_BADARG;
    if (rows == 0 || K == 0) return 0;
    if (!packed || !codes || !out_tiled || ((exp_out != nullptr) != (row_exp != nullptr))) return MI355Q_E_BADARG;
    if (K % (mode == 0 ? 64 : 32) != 0) return MI355Q_E_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(packed) % 2 || reinterpret_cast<uintptr_t>(out_tiled) % 16) return MI355Q_E_ALIGN;
    return launch_bfp_expand(mode, reinterpret_cast<const uint16_t*>(packed), codes, out_tiled, rows, K, width, exp_offset,
                             static_cast<hipStream_t>(stream), row_exp, exp_out);
}

int mi355q_bf16_tile(const float* x, uint16_t* y_tiled, int64_t rows, int64_t K, void* stream) {
    QuantArgs a;
    const int rc = fill_common(a, x, nullptr, nullptr, 1, rows, K, 1, 16, MI355Q_ZERO_BLOCK_FAST);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    if (y_tiled == nullptr) return MI355Q_E_BADARG;
    if (K % 32 != 0) return MI355Q_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y_tiled)) % 16) return MI355Q_E_ALIGN;
    a.code_bias = 127; a.e_min = -127; a.e_max = 128;
    set_mantissa(a, 7);
    return launch_quant_bf16_tiled(a, y_tiled, static_cast<hipStream_t>(stream), /*cast_only=*/true);
}

int mi355q_fp32_split_tile(const float* x, uint16_t* y_tiled, int64_t rows, int64_t K, int32_t role, void* stream) {
    if (rows < 0 || K < 0 || (role != 0 && role != 1)) return MI355Q_E_BADARG;
    if (rows == 0 || K == 0) return 0;
    if (!x || !y_tiled) return MI355Q_E_BADARG;
    if (K % 32 != 0) return MI355Q_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y_tiled)) % 16) return MI355Q_E_ALIGN;
    return launch_fp32_split_tile(x, y_tiled, rows, K, role, static_cast<hipStream_t>(stream));
}

static int bf16_gemm_tiled_impl(const uint16_t* x_tiled, const uint16_t* w_tiled, const float* bias, float* y, int64_t M, int64_t N,
                                int64_t K, int64_t ldy, int32_t x_segments, int64_t x_segment_stride_bytes, void* stream,
                                const float* residual = nullptr, int64_t ldr = 0);

int mi355q_bf16_gemm_tiled(const uint16_t* x_tiled, const uint16_t* w_tiled, const float* bias, float* y, int64_t M, int64_t N,
                           int64_t K, int64_t ldy, void* stream) {
    return bf16_gemm_tiled_impl(x_tiled, w_tiled, bias, y, M, N, K, ldy, 1, 0, stream);
}

int mi355q_bf16_gemm_tiled_res(const uint16_t* x_tiled, const uint16_t* w_tiled, const float* bias, const float* residual, int64_t ldr,
                               float* y, int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream) {
    if (!residual || ldr < N || ldr % 4 != 0) return MI355Q_E_BADARG;
    if (reinterpret_cast<uintptr_t>(residual) % 16) return MI355Q_E_ALIGN;
    return bf16_gemm_tiled_impl(x_tiled, w_tiled, bias, y, M, N, K, ldy, 1, 0, stream, residual, ldr);
}

int mi355q_bf16_gemm_tiled_seg(const uint16_t* x_tiled, const uint16_t* w_tiled, const float* bias, float* y, int64_t M, int64_t N,
                               int64_t K, int64_t ldy, int32_t x_segments, int64_t x_segment_stride_bytes, void* stream) {
    if (x_segments < 1 || (x_segments > 1 && (x_segment_stride_bytes <= 0 || x_segment_stride_bytes % 16))) return MI355Q_E_BADARG;
    if (x_segments > 1 && K > 0 && (K % (32 * (int64_t)x_segments) != 0)) return MI355Q_E_UNSUPPORTED;   // whole 64-byte K-steps per segment
    // (a stride shorter than one tiled segment would make the kernel read overlapping segments: wrong results with rc 0)
    if (x_segments > 1 && M > 0 && K > 0 && x_segment_stride_bytes < (int64_t)mi355q_bfp_tiled_bytes(M, 2 * K / x_segments)) return MI355Q_E_BADARG;
    return bf16_gemm_tiled_impl(x_tiled, w_tiled, bias, y, M, N, K, ldy, x_segments, x_segment_stride_bytes, stream);
}

static int bf16_gemm_tiled_impl(const uint16_t* x_tiled, const uint16_t* w_tiled, const float* bias, float* y, int64_t M, int64_t N,
                                int64_t K, int64_t ldy, int32_t x_segments, int64_t x_segment_stride_bytes, void* stream,
                                const float* residual, int64_t ldr) {
    if (M < 0 || N < 0 || K < 0 || ldy < N) return MI355Q_E_BADARG;
    if (M == 0 || N == 0) return 0;
    if (!y || (K > 0 && (!x_tiled || !w_tiled))) return MI355Q_E_BADARG;
    if (K % 32 != 0 || K == 0) return MI355Q_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x_tiled) | reinterpret_cast<uintptr_t>(w_tiled)) % 16) return MI355Q_E_ALIGN;
    GemmArgs a{};
    a.xm = reinterpret_cast<const int8_t*>(x_tiled);
    a.wm = reinterpret_cast<const int8_t*>(w_tiled);
    a.bias = bias;
    a.y = y;
    a.M = M; a.N = N; a.K = 2 * K; a.ldy = ldy;      // (the tile kernel counts the contraction in bytes)
    a.x_segs = x_segments;
    a.x_seg_stride = x_segment_stride_bytes;
    a.resid = residual;
    a.ldr = ldr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t te = g_timing.begin(st);
    const int rc = launch_bf16_gemm_tiled(a, st);
    g_timing.end(te, st);
    return rc;
}

int mi355q_block_minifloat_quantize(const float* x, float* y, uint8_t* bias, int64_t lead, int64_t rows, int64_t cols,
                                    int32_t b0, int32_t b1, int32_t width, int32_t exponent_width,
                                    int32_t exponent_bias_width, uint32_t flags, void* workspace, void* stream) {
    QuantArgs a;
    const int rc = fill_common(a, x, y, workspace, lead, rows, cols, b0, b1, flags);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    if (y == nullptr) return MI355Q_E_BADARG;
    const int mbits = width - exponent_width - 1;
    if (exponent_width < 1 || exponent_width > 8 || mbits < 0 || mbits > 23) return MI355Q_E_BADARG;
    if (exponent_bias_width < 1 || exponent_bias_width > 8) return MI355Q_E_BADARG;
    a.code = bias;
    a.span = (1 << exponent_width) - 1;
    a.bias_max = (1 << exponent_bias_width) - 1;
    set_mantissa(a, mbits);
    return launch_quant(a, 1, /*needs_fixup=*/bias != nullptr, static_cast<hipStream_t>(stream));
}

int mi355q_block_log_quantize(const float* x, float* y, uint8_t* bias, int64_t lead, int64_t rows, int64_t cols,
                              int32_t b0, int32_t b1, int32_t width, int32_t exponent_bias_width, uint32_t flags,
                              void* workspace, void* stream) {
    QuantArgs a;
    const int rc = fill_common(a, x, y, workspace, lead, rows, cols, b0, b1, flags);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    if (y == nullptr) return MI355Q_E_BADARG;
    if (width < 2 || width > 9) return MI355Q_E_BADARG;       // exponent code of width-1 <= 8 bits
    if (exponent_bias_width < 1 || exponent_bias_width > 8) return MI355Q_E_BADARG;
    a.code = bias;
    a.span = (1 << (width - 1)) - 1;
    a.bias_max = (1 << exponent_bias_width) - 1;
    set_mantissa(a, 0);
    return launch_quant(a, 2, /*needs_fixup=*/true, static_cast<hipStream_t>(stream));
}

// the block_minifloat / block_log fake-quantised values straight to bf16 (exact: <= 7 mantissa bits / signed powers of two;
// elements with |x| <= 1e-8, which the reference passes through, are rounded to bf16): the operands of a bf16-MFMA product
int mi355q_block_minifloat_quantize_bf16(const float* x, uint16_t* y, int64_t lead, int64_t rows, int64_t cols, int32_t b0,
                                         int32_t b1, int32_t width, int32_t exponent_width, int32_t exponent_bias_width,
                                         void* workspace, void* stream) {
    QuantArgs a;
    const int rc = fill_common(a, x, nullptr, workspace, lead, rows, cols, b0, b1, MI355Q_ZERO_BLOCK_FAST);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    if (y == nullptr) return MI355Q_E_BADARG;
    const int mbits = width - exponent_width - 1;
    if (exponent_width < 1 || exponent_width > 8 || mbits < 0) return MI355Q_E_BADARG;
    if (mbits > 7) return MI355Q_E_UNSUPPORTED;              // a quantised value must fit bf16's 8 significant bits
    if (exponent_bias_width < 1 || exponent_bias_width > 8) return MI355Q_E_BADARG;
    a.ybf = y;
    a.span = (1 << exponent_width) - 1;
    a.bias_max = (1 << exponent_bias_width) - 1;
    set_mantissa(a, mbits);
    return launch_quant(a, 1, /*needs_fixup=*/false, static_cast<hipStream_t>(stream));     // (an all-zero block stays zero)
}

int mi355q_block_log_quantize_bf16(const float* x, uint16_t* y, int64_t lead, int64_t rows, int64_t cols, int32_t b0,
                                   int32_t b1, int32_t width, int32_t exponent_bias_width, void* workspace, void* stream) {
    QuantArgs a;
    const int rc = fill_common(a, x, nullptr, workspace, lead, rows, cols, b0, b1, 0u);
    if (rc == (1 << 30)) return 0;
    if (rc) return rc;
    if (y == nullptr) return MI355Q_E_BADARG;
    if (width < 2 || width > 9) return MI355Q_E_BADARG;
    if (exponent_bias_width < 1 || exponent_bias_width > 8) return MI355Q_E_BADARG;
    // (all-zero blocks take the tensor-global fill -- a non-zero value here: the fix-up pass writes bf16 in its row-block form only)
    if (b0 != 1 || b1 != 16 || cols % 16 || reinterpret_cast<uintptr_t>(y) % 16) return MI355Q_E_UNSUPPORTED;
    a.ybf = y;
    a.span = (1 << (width - 1)) - 1;
    a.bias_max = (1 << exponent_bias_width) - 1;
    set_mantissa(a, 0);
    return launch_quant(a, 2, /*needs_fixup=*/true, static_cast<hipStream_t>(stream));
}

int mi355q_minifloat_quantize(const float* x, float* y, int64_t n, int32_t width, int32_t exponent_width, int32_t exponent_bias,
                              int32_t denorm, void* stream) {
    if (n < 0) return MI355Q_E_BADARG;
    if (n == 0) return 0;
    if (!x || !y) return MI355Q_E_BADARG;
    const int mbits = width - exponent_width - 1;
    if (exponent_width < 1 || exponent_width > 8 || mbits < 0 || mbits > 23) return MI355Q_E_BADARG;
    if (exponent_bias == MI355Q_BIAS_DEFAULT) exponent_bias = (1 << (exponent_width - 1)) - 1;
    if (exponent_bias < -100 || exponent_bias > 140) return MI355Q_E_UNSUPPORTED;      // (exponents must stay inside fp32's)
    QuantArgs a{};
    a.x = x; a.y = y; a.n_elems = n;
    a.span = (1 << exponent_width) - 1;                       // ieee: e in [-bias, span - bias]
    a.e_min = -exponent_bias;                                 // denorm: the same range
    a.e_max = a.span - exponent_bias;
    set_mantissa(a, mbits);
    return launch_quant_flat(a, denorm ? 3 : 1, exponent_bias, static_cast<hipStream_t>(stream));
}

int mi355q_log_quantize(const float* x, float* y, int64_t n, int32_t width, int32_t exponent_bias, void* stream) {
    if (n < 0 || width < 2 || width > 9) return MI355Q_E_BADARG;      // exponent code of width - 1 <= 8 bits
    if (n == 0) return 0;
    if (!x || !y) return MI355Q_E_BADARG;
    if (exponent_bias == MI355Q_BIAS_DEFAULT) exponent_bias = (1 << (width - 2)) - 1;
    if (exponent_bias < -100 || exponent_bias > 140) return MI355Q_E_UNSUPPORTED;
    QuantArgs a{};
    a.x = x; a.y = y; a.n_elems = n;
    a.span = (1 << (width - 1)) - 1;
    set_mantissa(a, 0);
    return launch_quant_flat(a, 2, exponent_bias, static_cast<hipStream_t>(stream));
}

int mi355q_integer_quantize(const float* x, float* y, int64_t n, int32_t width, int32_t frac_width,
                            int32_t is_signed, void* stream) {
    if (n < 0 || width < 1 || width > 24) return MI355Q_E_BADARG;
    if (n == 0) return 0;
    if (x == nullptr || y == nullptr) return MI355Q_E_BADARG;
    const float lo = is_signed ? -std::ldexp(1.0f, width - 1) : 0.0f;
    const float hi = is_signed ? std::ldexp(1.0f, width - 1) - 1.0f : std::ldexp(1.0f, width) - 1.0f;
    return launch_integer(x, y, n, std::ldexp(1.0f, frac_width), lo, hi, static_cast<hipStream_t>(stream));
}

int mi355q_bfp_gemm(const int8_t* xm, const uint8_t* xe, const int8_t* wm, const uint8_t* we, const float* bias,
                    float* y, int64_t M, int64_t N, int64_t K, int64_t ldy, int32_t x_mbits, int32_t x_exp_bias,
                    int32_t w_mbits, int32_t w_exp_bias, void* stream) {
    if (M < 0 || N < 0 || K < 0 || ldy < N) return MI355Q_E_BADARG;
    if (M == 0 || N == 0) return 0;
    if (!y || (K > 0 && (!xm || !xe || !wm || !we))) return MI355Q_E_BADARG;
    if (K % 16 != 0) return MI355Q_E_UNSUPPORTED;
    if (x_mbits < 1 || x_mbits > 7 || w_mbits < 1 || w_mbits > 7) return MI355Q_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(xm) | reinterpret_cast<uintptr_t>(wm)) % 16) return MI355Q_E_ALIGN;
    GemmArgs a{xm, xe, wm, we, bias, y, M, N, K, ldy, x_exp_bias + x_mbits + w_exp_bias + w_mbits};
    return launch_bfp_gemm(a, g_gemm_variant.load(), static_cast<hipStream_t>(stream));
}

size_t mi355q_bfp_rowflag_bytes(int64_t rows, int64_t K) {
    if (rows <= 0 || K <= 0) return 0;
    return static_cast<size_t>(rows) * static_cast<size_t>((K + 255) / 256);
}

int64_t mi355q_bfp_rows_pad(int64_t rows) { return rows <= 0 ? 0 : (rows + 255) / 256 * 256 + 256; }


size_t mi355q_bfp_tiled_bytes(int64_t rows, int64_t K) {
    if (rows <= 0 || K <= 0) return 0;
    return static_cast<size_t>((rows + 127) / 128 * 128) * static_cast<size_t>((K + 63) / 64 * 64);
}




static int bucket_cap_of(int32_t cap) { return cap == 0 ? ROW_BCAP : cap; }

size_t mi355q_bfp_row_list_bytes(int64_t rows, int32_t bucket_cap) {
    if (rows < 0 || bucket_cap < 0 || bucket_cap > ROW_BCAP_MAX) return 0;
    return (size_t)row_list_words(rows, bucket_cap_of(bucket_cap)) * 4;
}

int mi355q_bfp_align_rows(const int8_t* mant_in, const uint8_t* exp_in, int8_t* mant_tiled, uint8_t* exp_out,
                          uint8_t* rowflag, float* rowscale, int32_t* list, int32_t exp_offset, int64_t rows, int64_t K,
                          int32_t bucket_cap, void* stream) {
    if (rows < 0 || K < 0 || bucket_cap < 0 || bucket_cap > ROW_BCAP_MAX) return MI355Q_E_BADARG;
    if (rows == 0 || K == 0) return 0;
    if (!mant_in || !exp_in || !mant_tiled || !exp_out || !rowflag || !rowscale) return MI355Q_E_BADARG;
    if (K % 64 != 0 || K > MI355Q_ROW_ALIGN_MAX_K) return MI355Q_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(mant_in) | reinterpret_cast<uintptr_t>(mant_tiled)) % 4) return MI355Q_E_ALIGN;
    return launch_bfp_align_rows(mant_in, exp_in, mant_tiled, exp_out, rowflag, rowscale, exp_offset, list, rows, K,
                                 static_cast<hipStream_t>(stream), bucket_cap_of(bucket_cap));
}

int mi355q_block_fp_quantize_aligned_rows(const float* x, int8_t* mant_tiled, uint8_t* exp_out, uint8_t* rowflag,
                                          float* rowscale, int32_t* list, int32_t* list_to_clear, int64_t rows,
                                          int64_t K, int32_t width, int32_t exponent_width, int32_t exponent_bias,
                                          int32_t bucket_cap, void* stream) {
    return mi355q_block_fp_quantize_aligned_rows_pre(x, nullptr, MI355Q_PRE_NONE, mant_tiled, exp_out, rowflag, rowscale, list,
                                                     list_to_clear, rows, K, width, exponent_width, exponent_bias, bucket_cap,
                                                     stream);
}

int mi355q_block_fp_quantize_aligned_rows_pre(const float* x, const float* x2, int32_t pre_op, int8_t* mant_tiled,
                                              uint8_t* exp_out, uint8_t* rowflag, float* rowscale, int32_t* list,
                                              int32_t* list_to_clear, int64_t rows, int64_t K, int32_t width,
                                              int32_t exponent_width, int32_t exponent_bias, int32_t bucket_cap,
                                              void* stream) {
    if (pre_op == MI355Q_PRE_LAYERNORM) return MI355Q_E_BADARG;          // (mi355q_block_fp_quantize_aligned_rows_norm)
    return mi355q_block_fp_quantize_aligned_rows_norm(x, x2, nullptr, pre_op, 0.f, mant_tiled, exp_out, rowflag, rowscale, list,
                                                      list_to_clear, rows, K, width, exponent_width, exponent_bias, bucket_cap,
                                                      stream);
}

int mi355q_block_fp_quantize_aligned_rows_norm(const float* x, const float* x2, const float* x3, int32_t pre_op, float eps,
                                               int8_t* mant_tiled,
                                               uint8_t* exp_out, uint8_t* rowflag, float* rowscale, int32_t* list,
                                               int32_t* list_to_clear, int64_t rows, int64_t K, int32_t width,
                                               int32_t exponent_width, int32_t exponent_bias, int32_t bucket_cap,
                                               void* stream) {
    return mi355q_block_fp_quantize_aligned_rows_seg(x, x2, x3, pre_op, eps, mant_tiled, exp_out, rowflag, rowscale, list,
                                                     list_to_clear, rows, K, 0, 0, width, exponent_width, exponent_bias,
                                                     bucket_cap, stream);
}

static int quantize_aligned_rows_impl(const float* x, const float* x2, const float* x3, int32_t pre_op, float eps,
                                      int8_t* mant_tiled, uint8_t* exp_out, uint8_t* rowflag, float* rowscale,
                                      int32_t* list, int32_t* list_to_clear, int64_t rows, int64_t K, int64_t seg_len,
                                      int64_t seg_stride, int32_t width, int32_t exponent_width, int32_t exponent_bias,
                                      int32_t bucket_cap, void* stream) {
    if (seg_len < 0 || seg_stride < 0) return MI355Q_E_BADARG;
    if (seg_len > 0 && (seg_len % 4 || seg_stride % 4 || K % seg_len || seg_stride < rows * seg_len)) return MI355Q_E_BADARG;
    if (!pre_op_ok(pre_op, x2, true) || !(eps >= 0.f) || reinterpret_cast<uintptr_t>(x3) % 16) return MI355Q_E_BADARG;
    if (rows < 0 || K < 0 || bucket_cap < MI355Q_ROW_NO_ALIGN || bucket_cap > ROW_BCAP_MAX) return MI355Q_E_BADARG;
    if (rows == 0 || K == 0) return 0;
    if (!x || !mant_tiled || !exp_out || !rowflag || !rowscale || (!list && bucket_cap >= 0) || (list && list_to_clear == list))
        return MI355Q_E_BADARG;
    if (K % 64 != 0 || K > MI355Q_ROW_ALIGN_MAX_K) return MI355Q_E_UNSUPPORTED;
    if (exponent_width < 1 || exponent_width > 8 || width < 2 || width > 8) return MI355Q_E_BADARG;
    if (reinterpret_cast<uintptr_t>(x) % 16 || reinterpret_cast<uintptr_t>(mant_tiled) % 16) return MI355Q_E_ALIGN;
    if (exponent_bias == MI355Q_BIAS_DEFAULT) exponent_bias = (1 << (exponent_width - 1)) - 1;
    if (exponent_bias < 0) return MI355Q_E_UNSUPPORTED;          // biased uint8 exponent codes: non-negative biases only
    QuantArgs a{};
    a.x = x;
    a.x2 = x2;
    a.pre_op = pre_op;
    a.pre_eps = eps;
    a.x3 = pre_op == MI355Q_PRE_LAYERNORM ? x3 : nullptr;
    a.code = exp_out;
    a.lead = 1; a.rows = rows; a.cols = K;
    a.b0 = 1; a.b1 = 16;
    a.n_elems = rows * K;
    a.nbr = rows; a.nbc = K / 16;
    a.n_blocks = rows * (K / 16);
    a.flags = MI355Q_ZERO_BLOCK_FAST;
    a.code_bias = exponent_bias;
    a.e_min = -exponent_bias;
    a.e_max = (1 << exponent_width) - 1 - exponent_bias;
    a.seg_len = seg_len == K ? 0 : seg_len;                 // (one segment: plain rows)
    a.seg_stride = seg_stride;
    set_mantissa(a, width - 1);
    return launch_quant_align_rows(a, mant_tiled, rowflag, rowscale, exponent_bias + width - 1, list, list_to_clear,
                                   static_cast<hipStream_t>(stream), bucket_cap < 0 ? -1 : bucket_cap_of(bucket_cap));
}

int mi355q_block_fp_quantize_aligned_rows_seg(const float* x, const float* x2, const float* x3, int32_t pre_op, float eps,
                                              int8_t* mant_tiled, uint8_t* exp_out, uint8_t* rowflag, float* rowscale,
                                              int32_t* list, int32_t* list_to_clear, int64_t rows, int64_t K, int64_t seg_len,
                                              int64_t seg_stride, int32_t width, int32_t exponent_width, int32_t exponent_bias,
                                              int32_t bucket_cap, void* stream) {
    return quantize_aligned_rows_impl(x, x2, x3, pre_op, eps, mant_tiled, exp_out, rowflag, rowscale, list, list_to_clear, rows, K,
                                      seg_len, seg_stride, width, exponent_width, exponent_bias, bucket_cap, stream);
}

// the class-aware activation quantiser of the mixed contraction (mi355q.h; mi355q_quant_cls.hip)
int mi355q_block_fp_quantize_classes(const float* x, const uint16_t* colmap, int64_t n0_blocks, int64_t n1_blocks, int8_t* mant_tiled,
                                     uint8_t* exp_out, uint8_t* rowflag, float* rowscale, int32_t* list, int32_t* list_to_clear,
                                     void* x1_bf16_tiled, int64_t rows, int64_t K, int32_t width, int32_t exponent_width,
                                     int32_t exponent_bias, int32_t bucket_cap, void* stream) {
    if (rows < 0 || K < 0 || n0_blocks < 0 || n1_blocks < 0 || bucket_cap < 0 || bucket_cap > ROW_BCAP_MAX) return MI355Q_E_BADARG;
    if (rows == 0 || K == 0) return 0;
    if (!x || !colmap || !mant_tiled || !exp_out || !rowflag || !rowscale || !list || !x1_bf16_tiled || list_to_clear == list)
        return MI355Q_E_BADARG;
    // whole 64-byte K-steps in both operands: class 0 a multiple of 4 blocks, class 1 a multiple of 2
    if (K % 64 != 0 || K > MI355Q_ROW_ALIGN_MAX_K || (n0_blocks + n1_blocks) * 16 != K || n0_blocks % 4 || n1_blocks % 2 || n0_blocks >= 32768 ||
        n1_blocks >= 32768)
        return MI355Q_E_UNSUPPORTED;
    if (exponent_width < 1 || exponent_width > 8 || width < 2 || width > 8) return MI355Q_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(mant_tiled) | reinterpret_cast<uintptr_t>(x1_bf16_tiled)) % 16) return MI355Q_E_ALIGN;
    if (exponent_bias == MI355Q_BIAS_DEFAULT) exponent_bias = (1 << (exponent_width - 1)) - 1;
    if (exponent_bias < 0) return MI355Q_E_UNSUPPORTED;
    QuantArgs a{};
    a.x = x;
    a.code = exp_out;
    a.lead = 1; a.rows = rows; a.cols = K;
    a.b0 = 1; a.b1 = 16;
    a.n_elems = rows * K;
    a.nbr = rows; a.nbc = K / 16;
    a.n_blocks = rows * (K / 16);
    a.flags = MI355Q_ZERO_BLOCK_FAST;
    a.code_bias = exponent_bias;
    a.e_min = -exponent_bias;
    a.e_max = (1 << exponent_width) - 1 - exponent_bias;
    set_mantissa(a, width - 1);
    return launch_quant_classes(a, colmap, (int)n0_blocks, (int)n1_blocks, mant_tiled, rowflag, rowscale, exponent_bias + width - 1, list,
                                list_to_clear, static_cast<uint16_t*>(x1_bf16_tiled), static_cast<hipStream_t>(stream),
                                bucket_cap_of(bucket_cap));
}

static int gemm_aligned_impl(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w, const float* bias, float* y,
                            int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream, const float* residual = nullptr, int64_t ldr = 0) {
    if (!x || !w || M < 0 || N < 0 || K < 0 || ldy < N) return MI355Q_E_BADARG;
    if (M == 0 || N == 0) return 0;
    if (!y || (K > 0 && (!x->mant || !x->exp || !w->mant || !w->exp || !x->rowflag || !w->rowflag)))
        return MI355Q_E_BADARG;
    if (K % 64 != 0) return MI355Q_E_UNSUPPORTED;   // tiled operands; use mi355q_bfp_gemm otherwise
    if (x->mbits < 1 || x->mbits > 7 || w->mbits < 1 || w->mbits > 7) return MI355Q_E_BADARG;
    // both operands in the same alignment flavour; x may also be in row format with NOTHING aligned (row_aligned = 2)
    if (x->row_aligned != w->row_aligned && !(x->row_aligned == 2 && w->row_aligned == 1)) return MI355Q_E_BADARG;
    if (!x->row_aligned && x->list && w->list && x->list_cap != w->list_cap) return MI355Q_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(x->mant) | reinterpret_cast<uintptr_t>(w->mant)) % 16) return MI355Q_E_ALIGN;
    GemmArgs a{x->mant, x->exp, w->mant, w->exp, bias, y, M, N, K, ldy,
               x->exp_bias + x->mbits + w->exp_bias + w->mbits, x->row_aligned ? 1 : 0,
               x->exp_bias + x->mbits, w->exp_bias + w->mbits,
               ROW_BCAP, ROW_BCAP, 0};
    const int variant = g_gemm_variant.load();
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (x->row_aligned) {
        // ROW-aligned operands: plain int8 GEMM with one scale per row + exception add-back in its epilogue; the
        // second launch only acts when an exception bucket overflowed (then it forms the whole product blockwise)
        if (!x->gscale || !w->gscale) return MI355Q_E_BADARG;
        if (w->list_cap < 0 || w->list_cap > ROW_BCAP_MAX) return MI355Q_E_BADARG;
        a.w_bcap = bucket_cap_of(w->list_cap);          // (before the early return: the kernel below indexes w's buckets)
        if (residual && x->row_aligned == 2) return MI355Q_E_UNSUPPORTED;      // (the residual add rides the one-launch route's stores only)
        if (x->row_aligned == 2)            // unaligned activations: the blockwise-exact kernel, w's exception blocks per tile
            return launch_bfp_gemm_aligned(a, x->rowflag, w->rowflag, nullptr, w->list, 0, 0, st);
        if (x->list_cap < 0 || x->list_cap > ROW_BCAP_MAX) return MI355Q_E_BADARG;
        a.x_bcap = bucket_cap_of(x->list_cap);
        // x's exception blocks: in-LDS vectors of the GEMM (120-entry buckets) or the row post-pass (larger buckets)
        a.x_post = a.x_bcap != ROW_BCAP ? 1 : 0;
        const bool fast_ok = x->list && w->list && K % 128 == 0 && K <= MI355Q_ROW_ALIGN_MAX_K && a.w_bcap == ROW_BCAP;
        if (residual && (variant == 2 || variant == 8 || !fast_ok || a.x_post)) return MI355Q_E_UNSUPPORTED;
        a.resid = residual;
        a.ldr = ldr;
        if (variant == 2 || !fast_ok)
            return launch_bfp_gemm_aligned(a, x->rowflag, w->rowflag, x->list, w->list, 0, 0, st);
        if (variant == 8) return launch_bfp_gemm_v8(a, x->gscale, w->gscale, nullptr, nullptr, 0, st);
        // ONE launch: the row-scale GEMM forms the correction vectors of its tile's exception blocks itself and adds them
        // in its epilogue; if an exception bucket overflowed anywhere its workgroups form the whole product
        // blockwise-exact between them instead (decided on the device)
        a.x_mbits = x->mbits;
        a.w_mbits = w->mbits;
        hipEvent_t te = g_timing.begin(st);
        int rc = launch_bfp_gemm_v8(a, x->gscale, w->gscale, x->list, w->list, 0, st, x->rowflag, w->rowflag);
        g_timing.end(te, st);
        if (rc == 0 && a.x_post) rc = launch_bfp_gemm_rowpost(a, x->list, w->list, x->gscale, w->gscale, st);
        return rc;
    }
    return MI355Q_E_UNSUPPORTED;      // (operands aligned per 256-value group: the format went with its kernel in round 5)
}

int mi355q_bfp_gemm_aligned(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w, const float* bias, float* y,
                            int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream) {
    return gemm_aligned_impl(x, w, bias, y, M, N, K, ldy, stream);
}

int mi355q_bfp_gemm_aligned_res(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w, const float* bias, const float* residual,
                                int64_t ldr, float* y, int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream) {
    if (!residual || ldr < N || ldr % 4 != 0) return MI355Q_E_BADARG;
    if (reinterpret_cast<uintptr_t>(residual) % 16) return MI355Q_E_ALIGN;
    return gemm_aligned_impl(x, w, bias, y, M, N, K, ldy, stream, residual, ldr);
}

// y = x . w^T + bias with the contraction in two column classes (mi355q.h): class 0 = K0 values as row-aligned int8 operands with
// their exception lists (what mi355q_bfp_gemm_aligned multiplies), class 1 = K1 values as tiled bf16 operands (what
// mi355q_bf16_gemm_tiled multiplies) -- one launch of the 256 x 256 tile kernel
int mi355q_bfp_gemm_mixed(const mi355q_bfp_operand* x0, const mi355q_bfp_operand* w0, const void* x1, const void* w1, const float* bias,
                          float* y, int64_t M, int64_t N, int64_t K0, int64_t K1, int64_t ldy, void* stream) {
    if (!x0 || !w0 || M < 0 || N < 0 || K0 < 0 || K1 < 0 || ldy < N) return MI355Q_E_BADARG;
    if (M == 0 || N == 0) return 0;
    if (!y || !x1 || !w1 || !x0->mant || !x0->exp || !w0->mant || !w0->exp || !x0->rowflag || !w0->rowflag || !x0->gscale || !w0->gscale ||
        !x0->list || !w0->list)
        return MI355Q_E_BADARG;
    if (x0->mbits < 1 || x0->mbits > 7 || w0->mbits < 1 || w0->mbits > 7) return MI355Q_E_BADARG;
    if (x0->row_aligned != 1 || w0->row_aligned != 1 || bucket_cap_of(x0->list_cap) != ROW_BCAP || bucket_cap_of(w0->list_cap) != ROW_BCAP ||
        K0 % 128 != 0 || K0 < 256 || K0 > MI355Q_ROW_ALIGN_MAX_K || K1 % 64 != 0 || K1 < 128 || K1 > MI355Q_ROW_ALIGN_MAX_K)
        return MI355Q_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x0->mant) | reinterpret_cast<uintptr_t>(w0->mant) | reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(w1)) % 16)
        return MI355Q_E_ALIGN;
    GemmArgs a{x0->mant, x0->exp, w0->mant, w0->exp, bias, y, M, N, K0, ldy,
               x0->exp_bias + x0->mbits + w0->exp_bias + w0->mbits, 1,
               x0->exp_bias + x0->mbits, w0->exp_bias + w0->mbits,
               ROW_BCAP, ROW_BCAP, 0};
    a.x_mbits = x0->mbits;
    a.w_mbits = w0->mbits;
    a.xm1 = static_cast<const int8_t*>(x1);
    a.wm1 = static_cast<const int8_t*>(w1);
    a.K1 = K1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t te = g_timing.begin(st);
    const int rc = launch_bfp_gemm_v9_mixed(a, x0->gscale, w0->gscale, x0->list, w0->list, st, x0->rowflag, w0->rowflag);
    g_timing.end(te, st);
    return rc;
}

// x . [gate; up]^T with the gated MLP's elementwise step and the consumer's quantiser in the store epilogue (mi355q.h); epi_op 2:
// x . w^T with relu and the consumer's quantiser there (I = the layer's out_features)
static int gemm_aligned_epilogue_impl(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w, const float* bias, float* scratch,
                                      void* out_bf16_tiled, int64_t M, int64_t I, int64_t K, int32_t q_width, int32_t q_exponent_width,
                                      int32_t q_exponent_bias, void* stream, int epi_op) {
    if (!x || !w || M < 0 || I < 0 || K < 0) return MI355Q_E_BADARG;
    if (M == 0 || I == 0) return 0;
    if (!scratch || !out_bf16_tiled || !x->mant || !x->exp || !w->mant || !w->exp || !x->rowflag || !w->rowflag || !x->gscale || !w->gscale ||
        !x->list || !w->list)
        return MI355Q_E_BADARG;
    if (x->mbits < 1 || x->mbits > 7 || w->mbits < 1 || w->mbits > 7) return MI355Q_E_BADARG;
    if (q_exponent_width < 1 || q_exponent_width > 8 || q_width < 2 || q_width > 9) return MI355Q_E_UNSUPPORTED;     // (exact in bf16: <= 8 mantissa bits)
    if (q_exponent_bias == MI355Q_BIAS_DEFAULT) q_exponent_bias = (1 << (q_exponent_width - 1)) - 1;
    if (x->row_aligned != 1 || w->row_aligned != 1 || bucket_cap_of(x->list_cap) != ROW_BCAP || bucket_cap_of(w->list_cap) != ROW_BCAP ||
        K % 128 != 0 || K < 256 || K > MI355Q_ROW_ALIGN_MAX_K || I % 32 != 0)
        return MI355Q_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x->mant) | reinterpret_cast<uintptr_t>(w->mant) | reinterpret_cast<uintptr_t>(out_bf16_tiled) |
         reinterpret_cast<uintptr_t>(scratch)) % 16)
        return MI355Q_E_ALIGN;
    const int64_t N = epi_op == 2 ? I : 2 * I;
    GemmArgs a{x->mant, x->exp, w->mant, w->exp, bias, scratch, M, N, K, N,
               x->exp_bias + x->mbits + w->exp_bias + w->mbits, 1,
               x->exp_bias + x->mbits, w->exp_bias + w->mbits,
               ROW_BCAP, ROW_BCAP, 0};
    a.x_mbits = x->mbits;
    a.w_mbits = w->mbits;
    a.yb = out_bf16_tiled;
    a.q_mbits = q_width - 1;
    a.q_emin = -q_exponent_bias;
    a.q_emax = (1 << q_exponent_width) - 1 - q_exponent_bias;
    a.epi_op = epi_op;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t te = g_timing.begin(st);
    const int rc = launch_bfp_gemm_v9_gated(a, x->gscale, w->gscale, x->list, w->list, st, x->rowflag, w->rowflag);
    g_timing.end(te, st);
    return rc;
}

int mi355q_bfp_gemm_aligned_gated(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w, const float* bias, float* scratch,
                                  void* out_bf16_tiled, int64_t M, int64_t I, int64_t K, int32_t q_width, int32_t q_exponent_width,
                                  int32_t q_exponent_bias, void* stream) {
    return gemm_aligned_epilogue_impl(x, w, bias, scratch, out_bf16_tiled, M, I, K, q_width, q_exponent_width, q_exponent_bias, stream, 1);
}

int mi355q_bfp_gemm_aligned_relu(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w, const float* bias, float* scratch,
                                 void* out_bf16_tiled, int64_t M, int64_t N, int64_t K, int32_t q_width, int32_t q_exponent_width,
                                 int32_t q_exponent_bias, void* stream) {
    return gemm_aligned_epilogue_impl(x, w, bias, scratch, out_bf16_tiled, M, N, K, q_width, q_exponent_width, q_exponent_bias, stream, 2);
}

static int gemm_aligned_multi_impl(const mi355q_bfp_operand* x, const mi355q_bfp_operand* const* w, const float* const* bias,
                                  float* const* y, int32_t count, int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream) {
    if (!x || !w || !y || count < 1 || count > 3 || M < 0 || N < 0 || K < 0 || ldy < N) return MI355Q_E_BADARG;
    if (M == 0 || N == 0) return 0;
    for (int i = 0; i < count; ++i) {
        if (!w[i] || !y[i] || !w[i]->mant || !w[i]->exp || !w[i]->rowflag || !w[i]->gscale || !w[i]->list) return MI355Q_E_BADARG;
        // one grid over all of them: the same flavour, scales and bucket size everywhere
        if (w[i]->row_aligned != 1 || w[i]->mbits != w[0]->mbits || w[i]->exp_bias != w[0]->exp_bias ||
            bucket_cap_of(w[i]->list_cap) != ROW_BCAP)
            return MI355Q_E_UNSUPPORTED;
        if (reinterpret_cast<uintptr_t>(w[i]->mant) % 16) return MI355Q_E_ALIGN;
    }
    if (!x->mant || !x->exp || !x->rowflag || !x->gscale || !x->list) return MI355Q_E_BADARG;
    if (x->row_aligned != 1 || bucket_cap_of(x->list_cap) != ROW_BCAP || K % 128 != 0 || K > MI355Q_ROW_ALIGN_MAX_K ||
        g_gemm_variant.load() != 0)
        return MI355Q_E_UNSUPPORTED;                  // (callers then launch mi355q_bfp_gemm_aligned per weight)
    if (x->mbits < 1 || x->mbits > 7 || w[0]->mbits < 1 || w[0]->mbits > 7) return MI355Q_E_BADARG;
    if (reinterpret_cast<uintptr_t>(x->mant) % 16) return MI355Q_E_ALIGN;
    if (count == 1) return gemm_aligned_impl(x, w[0], bias ? bias[0] : nullptr, y[0], M, N, K, ldy, stream);
    GemmArgs a{x->mant, x->exp, w[0]->mant, w[0]->exp, bias ? bias[0] : nullptr, y[0], M, N, K, ldy,
               x->exp_bias + x->mbits + w[0]->exp_bias + w[0]->mbits, 1,
               x->exp_bias + x->mbits, w[0]->exp_bias + w[0]->mbits,
               ROW_BCAP, ROW_BCAP, 0};
    a.ngroup = count;
    for (int i = 0; i < count; ++i) {
        a.g_wm[i] = w[i]->mant; a.g_we[i] = w[i]->exp; a.g_sw[i] = w[i]->gscale; a.g_wlist[i] = w[i]->list;
        a.g_wf[i] = w[i]->rowflag; a.g_bias[i] = bias ? bias[i] : nullptr; a.g_y[i] = y[i];
    }
    a.x_mbits = x->mbits;
    a.w_mbits = w[0]->mbits;
    return launch_bfp_gemm_v8(a, x->gscale, w[0]->gscale, x->list, w[0]->list, 0, static_cast<hipStream_t>(stream), x->rowflag,
                              w[0]->rowflag);
}

int mi355q_bfp_gemm_aligned_multi(const mi355q_bfp_operand* x, const mi355q_bfp_operand* const* w, const float* const* bias,
                                  float* const* y, int32_t count, int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream) {
    return gemm_aligned_multi_impl(x, w, bias, y, count, M, N, K, ldy, stream);
}

// ---- W4A4 / W5A5 on the MX scaled matrix instruction (mi355q_mx.hip)
size_t mi355q_mx_plane_bytes(int64_t rows, int64_t K, int32_t plane) {
    if (rows <= 0 || K <= 0 || plane < 0 || plane > 2) return 0;
    const size_t rp = (size_t)((rows + 255) / 256 * 256), kp = (size_t)((K + 127) / 128 * 128);
    return plane == 0 ? rp * kp / 2 : (plane == 1 ? rp * kp / 4 : rp * kp / 32);
}

static int mx_quant_args(QuantArgs& a, const float* x, int64_t rows, int64_t K, int32_t width, int32_t exponent_width, int32_t exponent_bias) {
    if (rows < 0 || K < 0) return MI355Q_E_BADARG;
    if (K % 128 != 0 || K > MI355Q_ROW_ALIGN_MAX_K) return MI355Q_E_UNSUPPORTED;
    if (exponent_width < 1 || exponent_width > 8 || width < 2 || width > 5) return MI355Q_E_UNSUPPORTED;   // <= 4 mantissa bits: exact in e2m3
    if (exponent_bias == MI355Q_BIAS_DEFAULT) exponent_bias = (1 << (exponent_width - 1)) - 1;
    if (exponent_bias < 0) return MI355Q_E_UNSUPPORTED;
    a = QuantArgs{};
    a.x = x;
    a.lead = 1; a.rows = rows; a.cols = K;
    a.b0 = 1; a.b1 = 16;
    a.n_elems = rows * K;
    a.nbr = rows; a.nbc = K / 16;
    a.n_blocks = rows * (K / 16);
    a.flags = MI355Q_ZERO_BLOCK_FAST;
    a.code_bias = exponent_bias;
    a.e_min = -exponent_bias;
    a.e_max = (1 << exponent_width) - 1 - exponent_bias;
    set_mantissa(a, width - 1);
    return 0;
}

int mi355q_block_fp_quantize_mx(const float* x, uint8_t* codes16, uint8_t* codes8, uint8_t* scales, int32_t* bad, int32_t* bad_to_clear,
                                int64_t rows, int64_t K, int32_t width, int32_t exponent_width, int32_t exponent_bias, void* stream) {
    QuantArgs a;
    if (const int rc = mx_quant_args(a, x, rows, K, width, exponent_width, exponent_bias)) return rc;
    if (rows == 0 || K == 0) return 0;
    if (!x || !codes16 || !codes8 || !scales || !bad) return MI355Q_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(codes16) | reinterpret_cast<uintptr_t>(codes8) | reinterpret_cast<uintptr_t>(scales)) % 16)
        return MI355Q_E_ALIGN;
    if (bad_to_clear == bad) return MI355Q_E_BADARG;
    // no "next call's word" to clear (a caller recording a HIP graph: the flag words cannot alternate between replays): this
    // call clears its OWN word in front of the kernel -- a memset node in the graph -- so that a replay whose activations fit
    // the format is never sent to the exact route by an earlier replay's flag
    if (!bad_to_clear) {
        if (hipMemsetAsync(bad, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)) != hipSuccess) return (int)hipGetLastError();
    }
    return launch_quant_mx_rows(a, codes16, codes8, scales, bad, bad_to_clear, static_cast<hipStream_t>(stream));
}

int mi355q_mx_gemm(const uint8_t* x16, const uint8_t* x8, const uint8_t* xs, const uint8_t* w16, const uint8_t* w8, const uint8_t* ws,
                   const int32_t* bad2, const float* x_fp32, const float* w_fp32, const float* bias, float* y, int64_t M, int64_t N,
                   int64_t K, int64_t ldy, int32_t x_width, int32_t x_exponent_width, int32_t x_exponent_bias, void* stream) {
    if (M < 0 || N < 0 || ldy < N) return MI355Q_E_BADARG;
    MxGemmArgs a{};
    if (const int rc = mx_quant_args(a.qx, x_fp32, M, K, x_width, x_exponent_width, x_exponent_bias)) return rc;
    if (M == 0 || N == 0) return 0;
    if (K == 0 || !x16 || !x8 || !xs || !w16 || !w8 || !ws || !bad2 || !x_fp32 || !w_fp32 || !y) return MI355Q_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(x16) | reinterpret_cast<uintptr_t>(x8) | reinterpret_cast<uintptr_t>(xs) | reinterpret_cast<uintptr_t>(w16) |
         reinterpret_cast<uintptr_t>(w8) | reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(x_fp32) | reinterpret_cast<uintptr_t>(w_fp32)) % 16)
        return MI355Q_E_ALIGN;
    a.x16 = x16; a.x8 = x8; a.xs = xs; a.w16 = w16; a.w8 = w8; a.ws = ws;
    a.bad = bad2;
    a.xf = x_fp32; a.wf = w_fp32;
    a.bias = bias; a.y = y;
    a.M = M; a.N = N; a.K = K; a.ldy = ldy;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t te = g_timing.begin(st);
    const int rc = launch_mx_gemm(a, st);
    g_timing.end(te, st);
    return rc;
}

size_t mi355q_bfp_matmul_workspace_bytes(int64_t B, int64_t K, int64_t N) {
    if (B <= 0 || K <= 0 || N <= 0) return 0;
    return (size_t)B * (size_t)((K + 63) / 64 * 64) * (size_t)N * 2 + 64;
}

static int bfp_matmul_impl(bool softmax, const float* mask, long long causal_off, const float* x, const float* y, float* out, void* workspace, int64_t B, int64_t M, int64_t K,
                      int64_t N, int32_t x_width, int32_t x_exponent_width, int32_t x_exponent_bias, int32_t y_width,
                      int32_t y_exponent_width, int32_t y_exponent_bias, void* stream) {
    if (B < 0 || M < 0 || K < 0 || N < 0) return MI355Q_E_BADARG;
    if (B == 0 || M == 0 || N == 0) return 0;
    if (!out || B > 65535) return MI355Q_E_BADARG;
    if (K == 0) return (int)hipMemsetAsync(out, 0, (size_t)B * M * N * 4, static_cast<hipStream_t>(stream));
    if (!x || !y || !workspace) return MI355Q_E_BADARG;
    if (x_exponent_width < 1 || x_exponent_width > 8 || y_exponent_width < 1 || y_exponent_width > 8) return MI355Q_E_BADARG;
    if (x_width < 2 || y_width < 2) return MI355Q_E_BADARG;
    // blocks of 16 along K (x) and N (y) must tile the operands; a quantised value must fit bf16's 8 significant bits
    if (K % 16 != 0 || N % 16 != 0 || x_width > 9 || y_width > 9) return MI355Q_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(workspace)) % 16)
        return MI355Q_E_ALIGN;
    QuantArgs ax{}, ay{};
    auto fill = [](QuantArgs& a, int width, int ew, int bias) {
        if (bias == MI355Q_BIAS_DEFAULT) bias = (1 << (ew - 1)) - 1;
        a.b0 = 1; a.b1 = 16;
        a.code_bias = bias;
        a.e_min = -bias;
        a.e_max = (1 << ew) - 1 - bias;
        set_mantissa(a, width - 1);
    };
    fill(ax, x_width, x_exponent_width, x_exponent_bias);
    fill(ay, y_width, y_exponent_width, y_exponent_bias);
    return launch_bfp_qmatmul(ax, ay, x, y, out, workspace, B, M, K, N, static_cast<hipStream_t>(stream), softmax, mask, causal_off);
}

unsigned long long mi355q_stream_capture_id(void* stream) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(static_cast<hipStream_t>(stream), &status, &id) != hipSuccess) return 0;
    return status == hipStreamCaptureStatusActive ? (id ? id : ~0ull) : 0;
}

int mi355q_rope_apply(const float* q, const float* k, const float* cos_q, const float* sin_q, const int64_t* position_ids,
                      float* q_out, float* k_out, int64_t B, int64_t Hq, int64_t Hk, int64_t T, int64_t D, int64_t table_rows,
                      const int64_t* q_strides, const int64_t* k_strides, void* stream) {
    if (B < 0 || Hq < 0 || Hk < 0 || T < 0 || D < 0) return MI355Q_E_BADARG;
    if (B == 0 || T == 0 || D == 0 || (Hq == 0 && Hk == 0)) return 0;
    if (!q || !k || !cos_q || !sin_q || !position_ids || !q_out || !k_out || !q_strides || !k_strides || table_rows <= 0)
        return MI355Q_E_BADARG;
    if (D % 8 != 0) return MI355Q_E_UNSUPPORTED;              // float4 halves
    for (int i = 0; i < 3; ++i)
        if (q_strides[i] % 4 || k_strides[i] % 4) return MI355Q_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(cos_q) |
         reinterpret_cast<uintptr_t>(sin_q) | reinterpret_cast<uintptr_t>(q_out) | reinterpret_cast<uintptr_t>(k_out)) % 16)
        return MI355Q_E_ALIGN;
    RopeArgs a{};
    a.x[0] = q; a.x[1] = k; a.y[0] = q_out; a.y[1] = k_out;
    a.sb[0] = q_strides[0]; a.sh[0] = q_strides[1]; a.st[0] = q_strides[2];
    a.sb[1] = k_strides[0]; a.sh[1] = k_strides[1]; a.st[1] = k_strides[2];
    a.heads[0] = Hq; a.heads[1] = Hk;
    a.cos = cos_q; a.sin = sin_q;
    a.pos = reinterpret_cast<const long long*>(position_ids);
    a.B = B; a.T = T; a.D = D; a.table_rows = table_rows;
    return launch_rope(a, static_cast<hipStream_t>(stream));
}

int mi355q_bfp_attention_set_kernel(int which) { return attention_set_kernel(which); }
int mi355q_bfp_attention_set_qpack(int on) { return attention_set_qpack(on); }

size_t mi355q_bfp_attention_workspace_bytes(int64_t B, int64_t T, int64_t D) {
    if (B <= 0 || T <= 0 || D <= 0) return 0;
    return attention_workspace_bytes(B, T, D);
}

int mi355q_bfp_attention(const float* q, const float* k, const float* v, const float* mask, int32_t causal, float scale_div,
                         float* out, void* workspace, int64_t B, int64_t M, int64_t T, int64_t D, const int32_t* qk_params,
                         const int32_t* pv_params, void* stream) {
    return mi355q_bfp_attention_strided(q, k, v, mask, causal, scale_div, out, workspace, B, M, T, D, qk_params, pv_params,
                                        nullptr, stream);
}

int mi355q_bfp_attention_strided(const float* q, const float* k, const float* v, const float* mask, int32_t causal, float scale_div,
                                 float* out, void* workspace, int64_t B, int64_t M, int64_t T, int64_t D,
                                 const int32_t* qk_params, const int32_t* pv_params, const int64_t* strides, void* stream) {
    return mi355q_bfp_attention_rope(q, k, v, mask, causal, scale_div, out, workspace, B, M, T, D, qk_params, pv_params, strides, nullptr,
                                     nullptr, nullptr, 0, 1, stream);
}

int mi355q_bfp_attention_rope(const float* q, const float* k, const float* v, const float* mask, int32_t causal, float scale_div,
                              float* out, void* workspace, int64_t B, int64_t M, int64_t T, int64_t D, const int32_t* qk_params,
                              const int32_t* pv_params, const int64_t* strides, const float* cos, const float* sin,
                              const int64_t* position_ids, int64_t table_rows, int32_t heads, void* stream) {
    return mi355q_bfp_attention_fused(q, k, v, mask, causal, 0.f, scale_div, out, nullptr, nullptr, workspace, B, M, T, D, qk_params, pv_params,
                                      strides, cos, sin, position_ids, table_rows, heads, stream);
}

int mi355q_bfp_attention_fused(const float* q, const float* k, const float* v, const float* mask, int32_t causal, float q_scale,
                               float scale_div, float* out, void* out_bf16_tiled, const int32_t* consumer_params, void* workspace, int64_t B, int64_t M,
                               int64_t T, int64_t D, const int32_t* qk_params, const int32_t* pv_params, const int64_t* strides,
                               const float* cos, const float* sin, const int64_t* position_ids, int64_t table_rows, int32_t heads,
                               void* stream) {
    QuantArgs ao{};
    if (out_bf16_tiled) {
        // the consumer's data_in quantiser {width, exponent width, exponent bias}; its operand is [M, B x D] in head order: ONE batch element
        if (!consumer_params || consumer_params[0] < 2 || consumer_params[1] < 1 || consumer_params[1] > 8 || (cos && heads != B)) return MI355Q_E_BADARG;
        if (consumer_params[0] > 9) return MI355Q_E_UNSUPPORTED;
        if (reinterpret_cast<uintptr_t>(out_bf16_tiled) % 16) return MI355Q_E_ALIGN;
        int bias = consumer_params[2];
        if (bias == MI355Q_BIAS_DEFAULT) bias = (1 << (consumer_params[1] - 1)) - 1;
        ao.b0 = 1; ao.b1 = 16;
        ao.code_bias = bias;
        ao.e_min = -bias;
        ao.e_max = (1 << consumer_params[1]) - 1 - bias;
        set_mantissa(ao, consumer_params[0] - 1);
        if (!out) out = reinterpret_cast<float*>(out_bf16_tiled);       // (never written; keeps the argument checks below uniform)
    }
    if (cos || sin || position_ids) {
        if (!cos || !sin || !position_ids || table_rows < 1 || heads < 1) return MI355Q_E_BADARG;
        if ((reinterpret_cast<uintptr_t>(cos) | reinterpret_cast<uintptr_t>(sin)) % 16 || reinterpret_cast<uintptr_t>(position_ids) % 8) return MI355Q_E_ALIGN;
    }
    if (B < 0 || M < 0 || T < 0 || D < 0) return MI355Q_E_BADARG;
    if (B == 0 || M == 0 || D == 0) return 0;
    if (!q || !k || !v || !out || !workspace || !qk_params || !pv_params || T == 0 || B > 65535) return MI355Q_E_BADARG;
    if (causal && T < M) return MI355Q_E_BADARG;              // (query i sees keys 0 .. i + T - M)
    for (int i = 0; i < 2; ++i) {
        const int32_t* pr = i ? pv_params : qk_params;
        if (pr[0] < 2 || pr[3] < 2 || pr[1] < 1 || pr[1] > 8 || pr[4] < 1 || pr[4] > 8) return MI355Q_E_BADARG;
        if (pr[0] > 9 || pr[3] > 9) return MI355Q_E_UNSUPPORTED;      // a quantised value must fit bf16's 8 significant bits
    }
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
         reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(mask)) % 16)
        return MI355Q_E_ALIGN;
    QuantArgs a[4] = {};
    for (int i = 0; i < 4; ++i) {
        const int32_t* pr = (i < 2 ? qk_params : pv_params) + 3 * (i & 1);
        int bias = pr[2];
        if (bias == MI355Q_BIAS_DEFAULT) bias = (1 << (pr[1] - 1)) - 1;
        a[i].b0 = 1; a[i].b1 = 16;
        a[i].code_bias = bias;
        a[i].e_min = -bias;
        a[i].e_max = (1 << pr[1]) - 1 - bias;
        set_mantissa(a[i], pr[0] - 1);
    }
    long long st6[8];
    if (strides)
        for (int i = 0; i < 8; ++i) {
            if (strides[i] % 4) return MI355Q_E_ALIGN;            // (16-byte loads / stores of every row)
            st6[i] = strides[i];
        }
    return launch_bfp_attention(a[0], a[1], a[2], a[3], q, k, v, mask, out, workspace, B, M, T, D, causal ? T - M : -1,
                                scale_div, static_cast<hipStream_t>(stream), strides ? st6 : nullptr, cos, sin,
                                reinterpret_cast<const long long*>(position_ids), table_rows, heads, static_cast<uint16_t*>(out_bf16_tiled),
                                out_bf16_tiled ? &ao : nullptr, q_scale);
}

// block_minifloat (fmt 1) / block_log (fmt 2) products: the same two kernels with the other quantisers' block parameters
static int values_matmul_impl(int fmt, bool softmax, const float* mask, long long causal_off, const float* x, const float* y,
                              float* out, void* workspace, int64_t B, int64_t M, int64_t K, int64_t N, int32_t x_width,
                              int32_t x_exponent_width, int32_t x_exponent_bias_width, int32_t y_width, int32_t y_exponent_width,
                              int32_t y_exponent_bias_width, void* stream) {
    if (B < 0 || M < 0 || K < 0 || N < 0) return MI355Q_E_BADARG;
    if (B == 0 || M == 0 || N == 0) return 0;
    if (!out || B > 65535) return MI355Q_E_BADARG;
    if (K == 0) return (int)hipMemsetAsync(out, 0, (size_t)B * M * N * 4, static_cast<hipStream_t>(stream));
    if (!x || !y || !workspace) return MI355Q_E_BADARG;
    QuantArgs ax{}, ay{};
    ax.b0 = ay.b0 = 1; ax.b1 = ay.b1 = 16;
    if (fmt == 1) {
        auto fill = [](QuantArgs& a, int width, int ew, int ebw) -> int {
            const int mbits = width - ew - 1;
            if (ew < 1 || ew > 8 || mbits < 0 || mbits > 23 || ebw < 1 || ebw > 8) return MI355Q_E_BADARG;
            if (mbits > 7) return MI355Q_E_UNSUPPORTED;          // a quantised value must fit bf16's 8 significant bits
            a.span = (1 << ew) - 1;
            a.bias_max = (1 << ebw) - 1;
            set_mantissa(a, mbits);
            return 0;
        };
        int rc = fill(ax, x_width, x_exponent_width, x_exponent_bias_width);
        if (rc == 0) rc = fill(ay, y_width, y_exponent_width, y_exponent_bias_width);
        if (rc) return rc;
    } else {
        if (x_width < 2 || x_width > 9 || x_exponent_bias_width < 1 || x_exponent_bias_width > 8) return MI355Q_E_BADARG;
        ax.span = (1 << (x_width - 1)) - 1;
        ax.bias_max = (1 << x_exponent_bias_width) - 1;
        set_mantissa(ax, 0);
        set_mantissa(ay, 0);
    }
    if (K % 16 != 0 || N % 16 != 0) return MI355Q_E_UNSUPPORTED;      // blocks of 16 along K (x) and N (y) tile the operands
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(workspace)) % 16)
        return MI355Q_E_ALIGN;
    return launch_bfp_qmatmul(ax, ay, x, y, out, workspace, B, M, K, N, static_cast<hipStream_t>(stream), softmax, mask, causal_off, fmt);
}

int mi355q_block_minifloat_matmul(const float* x, const float* y, float* out, void* workspace, int64_t B, int64_t M, int64_t K,
                                  int64_t N, int32_t x_width, int32_t x_exponent_width, int32_t x_exponent_bias_width,
                                  int32_t y_width, int32_t y_exponent_width, int32_t y_exponent_bias_width, void* stream) {
    return values_matmul_impl(1, false, nullptr, -1, x, y, out, workspace, B, M, K, N, x_width, x_exponent_width,
                              x_exponent_bias_width, y_width, y_exponent_width, y_exponent_bias_width, stream);
}

int mi355q_block_minifloat_softmax_matmul(const float* scores, const float* mask, int32_t causal, const float* y, float* out,
                                          void* workspace, int64_t B, int64_t M, int64_t K, int64_t N, int32_t x_width,
                                          int32_t x_exponent_width, int32_t x_exponent_bias_width, int32_t y_width,
                                          int32_t y_exponent_width, int32_t y_exponent_bias_width, void* stream) {
    if (K > 0 && (K <= 192 || N > 128)) return MI355Q_E_UNSUPPORTED;
    if (causal && K < M) return MI355Q_E_BADARG;              // (query i sees keys 0 .. i + K - M)
    if (mask && reinterpret_cast<uintptr_t>(mask) % 16) return MI355Q_E_ALIGN;
    return values_matmul_impl(1, true, mask, causal ? (long long)(K - M) : -1, scores, y, out, workspace, B, M, K, N, x_width,
                              x_exponent_width, x_exponent_bias_width, y_width, y_exponent_width, y_exponent_bias_width, stream);
}

size_t mi355q_block_log_matmul_workspace_bytes(int64_t B, int64_t K, int64_t N) {
    if (B <= 0 || K <= 0 || N <= 0) return 0;
    return 3 * (size_t)B * (size_t)((K + 63) / 64 * 64) * (size_t)N * 2 + 4096 + 64;      // three bf16 planes of y + the statistics slots (BL_STATS_SLOTS words)
}

int mi355q_block_log_matmul(const float* x, const float* y, float* out, void* workspace, int64_t B, int64_t M, int64_t K,
                            int64_t N, int32_t x_width, int32_t x_exponent_bias_width, void* stream) {
    return values_matmul_impl(2, false, nullptr, -1, x, y, out, workspace, B, M, K, N, x_width, 0, x_exponent_bias_width, 0, 0, 0, stream);
}

int mi355q_bfp_matmul(const float* x, const float* y, float* out, void* workspace, int64_t B, int64_t M, int64_t K,
                      int64_t N, int32_t x_width, int32_t x_exponent_width, int32_t x_exponent_bias, int32_t y_width,
                      int32_t y_exponent_width, int32_t y_exponent_bias, void* stream) {
    return bfp_matmul_impl(false, nullptr, -1, x, y, out, workspace, B, M, K, N, x_width, x_exponent_width, x_exponent_bias, y_width,
                           y_exponent_width, y_exponent_bias, stream);
}

int mi355q_bfp_softmax_matmul(const float* scores, const float* mask, int32_t causal, const float* y, float* out, void* workspace,
                              int64_t B, int64_t M, int64_t K, int64_t N, int32_t x_width, int32_t x_exponent_width,
                              int32_t x_exponent_bias, int32_t y_width, int32_t y_exponent_width, int32_t y_exponent_bias,
                              void* stream) {
    if (K > 0 && (K <= 192 || N > 128)) return MI355Q_E_UNSUPPORTED;
    if (causal && K < M) return MI355Q_E_BADARG;              // (query i sees keys 0 .. i + K - M)
    if (mask && reinterpret_cast<uintptr_t>(mask) % 16) return MI355Q_E_ALIGN;
    return bfp_matmul_impl(true, mask, causal ? (long long)(K - M) : -1, scores, y, out, workspace, B, M, K, N, x_width, x_exponent_width, x_exponent_bias, y_width,
                           y_exponent_width, y_exponent_bias, stream);
}

int mi355q_gemm_timing_enable(int enable) {
    const int prev = g_timing.enabled ? 1 : 0;
    g_timing.enabled = enable != 0;
    return prev;
}

int mi355q_gemm_timing_read(int32_t* count, float* avg_ms, float* min_ms) {
    if (!count || !avg_ms || !min_ms) return MI355Q_E_BADARG;
    double sum = 0.0;
    float mn = 0.f;
    int n = 0;
    for (size_t i = 0; i < g_timing.used; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_timing.pool[i].second);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_timing.pool[i].first, g_timing.pool[i].second);
        if (e != hipSuccess) return (int)e;
        sum += ms;
        mn = (n == 0 || ms < mn) ? ms : mn;
        ++n;
    }
    g_timing.used = 0;
    *count = n;
    *avg_ms = n ? (float)(sum / n) : 0.f;
    *min_ms = mn;
    return 0;
}

int mi355q_bfp_gemm_set_variant(int variant) { return g_gemm_variant.exchange(variant); }

}  // extern "C"
