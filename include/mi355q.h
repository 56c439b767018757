/* mi355q -- MI355X (gfx950) block-quantised linear / matmul hot path, C ABI.
 *
 * The reference (ChengZhang-98/llm-mixed-q) is pure Python/PyTorch and has NO FFI: its
 * boundary for this path is a name-keyed registry of Python callables.  This header is the
 * C-ABI layer the build adds underneath that registry (last row).  Each entry
 * point names the reference callable whose arithmetic it replaces; the Python mirror in
 * llm-mixed-q_amd/mi355q/quantize/ binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it, nothing synchronises the host;
 *   - return value: 0 on success, a positive hipError_t, or a negative MI355Q_E_* code;
 *     mi355q_error_string() explains either;
 *   - tensors are contiguous row-major.  A quantiser input is described as `lead` planes of
 *     `rows x cols` fp32 values tiled by `b0 x b1` blocks (blocks never straddle planes;
 *     ragged edge blocks behave as if zero-padded on the right/bottom).  The reference's
 *     five blocking cases (quantizers/utils.py:86-237, 261-284) map onto it as
 *         bias   [O]      -> lead 1, rows 1, cols O, b0 1,  b1 B
 *         act    [N,C]    -> lead 1, rows N, cols C, b0 1,  b1 B       (skip_first_dim)
 *         weight [O,K]    -> lead 1, rows O, cols K, b0 B0, b1 B1
 *         act    [B,T,C]  -> lead B, rows T, cols C, b0 B0, b1 B1      (skip_first_dim)
 *     with (b0,b1) the reference's right-aligned, clamped block shape (utils.py:42-66).
 *   - `workspace`: MI355Q_WORKSPACE_BYTES bytes of device memory, ZERO-INITIALISED once by
 *     the caller, private to one stream; the library leaves its control words and the zero-state slots zeroed
 *     after every call.  One word survives from call to call on purpose: the fill the last tensor with all-zero
 *     blocks ended up with (fp32 bits), which the next exact-mode call writes its all-zero blocks with before it
 *     knows its own -- a guess that costs nothing when wrong (the fix-up pass rewrites them as before) and a whole
 *     pass over the tensor less when right (attention probabilities under a causal mask, layer after layer).
 *   - Scratch the library owns itself, one per (device, stream), grow-only, allocated with hipMalloc on first need and
 *     therefore NOT while the stream is being captured into a graph (run the call once un-captured first, as
 *     mi355q.graphs.GraphedForward's warm-up does): the split-K slabs and tickets of the tile GEMM (a launch that would
 *     have to grow them under capture runs unsplit instead; buffers a capture has seen are never freed), and the
 *     zero-block map of the exact quantiser mode on tensors of 128 MiB and more (without it that mode reads x a second
 *     time instead).
 *   - Environment switches (diagnostics and A/B runs only; the default route never needs one): MI355Q_V9=0 keeps every
 *     launch on the round-2 tile kernel (mi355q_gemm_v8.hip) -- by default every launch of the 256 x 256 tile, grouped or
 *     not, with or without exception lists, the bf16 flavour too, with K % 128 == 0 and >= 4 K-steps per slice takes
 *     mi355q_gemm_v9.hip; MI355Q_V9_FIX=0 sends the launches with lists back to v8; MI355Q_V10=1|2|3|4 (+ MI355Q_V10_NS) pins a
 *     geometry of the small-tile kernel (mi355q_gemm_v10.hip), MI355Q_V10_AUTO=0 keeps launches off it; MI355Q_V8_TILE_ROWS,
 *     MI355Q_V8_SPLITS pin the tile GEMM's tile height / split-K (tests do); MI355Q_MATMUL_RW=1|2 the row groups
 *     per wave of the short-contraction tile product; MI355Q_MATMUL_TILE=0 sends the plain attention
 *     products back to kernel 2 of mi355q_matmul.hip; MI355Q_V8_STAMPS / MI355Q_V8_CLOCK / MI355Q_V9_STAMPS / MI355Q_V9_DBG /
 *     MI355Q_MATMUL_DBG are the phase-stamp diagnostics behind DESIGN.md section 5.  (Round 5 removed MI355Q_V8_SCHED,
 *     MI355Q_V8_SMALL_SCHED, MI355Q_V8_DBG, MI355Q_V9_GROUPS, MI355Q_QROWS_GRID, MI355Q_QV_PIECES, MI355Q_FIXUP_GRID, MI355Q_CORR.)
 */
#ifndef MI355Q_H
#define MI355Q_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355Q_ABI_VERSION 24
#define MI355Q_WORKSPACE_BYTES 16384

/* negative error codes (positive values are hipError_t) */
#define MI355Q_E_BADARG (-1)      /* null pointer, non-positive size, width out of range */
#define MI355Q_E_UNSUPPORTED (-2) /* legal for the reference, not built here (message says what) */
#define MI355Q_E_ALIGN (-3)       /* pointer / leading dimension alignment requirement violated */

/* exponent_bias argument: "none given" (the reference's exponent_bias=None -> 2^(exponent_width-1)-1) */
#define MI355Q_BIAS_DEFAULT INT32_MIN

/* flags */
#define MI355Q_ZERO_BLOCK_EXACT 0u /* default: all-zero blocks take the reference's global fill
                                      (block_fp.py:54-58): second, usually empty, launch */
#define MI355Q_ZERO_BLOCK_FAST 1u  /* all-zero blocks get mantissa 0 / the minimum exponent and
                                      block_log zeros take the block's own bias: one launch */

int mi355q_abi_version(void);
const char* mi355q_error_string(int code);

/* ---- block floating point ---------------------------------------------------------
 * replaces: quantizers/block_fp.py:21-96  (_block_fp_quantize) behind
 *           QUANTIZER_MAP["block_fp"] (quantizers/__init__.py:8-16).
 * y    (nullable) fp32, same shape as x: the fake-quantised tensor the reference returns,
 *      including the |x| <= 1e-8 pass-through (block_fp.py:93-94).  Every value equals the reference's; the SIGN OF A
 *      ZERO result is not pinned: a negative element whose mantissa rounds to 0 leaves the reference as -0.0 on its 2-D
 *      activation path and as +0.0 on its 3-D activation / weight paths (the unblock step of those sums with +0.0),
 *      and leaves this kernel as -0.0 (sign(x + 1e-9) * 0) on all of them.  No product can tell the two apart.
 * mant (nullable) int8, same shape as x: sign(x+1e-9) * integer mantissa, |mant| <= 2^(width-1)-1.
 * exp  (nullable, together with mant) uint8 [lead, ceil(rows/b0), ceil(cols/b1)]:
 *      shared exponent + exponent_bias (the stored, biased code).
 * width in [2,8] when mant is requested ([2,25] for y only); exponent_width in [1,8];
 * exponent_bias == MI355Q_BIAS_DEFAULT selects the default 2^(exponent_width-1)-1 (the reference's
 * exponent_bias=None, block_fp.py:61-62); any other value, negative ones included, is used literally
 * (e_min = -bias, e_max = 2^exponent_width - 1 - bias).  The packed / aligned operand entry points
 * (mant + biased uint8 exponent codes) take non-negative biases only. */
int mi355q_block_fp_quantize(const float* x, float* y, int8_t* mant, uint8_t* exp,
                             int64_t lead, int64_t rows, int64_t cols, int32_t b0, int32_t b1,
                             int32_t width, int32_t exponent_width, int32_t exponent_bias,
                             uint32_t flags, void* workspace, void* stream);

/* The fake-quantised values as bf16 (exact for width <= 9; elements |x| <= 1e-8, which pass through unquantised,
 * are rounded to bf16): the operand format of a bf16 MFMA GEMM on quantised values.  Row-vector blocks only
 * (b0 == 1, cols % b1 == 0, b1 % 4 == 0), else MI355Q_E_UNSUPPORTED.  y 8-byte aligned. */
int mi355q_block_fp_quantize_bf16(const float* x, uint16_t* y, int64_t lead, int64_t rows, int64_t cols,
                                  int32_t b0, int32_t b1, int32_t width, int32_t exponent_width,
                                  int32_t exponent_bias, void* workspace, void* stream);
/* The same for the other two block formats -- the operands of the reference's matmul_block_minifloat / matmul_block_log
 * (quantized_functions/matmul.py:199-249, 252-297) when the product runs on bf16 MFMAs: block_minifloat values carry at most
 * 7 mantissa bits here (more: MI355Q_E_UNSUPPORTED), block_log values are signed powers of two -- both exact in bf16, and a
 * product of two of them exact in fp32.  block_minifloat: all-zero blocks stay zero (no fix-up pass); block_log: the exact
 * zero-block rule (the tensor-global fill), [1,16] row blocks only (b0 == 1, b1 == 16, cols % 16 == 0, y 16-byte aligned),
 * workspace as for mi355q_block_log_quantize. */
int mi355q_block_minifloat_quantize_bf16(const float* x, uint16_t* y, int64_t lead, int64_t rows, int64_t cols, int32_t b0,
                                         int32_t b1, int32_t width, int32_t exponent_width, int32_t exponent_bias_width,
                                         void* workspace, void* stream);
int mi355q_block_log_quantize_bf16(const float* x, uint16_t* y, int64_t lead, int64_t rows, int64_t cols, int32_t b0,
                                   int32_t b1, int32_t width, int32_t exponent_bias_width, void* workspace, void* stream);

/* ---- operands whose blocks keep their own exponents: bf16 flavour of the tile GEMM ----------------
 * replaces: quantized_modules/linear.py:59-76 for inputs no row window fits (post-SiLU / post-ReLU activations, weights
 * with outlier input channels) -- the reference's x_q @ W_q^T on values that are exact in bf16 (width <= 9), products
 * exact in fp32, fp32 accumulation as in its F.linear.
 * mi355q_block_fp_quantize_bf16_tiled: x fp32 [rows, K], [1,16] blocks along K -> y_tiled bf16 in the GEMM's tile
 *   order (mi355q_bfp_tiled_bytes(rows, 2 K) bytes: 1-KiB pieces of 16 rows x 32 values, [8 values][row][16 bytes]
 *   inside) and, if y != NULL, the fp32 fake-quantised values (y == x allowed: the weights' in-place overwrite,
 *   linear.py:66-70).  K % 32 == 0, width <= 9.  All-zero blocks quantise to zeros (MI355Q_ZERO_BLOCK_FAST).
 * mi355q_bf16_gemm_tiled: y[M, N] = x . w^T (+ bias), fp32, ldy >= N; the same 256 x 256 / 128 x 256 tile kernel as the
 *   row-scale int8 GEMM with v_mfma_f32_16x16x32_bf16. */
int mi355q_block_fp_quantize_bf16_tiled(const float* x, float* y, uint16_t* y_tiled, int64_t rows, int64_t K, int32_t width,
                                        int32_t exponent_width, int32_t exponent_bias, void* workspace, void* stream);

/* block_minifloat values straight into the same tiled bf16 operand (quantized_modules/linear.py:174-203, the x quantiser of
 * PTQ LinearBlockMinifloat in front of its F.linear): x fp32 [rows, K], [1,16] blocks along K, each block's shared exponent
 * bias and elements as mi355q_block_minifloat_quantize forms them (block_minifloat.py:22-74, minifloat.py:134-196); a
 * minifloat of <= 7 mantissa bits is exact in bf16 (more: MI355Q_E_UNSUPPORTED).  K % 32 == 0.  All-zero blocks give zeros
 * (as the reference's do: its fill value only moves their bias). */
int mi355q_block_minifloat_quantize_bf16_tiled(const float* x, uint16_t* y_tiled, int64_t rows, int64_t K, int32_t width,
                                               int32_t exponent_width, int32_t exponent_bias_width, void* workspace,
                                               void* stream);

/* ---- the elementwise step in front of fc2 / down_proj folded into the operand quantisers ---------------
 * replaces: models/opt_quantized/modeling_opt.py:412-420 (`fc2(relu(fc1(x)))`: the activation_fn between the two quantised
 * Linears) and models/llama_quantized/modeling_llama.py:216 (`down_proj(act_fn(gate_proj(x)) * up_proj(x))`) -- there two /
 * three torch kernels whose result the Linear's x quantiser reads back; here the quantiser reads fc1's / gate's and up's
 * outputs itself.  pre_op: MI355Q_PRE_NONE x, MI355Q_PRE_RELU max(x, 0), MI355Q_PRE_SILU_MUL (x / (1 + exp(-x))) * x2 with
 * every operation rounded to fp32 as the separate kernels round it (x2: same shape as x, 16-byte aligned; NULL otherwise).
 * Everything else as mi355q_block_fp_quantize_bf16_tiled / mi355q_block_fp_quantize_aligned_rows (which are these with
 * MI355Q_PRE_NONE). */
#define MI355Q_PRE_NONE 0
#define MI355Q_PRE_RELU 1
#define MI355Q_PRE_SILU_MUL 2
#define MI355Q_PRE_RMSNORM 3        /* mi355q_block_fp_quantize_aligned_rows_norm only */
#define MI355Q_PRE_LAYERNORM 4      /* mi355q_block_fp_quantize_aligned_rows_norm only */
int mi355q_block_fp_quantize_bf16_tiled_pre(const float* x, const float* x2, int32_t pre_op, float* y, uint16_t* y_tiled,
                                            int64_t rows, int64_t K, int32_t width, int32_t exponent_width,
                                            int32_t exponent_bias, void* workspace, void* stream);
/* ... and with LlamaRMSNorm in front (ABI 22): pre_op = MI355Q_PRE_RMSNORM, x2 = the norm's weight [K], eps its epsilon; the row's
 * mean of squares is summed in the row quantiser's fixed order (mi355q_block_fp_quantize_aligned_rows_norm), so a layer sees the same
 * normalised values on either route; the other pre_op values as above.  The layers of a group that share the input (q / k / v,
 * gate / up) on the per-block-exponent route take ONE such operand (mi355q.quantize: grouped_linear). */
int mi355q_block_fp_quantize_bf16_tiled_norm(const float* x, const float* x2, int32_t pre_op, float eps, float* y, uint16_t* y_tiled,
                                             int64_t rows, int64_t K, int32_t width, int32_t exponent_width,
                                             int32_t exponent_bias, void* workspace, void* stream);
int mi355q_block_fp_quantize_aligned_rows_pre(const float* x, const float* x2, int32_t pre_op, int8_t* mant_tiled,
                                              uint8_t* exp_out, uint8_t* rowflag, float* rowscale, int32_t* list,
                                              int32_t* list_to_clear, int64_t rows, int64_t K, int32_t width,
                                              int32_t exponent_width, int32_t exponent_bias, int32_t bucket_cap,
                                              void* stream);
/* The same with LlamaRMSNorm in front (models/llama_quantized/modeling_llama.py:81-92: `weight * (x * rsqrt(mean(x^2, -1) +
 * eps))`, the input of q / k / v and of gate / up, which nothing else reads): pre_op = MI355Q_PRE_RMSNORM, x2 = the norm's
 * weight [K] (16-byte aligned), eps its epsilon; the row kernel already holds the whole row, so the normalised tensor is
 * never written.  The products round to fp32 one by one as the reference's ops do; the mean is summed in a fixed order of
 * this kernel's own (reproducible; within an ulp or two of any other fp32 summation order, torch's reduction kernels
 * included).  pre_op = MI355Q_PRE_LAYERNORM: nn.LayerNorm over the row instead (models/opt_quantized/modeling_opt.py:391-415,
 * self_attn_layer_norm in front of q / k / v, final_layer_norm in front of fc1): (x - mean) * rsqrt(var + eps) * x2 + x3,
 * biased variance, x3 = the norm's bias [K] or NULL.  Other pre_op values behave as in
 * mi355q_block_fp_quantize_aligned_rows_pre (eps, x3 ignored). */
int mi355q_block_fp_quantize_aligned_rows_norm(const float* x, const float* x2, const float* x3, int32_t pre_op, float eps,
                                               int8_t* mant_tiled,
                                               uint8_t* exp_out, uint8_t* rowflag, float* rowscale, int32_t* list,
                                               int32_t* list_to_clear, int64_t rows, int64_t K, int32_t width,
                                               int32_t exponent_width, int32_t exponent_bias, int32_t bucket_cap,
                                               void* stream);
/* The same reading x as P ROW SEGMENTS: element k of row r at x[(k / seg_len) * seg_stride + r * seg_len + k % seg_len] --
 * the rank-major result [P][rows][seg_len] of the all-gather that rebuilds a Linear's output from its out_features shards
 * (BASELINE north_star: "partition the per-layer GEMMs row-wise across the 8 GPUs with RCCL all-gather"; SURVEY 8e;
 * mi355q/sharded.py).  The next layer's quantiser reads the P pieces where the collective left them, so the [rows, K] tensor
 * is never re-assembled (a permute copy of rows x K fp32: 20 us at 4096 x 4096).  seg_len % 4 == 0, K % seg_len == 0,
 * seg_stride % 4 == 0; x2 of MI355Q_PRE_SILU_MUL lies like x; seg_len = 0: plain rows.  Outputs bit-identical to the
 * plain call on the re-assembled tensor. */
int mi355q_block_fp_quantize_aligned_rows_seg(const float* x, const float* x2, const float* x3, int32_t pre_op, float eps,
                                              int8_t* mant_tiled, uint8_t* exp_out, uint8_t* rowflag, float* rowscale,
                                              int32_t* list, int32_t* list_to_clear, int64_t rows, int64_t K, int64_t seg_len,
                                              int64_t seg_stride, int32_t width, int32_t exponent_width, int32_t exponent_bias,
                                              int32_t bucket_cap, void* stream);
/* ---- true width-bit weight storage ---------------------------------------------------
 * replaces: nothing the reference executes -- it realises the storage its profiler accounts for
 * (quantized_layer_profiler.py:18-27: width bits per value + exponent_width bits per block; README.md:11, 5x memory
 * density at 6 bits).
 * mi355q_bfp_pack_bits: mant int8 [rows, K] (canonical: |m| < 2^(width-1), as mi355q_block_fp_quantize writes them)
 *   -> packed [mi355q_bfp_packed_bytes(rows, K, width)]: per row a dense little-endian bit string of width-bit
 *   two's-complement values (a 16-block is 2 * width bytes).  K % 16 == 0.
 * mi355q_bfp_expand: packed + one code byte per block -> a tiled GEMM operand, a pure streaming pass:
 *   mode 0: int8 row-scale operand (mi355q_bfp_tiled_bytes(rows, K)); code = the block's left shift onto its row's
 *           exponent, 0xFF = exception block (zero in the operand, kept in the row's exception list); K % 64 == 0;
 *           with row_exp [rows] / exp_out [rows, K/16] (both or neither) also the operand's per-block exponent bytes
 *           (every block carries its row's);
 *   mode 1: tiled bf16 values m * 2^(code - exp_offset), exp_offset = exponent_bias + width - 1
 *           (mi355q_bfp_tiled_bytes(rows, 2 K)); code = the block's biased exponent; K % 32 == 0.
 * At rest: width + 0.5 bits per value. */
size_t mi355q_bfp_packed_bytes(int64_t rows, int64_t K, int32_t width);
int mi355q_bfp_pack_bits(const int8_t* mant, uint8_t* packed, int64_t rows, int64_t K, int32_t width, void* stream);
int mi355q_bfp_expand(const uint8_t* packed, const uint8_t* codes, void* out_tiled, int64_t rows, int64_t K, int32_t width,
                      int32_t mode, int32_t exp_offset, const uint8_t* row_exp, uint8_t* exp_out, void* stream);

/* values that are already quantised (exact in bf16), fp32 [rows, K] -> the same tiled bf16 (a cast; K % 32 == 0) */
int mi355q_bf16_tile(const float* x, uint16_t* y_tiled, int64_t rows, int64_t K, void* stream);
int mi355q_bf16_gemm_tiled(const uint16_t* x_tiled, const uint16_t* w_tiled, const float* bias, float* y, int64_t M, int64_t N,
                           int64_t K, int64_t ldy, void* stream);
/* An fp32 operand [rows, K] as the tiled bf16 operand of an fp32-EQUIVALENT product (round 6, ABI 24) -- for the layers the reference
 * leaves unquantised: the language-model head, nn.Linear in fp32 (models/llama_quantized/modeling_llama.py:772,866,
 * models/opt_quantized/modeling_opt.py:942-944), 14 % of a Llama-7B forward on the vendor library's fp32 GEMM.  Every value becomes three bf16
 * parts h + m + l (24 significand bits); the operand written is [rows, 6 K]: role 0 (left, activations) = [m | l | h | m | h | h],
 * role 1 (right, weights) = [m | h | l | h | m | h], so that mi355q_bf16_gemm_tiled(x6, w6, bias, y, M, N, 6 K, ...) adds the six part
 * products of weight >= 2^-18 in fp32, smallest first (the three dropped ones weigh <= 2^-26 of a product).  Against an fp64 product the
 * result is CLOSER than the vendor fp32 GEMM's (3.4e-7 vs 1.0e-6 of the mean magnitude at [2048, 4096] x [32000, 4096]^T) and 1.7 x
 * faster (profiles/r06_lm_head_split.json).  y_tiled: mi355q_bfp_tiled_bytes(rows, 12 K) bytes; K % 32 == 0; |x| within bf16's finite
 * range.  mi355q_split.hip. */
int mi355q_fp32_split_tile(const float* x, uint16_t* y_tiled, int64_t rows, int64_t K, int32_t role, void* stream);
/* ... with the residual add of the caller's decoder layer in the store (ABI 22): y = (x . w^T + bias) + residual, the two sums rounded
 * like F.linear followed by `residual + hidden_states` (modeling_llama.py:259, 265; modeling_opt.py:375, 425): the same bits, one
 * pass over [M, N] less.  residual [M, >= N] fp32, ldr elements a row (ldr % 4 == 0), 16-byte aligned; y may be residual itself. */
int mi355q_bf16_gemm_tiled_res(const uint16_t* x_tiled, const uint16_t* w_tiled, const float* bias, const float* residual, int64_t ldr,
                               float* y, int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream);
/* The same with x as `x_segments` COLUMN segments (ABI 20): segment s is the tiled bf16 operand of columns
 * [s K / x_segments, (s + 1) K / x_segments) of x -- [row piece][K-steps of the segment][1 KiB], what
 * mi355q_block_fp_quantize_bf16_tiled writes for that slice -- and the segments lie x_segment_stride_bytes apart: the
 * rank-major buffer an all-gather of per-rank QUANTISED output slices leaves (mi355q.sharded, gather = "quantised": 2 bytes per
 * value travel instead of 4, and every rank quantises only its own slice).  Bit-identical to the plain call on the
 * re-assembled operand.  K % (32 x_segments) == 0. */
int mi355q_bf16_gemm_tiled_seg(const uint16_t* x_tiled, const uint16_t* w_tiled, const float* bias, float* y, int64_t M, int64_t N,
                               int64_t K, int64_t ldy, int32_t x_segments, int64_t x_segment_stride_bytes, void* stream);

/* ---- block minifloat ----------------------------------------------------------------
 * replaces: quantizers/block_minifloat.py:22-74 -> quantizers/minifloat.py:134-196 behind
 *           QUANTIZER_MAP["block_minifloat"].   y: fake-quantised fp32 (required).
 * bias (nullable) uint8 [n_blocks]: the shared exponent bias per block. */
int mi355q_block_minifloat_quantize(const float* x, float* y, uint8_t* bias,
                                    int64_t lead, int64_t rows, int64_t cols, int32_t b0, int32_t b1,
                                    int32_t width, int32_t exponent_width,
                                    int32_t exponent_bias_width,
                                    uint32_t flags, void* workspace, void* stream);

/* ---- block logarithmic ----------------------------------------------------------------
 * replaces: quantizers/block_log.py:23-69 -> quantizers/log.py:22-56 behind
 *           QUANTIZER_MAP["block_log"].   y: fake-quantised fp32 (required).
 * bias (nullable) uint8 [n_blocks]. */
int mi355q_block_log_quantize(const float* x, float* y, uint8_t* bias,
                              int64_t lead, int64_t rows, int64_t cols, int32_t b0, int32_t b1,
                              int32_t width, int32_t exponent_bias_width,
                              uint32_t flags, void* workspace, void* stream);

/* ---- integer fixed point (RoPE tables; quantizers/integer.py:25-58) -------------------- */
int mi355q_integer_quantize(const float* x, float* y, int64_t n, int32_t width,
                            int32_t frac_width, int32_t is_signed, void* stream);

/* ---- block-fp GEMM: the contraction of a PTQ LinearBlockFP forward -----------------------
 * replaces: F.linear(x_q, W_q, b_q) in quantized_modules/linear.py:71 when x and W are
 *           block_fp with [1,16] blocks along in_features:
 *   y[m,n] = sum_kb 2^(xe[m,kb]-x_bias-x_mbits + we[n,kb]-w_bias-w_mbits)
 *                   * sum_{j<16} xm[m,16kb+j] * wm[n,16kb+j]          (+ bias[n])
 * with the inner sum an exact int8 x int8 -> int32 MFMA dot and the outer sum in fp32.
 * xm [M,K] int8, xe [M,K/16] uint8, wm [N,K] int8, we [N,K/16] uint8 as produced by
 * mi355q_block_fp_quantize with b0=1, b1=16; K % 16 == 0; bias (nullable) fp32 [N],
 * already quantised; y fp32 with leading dimension ldy >= N (ldy lets a rank write its
 * column slice of a wider output).  x_mbits/w_mbits = width-1 of each operand. */
int mi355q_bfp_gemm(const int8_t* xm, const uint8_t* xe, const int8_t* wm, const uint8_t* we,
                    const float* bias, float* y, int64_t M, int64_t N, int64_t K, int64_t ldy,
                    int32_t x_mbits, int32_t x_exp_bias, int32_t w_mbits, int32_t w_exp_bias,
                    void* stream);

/* ---- tiled operands ---------------------------------------------------------------------------
 * The tile GEMMs stream an operand's mantissas in TILE ORDER: int8 [mi355q_bfp_tiled_bytes(rows, K)] (K % 64 == 0): 1-KiB
 * pieces of 16 rows x 64 K-bytes, piece (row / 16) * (K / 64) + k / 64, laid out [block (k / 16) % 4][row % 16][16 bytes]
 * inside (the LDS image of the kernels; the 16 rows' blocks at one K position are 256 contiguous bytes); rows padded to 128.
 * Exception lists (int32): list[0] = overflow word, list[1..7] spare, then entries of 8 words {row (-1 = void), block
 * index (k / 16), biased exponent, 0, 16 mantissa bytes}.
 * (Round 5 removed the alignment per 256-value GROUP -- mi355q_bfp_align, mi355q_block_fp_quantize_aligned, the int32-chain
 * kernel behind them -- which the row alignment below superseded in round 1.) */
size_t mi355q_bfp_tiled_bytes(int64_t rows, int64_t K);
size_t mi355q_bfp_rowflag_bytes(int64_t rows, int64_t K);
int64_t mi355q_bfp_rows_pad(int64_t rows);

/* ---- ROW-aligned operands -------------------------------------------------------------------
 * A packed operand rewritten so that a whole row (all K/16 blocks) carries ONE effective exponent E and one fp32 scale:
 * blocks are shifted left onto E where that keeps their mantissas inside int8; the contraction becomes a plain int8 x int8 -> int32 GEMM with a row scale and a column
 * scale applied once (no rescale inside the K loop).  Blocks outside the row's exponent window are exceptions,
 * kept in a list with one BUCKET per 256 rows:
 *   rowflag  uint8 [rows];  rowscale fp32 [mi355q_bfp_rows_pad(rows)] = 2^(E - exp_offset), 0 where rowflag is 0
 *   list     int32 [mi355q_bfp_row_list_bytes(rows, cap) / 4]: list[0] = rows whose exceptions did not fit their
 *            bucket (such rows are copied unchanged, rowflag 0; the GEMM then takes its blockwise kernel),
 *            list[1..7] spare; bucket b (rows 256 b ...) at word 8 + b * (8 + 8 * cap): [0] entries reserved,
 *            [1..7] spare, then cap entries {row (-1 = void), block, biased exponent, 0, 16 mantissa bytes}.
 *            `bucket_cap` = cap, 0 = the default 120 (what the GEMM can hold in LDS: required of the WEIGHT operand);
 *            an ACTIVATION operand may use up to MI355Q_ROW_BUCKET_CAP_MAX -- its exception blocks are then added
 *            by a row post-pass after the GEMM (see mi355q_bfp_gemm_aligned), which has no per-tile limit.
 * K % 64 == 0, K <= MI355Q_ROW_ALIGN_MAX_K (int32 accumulation cannot overflow; the row is decided by one
 * workgroup that keeps it in registers).  mi355q_block_fp_quantize_aligned_rows is the fused activation form
 * (one pass over x [rows, K] fp32: quantise + pack + align + tile; `list` must hold zero counts on entry: alternate between
 * two lists and pass the OTHER one as `list_to_clear` -- this kernel empties it for the next call). */
#define MI355Q_ROW_ALIGN_MAX_K 16384
#define MI355Q_ROW_BUCKET_CAP_MAX 1016
#define MI355Q_ROW_NO_ALIGN (-1)   /* bucket_cap of mi355q_block_fp_quantize_aligned_rows: tiled row format, every block keeps
                                    * its own exponent (rowflag 0, rowscale 0, list untouched / may be NULL) */
size_t mi355q_bfp_row_list_bytes(int64_t rows, int32_t bucket_cap);
int mi355q_bfp_align_rows(const int8_t* mant_in, const uint8_t* exp_in, int8_t* mant_tiled, uint8_t* exp_out,
                          uint8_t* rowflag, float* rowscale, int32_t* list, int32_t exp_offset, int64_t rows,
                          int64_t K, int32_t bucket_cap, void* stream);
int mi355q_block_fp_quantize_aligned_rows(const float* x, int8_t* mant_tiled, uint8_t* exp_out, uint8_t* rowflag,
                                          float* rowscale, int32_t* list, int32_t* list_to_clear, int64_t rows,
                                          int64_t K, int32_t width, int32_t exponent_width, int32_t exponent_bias,
                                          int32_t bucket_cap, void* stream);

/* An aligned operand as one argument */
typedef struct mi355q_bfp_operand {
    const int8_t* mant;     /* tiled mantissas */
    const uint8_t* exp;     /* [rows, K/16] */
    const uint8_t* rowflag; /* [rows] */
    const float* gscale;    /* row scales [rows_pad] */
    const int32_t* list;    /* exception list, mi355q_bfp_row_list_bytes(rows, list_cap) */
    int32_t list_cap;       /* entries per bucket (0 = 120) */
    int32_t mbits;          /* width - 1 */
    int32_t exp_bias;
    int32_t row_aligned;    /* 2: row format, nothing aligned (x only: MI355Q_ROW_NO_ALIGN;
                             * the GEMM takes its blockwise-exact kernel: inputs no row window fits); 1: whole rows (mi355q_bfp_align_rows):
                             * rowflag [rows], gscale = rowscale [rows_pad], list = bucketed row list */
} mi355q_bfp_operand;

/* Same contraction as mi355q_bfp_gemm on ROW-aligned tiled operands (K % 64 == 0).
 * ROW-aligned operands (row_aligned = 1 in both): the row-scale int8 GEMM (256 x 256 or 128 x 256 tiles, whole K in
 * int32, one fp32 scale per row and per column).  Each tile multiplies the exception blocks of its own rows / columns
 * with the other operand itself (one fp32 vector of products per exception, kept in LDS) and adds them in its store
 * epilogue.  No atomics: results are reproducible.  K % 128 == 0 for the fast kernel.  A second launch leaves at once
 * unless an exception bucket overflowed; then the GEMM has written nothing and this launch forms the whole product
 * blockwise-exact (decided on the device).
 * If x uses buckets larger than 120 entries, x's exception blocks are instead added by a ROW POST-PASS after the GEMM
 * (one workgroup per (bucket, 64 columns) owns the y rows it updates: plain read-add-write, entries taken in
 * (row, block) order -- reproducible too, and without a per-tile limit; meant for post-activation inputs).  The
 * weight operand always uses the in-LDS vectors (bucket_cap 120). */
int mi355q_bfp_gemm_aligned(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w,
                            const float* bias, float* y, int64_t M, int64_t N, int64_t K, int64_t ldy,
                            void* stream);
/* ... with the residual add of the caller's decoder layer in the store (round 6, ABI 24; the bf16 flavour has had it since ABI 22:
 * mi355q_bf16_gemm_tiled_res): y = (x . w^T + bias) + residual, the two sums rounded one after the other like the separate torch add
 * (modeling_opt.py:375, modeling_llama.py:259) -- the same bits.  On the ONE-launch route of row-aligned operands only (120-entry
 * buckets on both sides, K % 128 == 0), else MI355Q_E_UNSUPPORTED (the caller adds).  residual fp32 [M, ldr], ldr % 4 == 0. */
int mi355q_bfp_gemm_aligned_res(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w, const float* bias, const float* residual,
                                int64_t ldr, float* y, int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream);

/* ---- W4A4 / W5A5 block_fp Linear on the MX scaled matrix instruction (ABI 21) -------------------------------------------
 * replaces: quantized_modules/linear.py:59-76 (F.linear(x_q, W_q, b_q)) at the widths of
 *           experiments/emnlp/configs/quantization/bfp_4bit.toml and the section-4.4 search
 *           (experiments/emnlp/configs/search/opt_1.3b_sst2.toml:24-37): block_fp mantissas of <= 4 bits are exact in FP6 e2m3,
 *           the [1,16] blocks' shared exponents go into the E8M0 scales of v_mfma_scale_f32_16x16x128_f8f6f4 (twice the int8
 *           MFMA rate per unit of K): no row alignment, no exception lists.
 * mi355q_block_fp_quantize_mx: x [rows, K] fp32 (K % 128 == 0, K <= MI355Q_ROW_ALIGN_MAX_K, width <= 5; the weights go
 *   through the same call with their own parameters) -> three planes in the product's tile order, each
 *   mi355q_mx_plane_bytes(rows, K, plane) bytes (plane 0: 16 of a lane's 24 code bytes, 1: the other 8, 2: scales); 16-byte
 *   aligned.  The two blocks of a 32-group share a scale; the block with the larger exponent carries its mantissas shifted
 *   left by the difference.  A group whose blocks lie too far apart for that (> 3 exponents at W4, > 2 at W5) RAISES *bad
 *   (`bad_to_clear`, != bad, is zeroed by the kernel: callers alternate between two words, each call clearing the next
 *   call's; with bad_to_clear == NULL -- a caller recording a HIP graph, whose words cannot alternate between replays -- the
 *   call zeroes *bad itself by a stream-ordered memset in front of its kernel).
 * mi355q_mx_gemm: y = x_q . w_q^T (+ bias), fp32 accumulation.  bad2[0] / bad2[1]: the flag words of x / w.  If either is
 *   raised the launch forms the exact product from x_fp32 (quantised in registers with x's parameters) and w_fp32 (the
 *   fake-quantised weights) itself -- decided on the device, uniform over the grid, ~10x slower: callers move such a layer to
 *   another route (quantized_modules/linear.py here: `auto`).  Results within fp32 accumulation of the exact integer
 *   contraction either way. */
size_t mi355q_mx_plane_bytes(int64_t rows, int64_t K, int32_t plane);
int mi355q_block_fp_quantize_mx(const float* x, uint8_t* codes16, uint8_t* codes8, uint8_t* scales, int32_t* bad, int32_t* bad_to_clear,
                                int64_t rows, int64_t K, int32_t width, int32_t exponent_width, int32_t exponent_bias, void* stream);
int mi355q_mx_gemm(const uint8_t* x16, const uint8_t* x8, const uint8_t* xs, const uint8_t* w16, const uint8_t* w8, const uint8_t* ws,
                   const int32_t* bad2, const float* x_fp32, const float* w_fp32, const float* bias, float* y, int64_t M, int64_t N,
                   int64_t K, int64_t ldy, int32_t x_width, int32_t x_exponent_width, int32_t x_exponent_bias, void* stream);

/* The contraction in TWO COLUMN CLASSES in one launch (round 6, ABI 23) -- for activations with OUTLIER CHANNELS, the case the
 * reference's README (README.md:9-11, docs/images/fig-1.png) is about: a few input channels tens of times larger than the rest.
 * The block columns that hold such a channel lie several exponents above their rows' window and their exponents follow ONE
 * element's magnitude, so no row window of the int8 container holds them (K / 64 channels x 60: 20 % of all blocks are
 * exceptions) and the whole layer had to take the bf16 flavour at half the int8 rate.  Here the caller splits the columns of x
 * and W (quantized_modules/linear.py:59-76: F.linear contracts over in_features in any order) into class 0 -- K0 values, row-
 * aligned int8 operands with their exception lists, exactly what mi355q_bfp_gemm_aligned takes -- and class 1 -- K1 values as
 * tiled bf16 operands (mi355q_block_fp_quantize_bf16_tiled / mi355q_bf16_tile), every block with its own exponent:
 *     y = sx sw (x0 . w0^T) [int8 MFMA, int32 sums turned into fp32 in place] + x1 . w1^T [bf16 MFMA, same registers] + bias
 * K0 % 128 == 0, K0 >= 256, K1 % 64 == 0, K1 >= 128, 120-entry buckets on both class-0 operands, else MI355Q_E_UNSUPPORTED.
 * Exact products, fp32 accumulation across the class boundary (the reference's fp32 GEMM sums in fp32 throughout); an
 * overflowed class-0 bucket turns the launch into its own exact tile-by-tile fallback like mi355q_bfp_gemm_aligned's. */
/* ... and the activation side of it in ONE pass over x [rows, K] fp32: every [1,16] block quantised exactly as
 * mi355q_block_fp_quantize_aligned_rows does, then sent where the caller's column map says -- colmap[kb] (uint16, one word per
 * block column) = position | class << 15: class 0 -> block `position` of the row-aligned int8 operand [rows, 16 n0_blocks]
 * (mant_tiled / exp_out / rowflag / rowscale / list as for mi355q_block_fp_quantize_aligned_rows; the row's exponent is decided
 * over its class-0 blocks only; exception entries record the block's position in THAT operand), class 1 -> block `position` of
 * the tiled bf16 operand [rows, 16 n1_blocks] (mi355q_bfp_tiled_bytes(rows, 32 n1_blocks) bytes), every block with its own
 * exponent.  n0_blocks % 4 == 0, n1_blocks % 2 == 0 (whole 64-byte K-steps); positions are a permutation within each class;
 * buffers zero-initialised once by the caller (rows past `rows` up to the next multiple of 128 are read by the product). */
int mi355q_block_fp_quantize_classes(const float* x, const uint16_t* colmap, int64_t n0_blocks, int64_t n1_blocks, int8_t* mant_tiled,
                                     uint8_t* exp_out, uint8_t* rowflag, float* rowscale, int32_t* list, int32_t* list_to_clear,
                                     void* x1_bf16_tiled, int64_t rows, int64_t K, int32_t width, int32_t exponent_width,
                                     int32_t exponent_bias, int32_t bucket_cap, void* stream);
int mi355q_bfp_gemm_mixed(const mi355q_bfp_operand* x0, const mi355q_bfp_operand* w0, const void* x1_bf16_tiled, const void* w1_bf16_tiled,
                          const float* bias, float* y, int64_t M, int64_t N, int64_t K0, int64_t K1, int64_t ldy, void* stream);

/* The gated MLP's first half as ONE launch with the consumer's operand as its only output (round 6, ABI 23; reference:
 * modeling_llama.py:216  down_proj(act_fn(gate_proj(x)) * up_proj(x)),  quantized_modules/linear.py:59-76 for each Linear).
 * `w`: gate_proj's and up_proj's row-aligned operands INTERLEAVED in chunks of 16 rows -- rows 32 c .. 32 c + 15 = gate rows
 * 16 c .. 16 c + 15, rows 32 c + 16 .. 32 c + 31 = the same rows of up (tiled mantissas, exponents, row flags / scales and the
 * exception list's row numbers alike; `bias` interleaved the same way or NULL) -- so that a lane of the tile kernel holds matching
 * elements of both products.  The store epilogue forms h = silu(gate) * up from the fp32 values the plain epilogue would have
 * stored (torch's arithmetic: x / (1 + exp(-x)), then the product, each rounded to fp32), runs the CONSUMER's block_fp quantiser
 * (q_width / q_exponent_width / q_exponent_bias of down_proj's data_in, [1,16] blocks along I = a fragment pair's 16 columns)
 * and writes the tiled bf16 operand [M, I] (mi355q_bfp_tiled_bytes(M, 2 I) bytes, zero-initialised once) that
 * mi355q_bf16_gemm_tiled reads -- bit for bit what mi355q_block_fp_quantize_bf16_tiled_pre(gate, up, silu_mul) makes of the two
 * fp32 tensors, which are never written.  `scratch`: fp32 [M, 2 I], touched on the slow paths only (an overflowed bucket, more
 * exception entries than a tile's LDS holds: the product lands there and the tile is converted behind its add-backs).
 * K % 128 == 0, K >= 256, I % 32 == 0, 120-entry buckets, q_width <= 9, else MI355Q_E_UNSUPPORTED. */
int mi355q_bfp_gemm_aligned_gated(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w_gate_up, const float* bias_gate_up, float* scratch,
                                  void* out_bf16_tiled, int64_t M, int64_t I, int64_t K, int32_t q_width, int32_t q_exponent_width,
                                  int32_t q_exponent_bias, void* stream);

/* ... and OPT's MLP the same way (modeling_opt.py:412-420: fc2(activation_fn(fc1(x))), relu): x . w^T + bias, relu and the CONSUMER's
 * block_fp quantiser in the store epilogue (a fragment's 16 columns are one [1,16] block), the consumer's tiled bf16 operand [M, N]
 * (mi355q_bfp_tiled_bytes(M, 2 N) bytes) as the only output -- bit for bit mi355q_block_fp_quantize_bf16_tiled_pre(relu) of the
 * product.  `scratch` fp32 [M, N]: slow paths only.  K % 128 == 0, K >= 256, N % 32 == 0. */
int mi355q_bfp_gemm_aligned_relu(const mi355q_bfp_operand* x, const mi355q_bfp_operand* w, const float* bias, float* scratch,
                                 void* out_bf16_tiled, int64_t M, int64_t N, int64_t K, int32_t q_width, int32_t q_exponent_width,
                                 int32_t q_exponent_bias, void* stream);

/* Several weight operands of the SAME shape against ONE activation operand in one launch (the q / k / v projections of an
 * attention block, gate / up of a gated MLP: reference modules called one after the other on the same input,
 * modeling_opt.py:231-245, modeling_llama.py:216,283-287): y[i] = x . w[i]^T + bias[i].  The column tiles of all of them
 * form one grid, which fills the 256 compute units where the single products leave them idle (2048 x 2048 -> 2048: 128
 * tiles each) or waste most of a second round (2048 x 4096 -> 11008: 344 tiles each).  count <= 3; row-aligned operands with
 * 120-entry buckets, K % 128 == 0, else MI355Q_E_UNSUPPORTED (callers launch mi355q_bfp_gemm_aligned per weight).
 * Results equal the separate launches' bit for bit. */
int mi355q_bfp_gemm_aligned_multi(const mi355q_bfp_operand* x, const mi355q_bfp_operand* const* w, const float* const* bias,
                                  float* const* y, int32_t count, int64_t M, int64_t N, int64_t K, int64_t ldy, void* stream);

/* ---- block_fp quantised batched matmul ------------------------------------------------------------
 * replaces: quantized_functions/matmul.py:146-196 (generic_matmul_block_fp behind matmul_block_fp / bmm_block_fp,
 *           :300-353) for operands flattened to 3-D:  out[b] = Qx(x[b]) @ Qy(y[b]),
 *   x fp32 [B, M, K] quantised in [1,16] blocks along K (data_in_* parameters), y fp32 [B, K, N] in [1,16] blocks along
 *   N (weight_* parameters), out fp32 [B, M, N]; all contiguous.  x is read once and quantised in registers on its way
 *   into bf16 MFMAs (exact: widths <= 9); y is quantised into `workspace` (mi355q_bfp_matmul_workspace_bytes) first.
 *   fp32 accumulation; elements |x| <= 1e-8 (passed through unquantised by the reference) enter the product rounded
 *   to bf16, at most 2e-11 absolute each.  K % 16 == 0, N % 16 == 0, widths <= 9, B <= 65535, else MI355Q_E_UNSUPPORTED (callers then
 *   use mi355q_block_fp_quantize on both operands + their own GEMM). */
size_t mi355q_bfp_matmul_workspace_bytes(int64_t B, int64_t K, int64_t N);
int mi355q_bfp_matmul(const float* x, const float* y, float* out, void* workspace, int64_t B, int64_t M, int64_t K,
                      int64_t N, int32_t x_width, int32_t x_exponent_width, int32_t x_exponent_bias, int32_t y_width,
                      int32_t y_exponent_width, int32_t y_exponent_bias, void* stream);

/* The same with the attention's softmax stage folded in:
 *   out[b] = Qx(softmax(max(scores[b] + mask, finfo.min), dim = -1)) @ Qy(y[b])
 * -- replaces `attn_weights = attn_weights + attention_mask; attn_weights = max(attn_weights, finfo.min);
 * attn_probs = softmax(attn_weights); bmm_1(attn_probs, value_states)` of the reference's callers
 * (models/opt_quantized/modeling_opt.py:262-312, models/llama_quantized/modeling_llama.py:318-344) without the masked
 * scores or the probabilities [heads, T, T] ever being written.  mask: additive fp32 [M, K] shared by all batches, or
 * NULL; causal != 0: keys behind query i's horizon i + K - M are masked (what a causal mask's finfo.min entries do) and
 * the steps behind a row block's horizon are skipped altogether (their probabilities are exactly 0).  Scores are read
 * three times (two statistics passes that hit L2, one main pass); softmax is fp32 exp(x - max) / sum like torch's, the
 * exponential to ~1 ulp.  K > 192 (shorter rows: softmax + mi355q_bfp_matmul), N <= 128. */
int mi355q_bfp_softmax_matmul(const float* scores, const float* mask, int32_t causal, const float* y, float* out, void* workspace,
                              int64_t B, int64_t M, int64_t K, int64_t N, int32_t x_width, int32_t x_exponent_width,
                              int32_t x_exponent_bias, int32_t y_width, int32_t y_exponent_width, int32_t y_exponent_bias,
                              void* stream);

/* ---- block_minifloat / block_log quantised batched matmul (ABI 19) ---------------------------------
 * replaces: quantized_functions/matmul.py:199-249 (generic_matmul_block_minifloat behind matmul_block_minifloat /
 *           bmm_block_minifloat) and :252-297 (generic_matmul_block_log behind matmul_block_log / bmm_block_log), operands
 *           flattened to 3-D like mi355q_bfp_matmul's -- the same two kernels (y packed transposed, x quantised in registers
 *           on its way into bf16 MFMAs, fp32 accumulation) with the other block quantisers:
 *   block_minifloat: out[b] = Qx(x[b]) @ Qy(y[b]); both operands in [1,16] blocks, shared bias per block
 *     (block_minifloat.py:13-79); width - exponent_width - 1 <= 7 mantissa bits (exact in bf16), else MI355Q_E_UNSUPPORTED.
 *     The softmax variant is mi355q_bfp_softmax_matmul's with these quantisers.
 *   block_log: out[b] = Qx(x[b]) @ y[b] -- the reference quantises x ONLY (matmul.py:278-297 passes y through).  x becomes
 *     signed powers of two (exact in bf16); y enters as three bf16 planes hi + mid + lo = y exactly, so every product is
 *     exact and the result is the fp32 product's up to summation order.  All-zero blocks of x take the reference's
 *     tensor-wide fill (block_log.py:48-58 -> block_fp.py:54-58): a statistics pass over x finds it first.
 *     workspace: mi355q_block_log_matmul_workspace_bytes.
 * K % 16 == 0, N % 16 == 0, B <= 65535, 16-byte aligned pointers. */
int mi355q_block_minifloat_matmul(const float* x, const float* y, float* out, void* workspace, int64_t B, int64_t M, int64_t K,
                                  int64_t N, int32_t x_width, int32_t x_exponent_width, int32_t x_exponent_bias_width,
                                  int32_t y_width, int32_t y_exponent_width, int32_t y_exponent_bias_width, void* stream);
int mi355q_block_minifloat_softmax_matmul(const float* scores, const float* mask, int32_t causal, const float* y, float* out,
                                          void* workspace, int64_t B, int64_t M, int64_t K, int64_t N, int32_t x_width,
                                          int32_t x_exponent_width, int32_t x_exponent_bias_width, int32_t y_width,
                                          int32_t y_exponent_width, int32_t y_exponent_bias_width, void* stream);
size_t mi355q_block_log_matmul_workspace_bytes(int64_t B, int64_t K, int64_t N);
int mi355q_block_log_matmul(const float* x, const float* y, float* out, void* workspace, int64_t B, int64_t M, int64_t K,
                            int64_t N, int32_t x_width, int32_t x_exponent_bias_width, void* stream);

/* ---- the quantised attention core in one pass ------------------------------------------------------------------
 * replaces, in the reference's attention modules (models/opt_quantized/modeling_opt.py:246-312,
 * models/llama_quantized/modeling_llama.py:309-344), the sequence
 *     w = bmm_0(q, k^T)  [ / scale_div ]   w = max(w + mask, finfo.min)   p = softmax(w, -1)   out = bmm_1(p, v)
 * with the four block_fp quantisers of the two products (matmul.py:146-196: [1,16] blocks along each operand's last dim,
 * data_in_* parameters for q and p, weight_* parameters for k^T and v) applied exactly where the reference applies them.
 * q fp32 [B, M, D], k and v fp32 [B, T, D] (k UNtransposed), out fp32 [B, M, D]; mask additive fp32 [M, T] or NULL;
 * causal != 0: query i sees keys 0 .. i + T - M; scale_div: 0 = none (OPT scales q beforehand), else scores / scale_div
 * (Llama: sqrt(head_dim)).  qk_params / pv_params: {x width, exponent width, exponent bias, y width, exponent width,
 * exponent bias} of bmm_0 / bmm_1.  Neither scores nor probabilities are written anywhere: a workgroup keeps the score
 * strip of its 16 queries in MFMA accumulators (T <= 2048), or forms the scores twice -- statistics, then probabilities --
 * with the K / V fragments staged through LDS for 64 queries at a time (any T).  T % 16 == 0, D % 32 == 0, D <= 128,
 * widths <= 9, else MI355Q_E_UNSUPPORTED (callers then chain mi355q_bfp_matmul and mi355q_bfp_softmax_matmul).  Key tiles
 * behind the horizon of a workgroup's last query are skipped (probabilities exactly 0 there). */
size_t mi355q_bfp_attention_workspace_bytes(int64_t B, int64_t T, int64_t D);
/* Which kernel serves mi355q_bfp_attention: 0 = by size (default), 1 = scores resident in registers, four waves share a
 * query group's keys (T <= 2048), 2 = streaming (scores formed twice, any T), 3 = resident with eight key-waves (head_dim
 * <= 64; otherwise as 1).  Returns the previous setting.  For A/B runs and tests. */
int mi355q_bfp_attention_set_kernel(int which);
/* Whether the launch in front of the attention kernels (the one that quantises k and v into MFMA fragments) also leaves the quantised
 * Q fragments of the first product, which the kernels then LOAD -- 1 (default): where it pays, T <= 2048, 64 <= M <= T, head_dim 128 or
 * head_dim 64 with T > 1024
 * (every key-wave of a query group otherwise forms all of the group's Q fragments from q itself: 169 -> 156 us at [32, 2048, 128]);
 * 0: never (before round 6); 2: wherever the fragments fit (head_dim 64 / 128, 64 <= M <= T; tests).  With the rotary embedding
 * (mi355q_bfp_attention_rope) the fragments are always packed.  The same bits either way.  Returns the previous setting (ABI 24). */
int mi355q_bfp_attention_set_qpack(int on);
int mi355q_bfp_attention(const float* q, const float* k, const float* v, const float* mask, int32_t causal, float scale_div,
                         float* out, void* workspace, int64_t B, int64_t M, int64_t T, int64_t D, const int32_t* qk_params,
                         const int32_t* pv_params, void* stream);
/* The same on strided operands -- the [heads, T, D] views of [T, heads, D] projections that the models hand over, without
 * a contiguous copy first -- and out written where the out-projection reads it ([T, heads, D], i.e. out batch stride D,
 * row stride heads * D: no transpose copy behind the kernel either): strides = {q batch, q row, k batch, k row, v batch,
 * v row, out batch, out row} in elements (innermost stride 1, multiples of 4), NULL = all contiguous [B, rows, D]. */
int mi355q_bfp_attention_strided(const float* q, const float* k, const float* v, const float* mask, int32_t causal, float scale_div,
                                 float* out, void* workspace, int64_t B, int64_t M, int64_t T, int64_t D,
                                 const int32_t* qk_params, const int32_t* pv_params, const int64_t* strides, void* stream);
/* ... with the ROTARY POSITION EMBEDDING of q and k applied on load (round 6, ABI 24) -- modeling_llama.py:289-299 turns q and k between
 * the projections and the first product (quantized_functions/rotary_positional_encoding.py:59-82):
 *     x' = x * cos[p] + rotate_half(x) * sin[p],   p = position_ids[batch][row]
 * mi355q_rope_apply does that as one launch that moves q and k through memory once more (8 B per element, 27 us per Llama-7B layer at
 * 2048 tokens); here the K pack applies it to the values it is about to quantise and the attention kernels to their Q fragments --
 * the same fp32 arithmetic (both products rounded on their own, then the sum): the same bits as mi355q_rope_apply followed by
 * mi355q_bfp_attention_strided, and the turned q / k never exist in memory.  cos / sin: the caller's quantised tables, fp32
 * [table_rows, D]; position_ids int64 [B / heads, M] (clamped into the table); q, k are [B = batch x heads, rows, D].  cos == NULL: no
 * embedding (= mi355q_bfp_attention_strided).  M == T and D = 64 or 128, else MI355Q_E_UNSUPPORTED. */
int mi355q_bfp_attention_rope(const float* q, const float* k, const float* v, const float* mask, int32_t causal, float scale_div,
                              float* out, void* workspace, int64_t B, int64_t M, int64_t T, int64_t D, const int32_t* qk_params,
                              const int32_t* pv_params, const int64_t* strides, const float* cos, const float* sin,
                              const int64_t* position_ids, int64_t table_rows, int32_t heads, void* stream);
/* ... and with the OUT-PROJECTION's operand as the output (round 6, ABI 24).  Behind the attention core both models reshape to
 * [tokens, heads x D] and call out_proj / o_proj (modeling_opt.py:318-328, modeling_llama.py:349-353); where that Linear runs on the
 * per-block-exponent route its first step is mi355q_block_fp_quantize_bf16_tiled on the fp32 attention output.  A [1,16] block of that
 * quantiser is one query's column tile of one head -- four lanes of the kernels' store epilogue -- so with out_bf16_tiled != NULL the
 * kernels quantise there (consumer_params = the consumer's data_in {width, exponent width, exponent bias}) and write the tiled bf16 operand
 * [M, B x D] (head order; mi355q_bfp_tiled_bytes(M, 2 B D) bytes; ONE batch element: B = heads) that mi355q_bf16_gemm_tiled(_res) reads:
 * bit for bit what the separate quantiser makes of the fp32 output, which is never written (`out` may be NULL then).  cos == NULL: no
 * rotary embedding; out_bf16_tiled == NULL: = mi355q_bfp_attention_rope.
 * q_scale != 0: q is multiplied by it on its way into the Q fragments -- OPT's `self.q_proj(hidden_states) * self.scaling`
 * (modeling_opt.py:231), one fp32 multiply like the torch kernel that otherwise writes the scaled q: the same bits, 8 B per element of q
 * less through memory.  Served by the pack launch: head_dim 64 / 128, M <= T, no rotary embedding, else MI355Q_E_UNSUPPORTED. */
int mi355q_bfp_attention_fused(const float* q, const float* k, const float* v, const float* mask, int32_t causal, float q_scale,
                               float scale_div, float* out, void* out_bf16_tiled, const int32_t* consumer_params, void* workspace, int64_t B, int64_t M,
                               int64_t T, int64_t D, const int32_t* qk_params, const int32_t* pv_params, const int64_t* strides,
                               const float* cos, const float* sin, const int64_t* position_ids, int64_t table_rows, int32_t heads,
                               void* stream);

/* ---- the un-blocked quantisers -------------------------------------------------------------------------------------
 * replaces: quantizers/minifloat.py:134-196 (minifloat_ieee_quantizer: implicit leading one, subnormals at the lowest
 *           exponent), :21-86 (minifloat_denorm_quantizer: no implicit one, exponent ceil(log2(|x| + 1e-9)) per element)
 *           and quantizers/log.py:22-56 (log_quantizer: sign * 2^clamp(rint(log2(|x| + 0.1 * 2^-bias)))) behind
 *           QUANTIZER_MAP["minifloat_ieee" | "minifloat_denorm" | "log"] (quantizers/__init__.py:8-16).
 * Element-wise over n fp32 values, one fixed exponent_bias (MI355Q_BIAS_DEFAULT: 2^(exponent bits - 1) - 1), bit-exact
 * with the reference (threshold tables instead of logarithms, like the block formats).  minifloats: |x| <= 1e-8 passes
 * through; log cannot represent 0.  Bound: HBM, 8 B per element. */
int mi355q_minifloat_quantize(const float* x, float* y, int64_t n, int32_t width, int32_t exponent_width, int32_t exponent_bias,
                              int32_t denorm, void* stream);
int mi355q_log_quantize(const float* x, float* y, int64_t n, int32_t width, int32_t exponent_bias, void* stream);

/* The capture sequence `stream` is recording into a HIP graph (non-zero, unique per capture), or 0 when it is not
 * capturing.  Host-side caches that skip a launch (the Python layer's reuse of an already quantised activation) are
 * only valid inside the capture -- or the eager stretch -- they were made in. */
unsigned long long mi355q_stream_capture_id(void* stream);

/* ---- rotary position embedding -------------------------------------------------------------------------------------
 * replaces: quantized_functions/rotary_positional_encoding.py:59-248 (apply_rotary_pos_emb_<arithmetic>; callers
 *           models/llama_quantized/modeling_llama.py:289-299) AFTER the caller has quantised the cos / sin tables with the
 *           arithmetic's own quantiser (two small tensors [table_rows, D]):
 *     q_out = q * cos_q[pos] + rotate_half(q) * sin_q[pos],   rotate_half(x) = cat(-x[D/2:], x[:D/2]),   the same for k
 * in one launch for both (position lookup included; the reference runs ten elementwise kernels and two gathers).  fp32,
 * products and sum rounded one by one like the reference's ops.  q [B, Hq, T, D] and k [B, Hk, T, D] by element strides
 * {batch, head, position} (innermost stride 1; multiples of 4), outputs contiguous [B, H, T, D]; position_ids int64 [B, T]
 * (clamped to the table).  D % 8 == 0. */
int mi355q_rope_apply(const float* q, const float* k, const float* cos_q, const float* sin_q, const int64_t* position_ids,
                      float* q_out, float* k_out, int64_t B, int64_t Hq, int64_t Hk, int64_t T, int64_t D, int64_t table_rows,
                      const int64_t* q_strides, const int64_t* k_strides, void* stream);

/* Kernel timing for benchmarks: when enabled, mi355q_bfp_gemm_aligned brackets its MAIN kernel (the
 * int32-chain GEMM, not the correction / fallback launches) with HIP events on the launch stream.
 * mi355q_gemm_timing_read synchronises the recorded events, returns their count and average / minimum
 * duration in milliseconds, and clears the record.  At most 4096 launches are recorded. */
int mi355q_gemm_timing_enable(int enable);
int mi355q_gemm_timing_read(int32_t* count, float* avg_ms, float* min_ms);

/* Which kernel mi355q_bfp_gemm_aligned dispatches to, for A/B benchmarking and tests only (returns the previous
 * value): 0 automatic; 2 the blockwise-exact kernel alone; 6 / 8 the int32-chain (groups) / row-scale (rows)
 * kernel alone, WITHOUT the exception add-back (timing the product of the rewritten operands). */
int mi355q_bfp_gemm_set_variant(int variant);

#ifdef __cplusplus
}
#endif
#endif /* MI355Q_H */
