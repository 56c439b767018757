// Internal declarations shared by the kernel translation units and the C-ABI wrapper.
#ifndef MI355Q_INTERNAL_H
#define MI355Q_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355q {

// One quantiser launch: `lead` planes of rows x cols fp32 tiled by b0 x b1 blocks.
struct QuantArgs {
    const float* x;
    float* y;          // fake-quantised fp32 (nullable for block_fp)
    uint16_t* ybf;     // the same as bf16 (block_fp vector path only; exact for widths <= 9), nullable
    int8_t* mant;      // block_fp signed mantissas (nullable)
    uint8_t* code;     // per block: biased shared exponent (bfp) or shared bias (bm / bl); nullable
    unsigned* ws;      // MI355Q_WORKSPACE_BYTES, zeroed
    long long lead, rows, cols;
    long long n_elems, n_blocks;
    long long nbr, nbc;   // blocks per plane along rows / cols
    int b0, b1;
    int e_min, e_max;     // bfp: clamp range of the shared exponent
    int code_bias;        // bfp: exponent_bias
    int span;             // bm: 2^exponent_width - 1 ; bl: 2^(width-1) - 1
    int bias_max;         // bm / bl: 2^exponent_bias_width - 1
    float shift, inv_shift, mant_max;   // 2^mbits, 2^-mbits, 2^mbits - 1
    unsigned flags;
    // what the two GEMM-operand quantisers (aligned rows, bf16 tiled) read: 0 x itself, MI355Q_PRE_RELU max(x, 0),
    // MI355Q_PRE_SILU_MUL silu(x) * x2 -- the elementwise step the reference's MLPs put in front of fc2 / down_proj;
    // MI355Q_PRE_RMSNORM (x * rsqrt(mean(x^2) + eps)) * weight -- LlamaRMSNorm in front of q / k / v and gate / up
    const float* x2;
    int pre_op;
    float pre_eps;        // MI355Q_PRE_RMSNORM / _LAYERNORM (aligned-rows quantiser only): x2 = the norm's weight [cols],
    const float* x3;      // pre_eps its epsilon, x3 = the LayerNorm's bias [cols] (null: none)
    // exact zero-block mode, [1,16] row blocks: kernel 1 leaves one 64-bit ballot per 64 float4 slots (bit 4 b set: block b
    // of those sixteen is all zero) for the fix-up pass, which then visits the zero blocks without reading x again
    unsigned long long* zmap;
    // aligned-rows quantiser only: x (and the second input of silu_mul) as P row segments, [P][rows][seg_len] with the
    // segments seg_stride elements apart (0: plain rows) -- mi355q_block_fp_quantize_aligned_rows_seg
    long long seg_len, seg_stride;
};

int launch_quant(const QuantArgs& a, int fmt, bool needs_fixup, hipStream_t st);
int launch_quant_bf16_tiled(const QuantArgs& a, uint16_t* yt, hipStream_t st, bool cast_only = false, int fmt = 0 /* FMT_BFP; 1 = FMT_BM */);
int launch_quant_align_rows(const QuantArgs& a, int8_t* mt, uint8_t* flag, float* rscale, int exp_offset, int* list,
                            int* list_to_clear, hipStream_t st, int bcap);
// fp32 -> three bf16 parts, six column segments, tile order (mi355q_split.hip)
int launch_fp32_split_tile(const float* x, uint16_t* yt, long long rows, long long K, int role, hipStream_t st);
// the class-aware activation quantiser of the mixed contraction (mi355q_quant_cls.hip): cmap[kb] = position | class << 15
int launch_quant_classes(const QuantArgs& a, const uint16_t* cmap, int n0, int n1, int8_t* mt, uint8_t* flag, float* rscale,
                         int exp_offset, int* list, int* list_to_clear, uint16_t* bt, hipStream_t st, int bcap);
int launch_bfp_qmatmul(const QuantArgs& ax, const QuantArgs& ay, const float* x, const float* y, float* out, void* yt,
                       long long B, long long M, long long K, long long N, hipStream_t st, bool softmax = false,
                       const float* mask = nullptr, long long causal_off = -1, int fmt = 0);
size_t attention_workspace_bytes(long long B, long long T, long long D);
int attention_set_kernel(int which);
int attention_set_qpack(int on);
int launch_bfp_attention(const QuantArgs& aq, const QuantArgs& ak, const QuantArgs& ap, const QuantArgs& av, const float* q,
                         const float* k, const float* v, const float* mask, float* out, void* workspace, long long B,
                         long long M, long long T, long long D, long long causal_off, float scale_div, hipStream_t st,
                         const long long* strides = nullptr, const float* rope_cos = nullptr, const float* rope_sin = nullptr,
                         const long long* rope_pos = nullptr, long long rope_rows = 0, int rope_heads = 1, uint16_t* out_tiled = nullptr,
                         const QuantArgs* ao = nullptr, float q_scale = 0.f);
struct RopeArgs {
    const float* x[2];          // q, k
    float* y[2];
    long long sb[2], sh[2], st[2];      // element strides of batch, head, position (innermost stride 1)
    long long heads[2];
    const float* cos;           // [table_rows, D], already quantised
    const float* sin;
    const long long* pos;       // [B, T]
    long long B, T, D, table_rows;
};
int launch_rope(const RopeArgs& a, hipStream_t st);
int launch_quant_flat(const QuantArgs& a, int fmt, int bias, hipStream_t st);     // fmt 1 minifloat_ieee, 2 log, 3 minifloat_denorm
int launch_integer(const float* x, float* y, long long n, float scale, float lo, float hi, hipStream_t st);

struct GemmArgs {
    const int8_t* xm;
    const uint8_t* xe;
    const int8_t* wm;
    const uint8_t* we;
    const float* bias;
    float* y;
    long long M, N, K, ldy;
    int scale_bias;   // subtracted from xe + we to get the power of two of a block product
    int row_mode;     // 1: operands are ROW-aligned (rowflag[row], bucketed exception lists)
    int x_off, w_off; // exponent_bias + mbits of each operand (scale_bias = x_off + w_off)
    // row mode: entries per exception bucket of each operand; x_post = x's entries are added by the row post-pass
    // (mi355q_gemm_post.hip) instead of the GEMM's in-LDS vectors
    int x_bcap, w_bcap, x_post;
    // split-K of the tile GEMM (under-filled grids): `splits` workgroups share a tile, each over a slice of K; raw
    // accumulator slabs [tile][split] and one arrival ticket per tile in a library-owned workspace (zero when idle)
    int splits;
    int dbg;          // diagnostic builds only (MI355Q_V8_DBG with MI355Q_V8_STAMPS): 1 no LDS-DMA, 2 no barrier, 4 no fragment reads
    void* slabs;
    int* tickets;
    // grouped launch of the tile GEMM (mi355q_bfp_gemm_aligned_multi): `ngroup` weight operands of the same shape against
    // ONE x; the column tiles of all of them form one grid (column tile tn belongs to weight tn / tiles_n).  0: off.
    int ngroup;
    const int8_t* g_wm[3];
    const uint8_t* g_we[3];
    const float* g_sw[3];
    const int* g_wlist[3];
    const uint8_t* g_wf[3];
    const float* g_bias[3];
    float* g_y[3];
    int x_mbits, w_mbits;         // mantissa bits of the operands (0: not given) -- the launcher's choice of kernel
    unsigned long long* stamps;   // diagnostic builds of the 256 x 256 tile kernel (MI355Q_V9_STAMPS): [workgroup][2][8] phase times
    // y = (x . w^T + bias) + resid: the residual add the callers' decoder layers put behind o_proj / fc2 / down_proj, in the store
    // (bf16 flavour: mi355q_bf16_gemm_tiled_res); [M, >= N] fp32, ldr elements a row, 16-byte aligned rows; null: none
    const float* resid;
    long long ldr;
    // bf16 flavour: x as `x_segs` column segments (the rank-major result of an all-gather of per-rank quantised slices,
    // mi355q_bf16_gemm_tiled_seg): segment s holds K-steps s * steps / x_segs .. of every row piece, x_seg_stride bytes apart
    int x_segs;
    long long x_seg_stride;
    // the mixed contraction (mi355q_gemm_v9.hip, MIXED; mi355q_bfp_gemm_mixed): class 1 of the columns as tiled bf16 operands --
    // xm1 [M, K1], wm1 [N, K1] -- behind the row-aligned int8 class 0 (xm / wm over K values); K1 = 0: off
    const int8_t* xm1;
    const int8_t* wm1;
    long long K1;
    // the gated epilogue (mi355q_gemm_v9g.hip, mi355q_bfp_gemm_aligned_gated): yb = the consumer's tiled bf16 operand [M, N / 2],
    // q_* = its block_fp quantiser (mantissa bits, clamp range of the shared exponent); null: off
    void* yb;
    int q_mbits, q_emin, q_emax;
    int epi_op;       // 1: silu(gate) * up on interleaved gate / up columns (yb [M, N / 2]); 2: relu (yb [M, N])
};
int launch_bfp_gemm(const GemmArgs& a, int variant, hipStream_t st);
int launch_bfp_gemm_aligned(const GemmArgs& a, const uint8_t* xf, const uint8_t* wf, const int* xlist,
                            const int* wlist, int list_cap, int guard, hipStream_t st);
int launch_bfp_align_rows(const int8_t* mi, const uint8_t* ei, int8_t* mt, uint8_t* eo, uint8_t* flag, float* rscale,
                          int exp_offset, int* list, long long rows, long long K, hipStream_t st, int bcap);
int launch_bfp_gemm_rowpost(const GemmArgs& a, const int* xlist, const int* wlist, const float* xscale, const float* wscale,
                            hipStream_t st);
int launch_bfp_gemm_v8(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist,
                       int list_cap, hipStream_t st, const uint8_t* xf = nullptr, const uint8_t* wf = nullptr);
// the mixed contraction of the 256 x 256 tile kernel (mi355q_gemm_v9m.hip): a.K1 / a.xm1 / a.wm1 set
int launch_bfp_gemm_v9_mixed(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist, hipStream_t st,
                             const uint8_t* xf, const uint8_t* wf);
// benchmark timing (mi355q_gemm_timing): a launcher that can hands these to hipExtLaunchKernelGGL -- the events then carry the DISPATCH's own
// start / end timestamps (what rocprofv3 reports), not the times two marker packets around it complete -- and clears `start` to say so
struct KernelEvents { hipEvent_t start, stop; };
extern KernelEvents g_kernel_events;
int launch_bfp_gemm_v9_gated(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist, hipStream_t st,
                             const uint8_t* xf, const uint8_t* wf);
int launch_bfp_gemm_v9_resid(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist, hipStream_t st,
                             const uint8_t* xf, const uint8_t* wf);
int launch_bfp_pack_bits(const int8_t* mant, uint16_t* out, long long rows, long long K, int width, hipStream_t st);
int launch_bfp_expand(int mode, const uint16_t* packed, const uint8_t* codes, void* out, long long rows, long long K, int width,
                      int off, hipStream_t st, const uint8_t* rowexp = nullptr, uint8_t* exp_out = nullptr);
int launch_bf16_gemm_tiled(const GemmArgs& a, hipStream_t st);
// W4A4 / W5A5 on the MX scaled matrix instruction (mi355q_mx.hip; operand planes: mi355q_quant.hip, MxOut)
int launch_quant_mx_rows(const QuantArgs& a, uint8_t* c16, uint8_t* c8, uint8_t* sc, int* bad, int* bad_clear, hipStream_t st);
struct MxGemmArgs {
    const uint8_t *x16, *x8, *xs, *w16, *w8, *ws;     // the two operands
    const int* bad;                                   // [2]: raised by the quantisers of x / w
    const float *xf, *wf;                             // the fp32 tensors the operands were made from (x: NOT quantised, w: fake-quantised)
    const float* bias;
    float* y;
    long long M, N, K, ldy;
    QuantArgs qx;                                     // x's quantiser parameters (the exact route)
};
int launch_mx_gemm(const MxGemmArgs& a, hipStream_t st);

}  // namespace mi355q
#endif
