// mi355q_gemm_v9.hip -- the 256 x 256 tile of the row-scale block-floating-point GEMM (gfx950), round 3.
//
//     y[m,n] = sx[m] * sw[n] * ( sum_k xm'[m,k] * wm'[n,k] )  (+ bias[n])  (+ exception blocks),  K <= 16384, K % 128 == 0
// on ROW-aligned tiled operands (mi355q_align_row.h, mi355q_gemm_v2.h); the arithmetic of mi355q_gemm_v8.hip, whose
// 128-row tile and staggered schedules stay there.  Reference path: quantized_modules/linear.py:59-76 (F.linear on the
// fake-quantised operands).  What is different from the v8 kernel, each change measured in tools/ubench/kloop.hip
// (profiles/r03_kloop_*.txt):
//   * LDS rings of FOUR A halves and THREE B halves (16 KiB each, 112 KiB): two K-steps of LDS-DMA in flight instead of
//     one.  A step's B fragments are read one step ahead of its A fragments (they wait in registers), so a B half is dead
//     one barrier earlier than the A half of the same step and three B slots give the flight time of four A slots.
//   * LDS-DMA by buffer_load ... lds: per-lane offsets fixed for the whole loop, the K-step in the scalar offset (no
//     per-piece 64-bit vector adds); steps past the end are requested out of bounds (no memory traffic, zeros).
//   * one filler (fragment read / LDS-DMA piece) in front of each MFMA instead of clumps in front of groups of four.
//   * operands swapped in the MFMA: a lane then holds four CONSECUTIVE columns of one row, the epilogue stores 16 bytes
//     per lane (a quarter of the store instructions, the store tail 2 us shorter).
//   * exception add-back without a prologue of its own: the K loop starts as soon as its first stage has landed; the
//     tile's bucket bookkeeping runs in K-step 1 and from K-step 2 on every wave serves one entry per K-step (gather the
//     other operand's 256 blocks at the entry's K position into registers in one step, multiply and accumulate the
//     entry's vector of 256 products in spare LDS in the next), hidden behind the other wave of its SIMD.  The entries of
//     one tile row / column are chained in ascending block order and summed by ONE wave in that order: reproducible.
//     The epilogue looks rows up once per 16-row fragment and columns once per wave.
//   * (round 4's FIX_ = 2 -- the add-back READ from vectors the activation quantiser formed -- measured a net loss, the
//     quantiser paying more than the product saved, profiles/r04_corr_breakdown.txt, and was removed in round 5.)
// Roofline: int8 MFMA, 2*M*N*K ops.  y leaves as full fp32: 64 MiB at 4096^2, ~10 us at the rate the fabric takes
// write-backs, none of it overlapped with one tile per compute unit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <map>
#include <hip/hip_ext.h>
#include <mutex>
#include <type_traits>
#include <utility>

#include "mi355q_gemm_tile.h"
#ifdef V9_GATED_TU
#include "mi355q_quant_dev.h"
#endif

// The mixed contraction (MIXED below) is built as its OWN translation unit, mi355q_gemm_v9m.hip = this file with V9_MIXED_TU
// defined: the accumulators become a parameter of the K-step body there (it runs on int32 and on fp32 registers), which in this
// unit would change the register allocation of the kernels that ship since round 3 -- here the text of their K loop is untouched.
#ifdef V9_MIXED_TU
#define bfp_gemm_v9 bfp_gemm_v9m
#define V9_ACCP auto& acc,
#define V9_ACCA(x) x,
#else
#define V9_ACCP
#define V9_ACCA(x)
#endif
// The GATED epilogue (round 6, mi355q_bfp_gemm_aligned_gated) is a third unit, mi355q_gemm_v9g.hip = this file with V9_GATED_TU
// defined: the product of x against the INTERLEAVED gate / up weights of a gated MLP (modeling_llama.py:216: down_proj(act(gate(x))
// * up(x))) never leaves as fp32 -- the store epilogue forms silu(gate) * up from the accumulators (a lane holds matching elements
// of both: rows of the two weights alternate in chunks of 16), runs the CONSUMER's block_fp quantiser on the [1,16] blocks (= the
// 16 columns of a fragment pair) and writes the tiled bf16 operand down_proj's product reads: 2 bytes per value instead of two fp32
// tensors written and read back by a separate quantiser launch.
#ifdef V9_GATED_TU
#define bfp_gemm_v9 bfp_gemm_v9g
#endif
// The int8 product WITH the caller's residual add in its one-pass store epilogue (round 6, mi355q_bfp_gemm_aligned_res) is a fourth unit,
// mi355q_gemm_v9r.hip = this file with V9_RESID_TU: four lines in that epilogue, which in this unit moved 2 230 hunks of the headline
// kernel's text (same register count, another allocation) -- so here they are not compiled.
#ifdef V9_RESID_TU
#define bfp_gemm_v9 bfp_gemm_v9r
#endif

namespace mi355q {
#ifdef V9_GATED_TU
constexpr bool V9_GATED = true;
#else
constexpr bool V9_GATED = false;
#endif

constexpr int V9_HALF = 256 * 64, V9_NA = 4, V9_NB = 3, V9_NT = 512, V9_NW = 8;
constexpr int V9_B0 = V9_NA * V9_HALF;                       // B ring behind the A ring
constexpr int V9_STAGES = (V9_NA + V9_NB) * V9_HALF;         // 112 KiB
constexpr int V9_XB = 0, V9_WB = V9_XB + 4096, V9_MAP = V9_WB + 4096;      // (offsets in the side area)
constexpr int V9_SXT = V9_MAP + 2048, V9_SWT = V9_SXT + 1024, V9_BIAS = V9_SWT + 1024;
constexpr int V9_FLAGS = V9_BIAS + 1024, V9_OVF = V9_FLAGS + 256, V9_HDR = V9_OVF + 512, V9_CORR = V9_HDR + 256;
constexpr int V9_LDS = 159 * 1024, V9_SIDE = V9_LDS - V9_STAGES;
constexpr int V9_GLUT_BYTES = V9_GATED ? 1280 : 0;            // (gated epilogue: the consumer quantiser's log2 threshold table, at the end of the side area)
[[maybe_unused]] constexpr int V9_GLUT = V9_SIDE - V9_GLUT_BYTES;
constexpr int V9_FAST_MAX = (V9_SIDE - V9_CORR - V9_GLUT_BYTES) / 1024;       // entries (x + w) whose vectors fit beside the rings
constexpr int V9_TPRE = 24;                                  // entries whose gathers ride in the three K-steps past the end (96 pieces)
constexpr int V9_TVEC = 16;                                  // ... vectors beyond V9_FAST_MAX: in the ring half the last K-step leaves dead
constexpr int V9_NB_ENT = 2;                                 // entries a wave gathers per batch behind the K loop
constexpr int V9_GSCR = V9_NW * V9_NB_ENT * 4096;            // ... their blocks: scratch at the start of the ring area
constexpr int V9_SLOW_MAX = (V9_STAGES - V9_GSCR) / 1024;    // vectors that fit the ring area behind the K loop
static_assert(ROW_BUCKET_WORDS * 4 <= 4096, "bucket copy");
static_assert(V9_FAST_MAX >= (V9_GATED ? 31 : 32), "spare LDS for correction vectors");

typedef __bf16 v9_bf16x8 __attribute__((ext_vector_type(8)));
// (w fragment as the MFMA's A operand, x fragment as its B operand: D[n = 4 (lane / 16) + r][m = lane % 16])
__device__ __forceinline__ i32x4 v9_mma(const i32x4& fw, const i32x4& fx, const i32x4& c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(fw, fx, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 v9_mma(const i32x4& fw, const i32x4& fx, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v9_bf16x8, fw), __builtin_bit_cast(v9_bf16x8, fx), c, 0, 0, 0);
}

#define V9_WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define V9_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define V9_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
#define V9_SB() __builtin_amdgcn_sched_barrier(0)
// LDS-DMA as inline assembly: the compiler must not know that LDS-DMA is pending -- it orders every LDS read of its own
// behind ALL of it (s_waitcnt vmcnt(0)), which would drain the operand stream at each step of the exception service.
// M0 (the LDS destination of the wave's lane 0) is written in the statement that uses it; nothing else in this kernel
// keeps a value in M0.  16 / 4 bytes per lane by 64-bit lane address, 16 / 4 by buffer descriptor + lane offset + scalar offset.
#define V9_GLDS16(gp, lds) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds) : "memory")
#define V9_GLDS4(gp, lds) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gp), "s"(lds) : "memory")
#define V9_BLDS16(vo, rs, so, lds) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so), "s"(lds) : "memory")
#define V9_BLDS4(vo, rs, so, lds) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so), "s"(lds) : "memory")
__device__ __forceinline__ i32x4 v9_desc(const void* base, int bytes) {       // raw buffer descriptor of `bytes` bytes at `base`
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    return i32x4{(int)(unsigned)b, (int)(unsigned)(b >> 32), bytes, 0x00020000};
}

#ifdef V9_GATED_TU
// ---- the gated epilogue's quantiser: the arithmetic of bfp_quant_bf16_tiled_kernel (mi355q_quant.hip: block_fp.py:54-96 with the
//      all-zero-block fill 1) on values a caller holds in registers.  `m`: the block's largest |value| as a bit pattern; the N
//      values of `h` that this lane holds of it become their fake-quantised selves (exact in bf16 for widths <= 9).
struct GatedQ { int mbits, e_min, e_max; float mant_max, shift, inv_shift; };
template <int N>
__device__ __forceinline__ void gated_quant(float (&h)[N], unsigned m, const unsigned* __restrict__ glut, const GatedQ& q) {
    const float bm1 = m != 0u ? __uint_as_float(m) : 1.0f;
    const int k = __builtin_amdgcn_frexp_expf(bm1) - 1;
    const unsigned f = __float_as_uint(__builtin_amdgcn_frexp_mantf(bm1)) & 0x7FFFFFu;
    const int e = clampi(k + ((f != 0u && f >= glut[lut_index(k)]) ? 1 : 0), q.e_min, q.e_max);
    const int up = q.mbits - e;
    if (up >= 28) {                                       // blocks below 2^-23: the general rule (quant_elem<FMT_BFP>)
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float x = h[t], ax = fabsf(x), sg = sgn(x + EPS9);
            const float r = __builtin_ldexpf(ax + EPS9, -e) * q.shift;
            const float mm = clampf(__builtin_rintf(r), 0.f, q.mant_max);
            const float v = __builtin_ldexpf(sg, e) * (mm * q.inv_shift);
            h[t] = ax <= ATOL ? x + 0.0f : v;
        }
    } else {
        constexpr float MAGIC = 12582912.0f;
        const float sc = __builtin_ldexpf(1.0f, up), es = EPS9 * sc, inv = __builtin_ldexpf(1.0f, -up);
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float x = h[t];
            const float r = __builtin_amdgcn_fmed3f(__builtin_fmaf(x, sc, __builtin_copysignf(es, x)), -q.mant_max, q.mant_max);
            const float v = __builtin_fmaf(r + MAGIC, inv, -MAGIC * inv);
            h[t] = fabsf(x) <= ATOL ? x : v;
        }
    }
}
// where the 8 bytes of the four values at h column hc (a multiple of 4) of `row` go in the consumer's tiled bf16 operand
// [rows, I]: pieces of 16 rows x 32 values, [8-value group 0..3][row][16 bytes]
__device__ __forceinline__ unsigned gated_offset(unsigned row, unsigned hc, unsigned kpI) {
    return ((row >> 4) * kpI + (hc >> 5)) * 1024u + ((hc & 31u) >> 3) * 256u + (row & 15u) * 16u + ((hc & 7u) >> 2) * 8u;
}
// Slow paths (an overflowed bucket: the blockwise-exact product; more entries than a tile's LDS holds: atomics behind the stores):
// the fp32 tile [m0, m0 + bm) x [n0, n0 + bn) of the interleaved product lies in the scratch a.y; one thread per (row, 16 h
// columns) reads its gate and up values back and does what the register epilogue does.
__device__ __forceinline__ void gated_post_tile(const GemmArgs& a, const unsigned* __restrict__ glut, const GatedQ& q, long long m0,
                                                long long n0, int bm, int bn, int tid, int nthreads) {
    if (a.epi_op == 2) {                                          // relu: h = max(y, 0), a block = 16 columns of y itself
        const int hb1 = bn >> 4;
        const unsigned kpN = (unsigned)(a.N >> 5);
        for (int it = tid; it < bm * hb1; it += nthreads) {
            const long long row = m0 + it / hb1, col = n0 + (long long)(it % hb1) * 16;
            if (row >= a.M || col >= a.N) continue;
            const float* yr = a.y + row * a.ldy + col;
            float h[16];
            unsigned m = 0u;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                h[t] = pre_relu(yr[t]);
                m = max(m, __float_as_uint(h[t]) & 0x7FFFFFFFu);
            }
            gated_quant<16>(h, m, glut, q);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(a.yb) + gated_offset((unsigned)row, (unsigned)col + 4 * g4, kpN)) =
                    make_uint2(pack_bf16(h[4 * g4], h[4 * g4 + 1]), pack_bf16(h[4 * g4 + 2], h[4 * g4 + 3]));
        }
        return;
    }
    const int hb = bn >> 5;                                       // h blocks per tile row
    const unsigned kpI = (unsigned)(a.N >> 6);                    // I / 32 pieces per 16 rows (I = N / 2)
    for (int it = tid; it < bm * hb; it += nthreads) {
        const long long row = m0 + it / hb, col = n0 + (long long)(it % hb) * 32;      // (gate: col .. col + 15, up: col + 16 .. + 31)
        if (row >= a.M || col >= a.N) continue;
        const float* yr = a.y + row * a.ldy + col;
        float h[16];
        unsigned m = 0u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            h[t] = pre_silu_mul(yr[t], yr[16 + t]);
            m = max(m, __float_as_uint(h[t]) & 0x7FFFFFFFu);
        }
        gated_quant<16>(h, m, glut, q);
        const unsigned hc = (unsigned)(col >> 1);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(a.yb) + gated_offset((unsigned)row, hc + 4 * g4, kpI)) =
                make_uint2(pack_bf16(h[4 * g4], h[4 * g4 + 1]), pack_bf16(h[4 * g4 + 2], h[4 * g4 + 3]));
    }
}
#endif

// FIX_ 1: with the exception add-back formed by the tile itself behind its K loop.  STAMP: diagnostic build, phase times go
// to a.stamps.
// MIXED (round 6): the contraction in TWO column classes in one launch -- class 0 (a.K values: a.xm / a.wm, row-aligned int8 with
// their exception lists) on the int8 MFMA into int32 accumulators, which are then turned into fp32 IN PLACE (times the row and
// column scales), and class 1 (a.K1 values: a.xm1 / a.wm1, tiled bf16, every block its own exponent) on the bf16 MFMA into the
// same registers.  For activations with OUTLIER CHANNELS (README.md:9-11, figure 1: a few input channels tens of times larger
// than the rest): the block columns that hold such a channel lie several exponents above their rows' window and their exponents
// follow one element's magnitude -- no row window of the int8 container holds them (20 % of all blocks at K / 64 channels x 60),
// so the whole layer used to run at the bf16 rate.  Here only those columns do: K1 / K of the MFMA work at half rate.
template <int FIX_, bool BF16, bool STAMP, bool MIXED = false>
__global__ __launch_bounds__(V9_NT, 1) void bfp_gemm_v9(const GemmArgs a_in, const float* __restrict__ sx,
                                                        const float* __restrict__ sw_in, const int* __restrict__ xlist,
                                                        const int* __restrict__ wlist_in, const uint8_t* __restrict__ xf,
                                                        const uint8_t* __restrict__ wf_in) {
    constexpr int FIX = FIX_ != 0 ? 1 : 0;
    constexpr bool TPF = FIX_ == 1;                       // the tile's gathers ride in the K-steps past the end (round 4)
    static_assert(!BF16 || FIX_ == 0, "the bf16 arithmetic has no exception lists");
    static_assert(!MIXED || (FIX_ == 1 && !BF16), "the mixed contraction is the int8 kernel with its lists + a bf16 tail");
    // Two LDS objects: the operand rings (filled by LDS-DMA, read by inline-asm ds_read_b128 only) and everything else.
    // The compiler orders its own LDS reads behind every LDS-DMA that may alias them -- with one array each of its reads
    // in the K loop (the exception service) would drain the operand stream (s_waitcnt vmcnt(0)).
    __shared__ __attribute__((aligned(16))) unsigned char ring[V9_STAGES];
    __shared__ __attribute__((aligned(16))) unsigned char side[V9_SIDE];
    unsigned char* const smem = side;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, l16 = lane & 15, lq = lane >> 4;
    unsigned long long st_t[6] = {0, 0, 0, 0, 0, 0}, st_x[3] = {0, 0, 0}, st_y[2] = {0, 0};
    if (STAMP) st_t[0] = __builtin_amdgcn_s_memrealtime();

    GemmArgs a = a_in;
    const float* __restrict__ sw = sw_in;
    const int* __restrict__ wlist = wlist_in;
    const uint8_t* __restrict__ wf = wf_in;
    const int ngroup = a.ngroup > 1 ? a.ngroup : 1;
    const int Mi = (int)a.M, Ni = (int)a.N;
    const int tiles_m = (Mi + 255) >> 8, tiles_n1 = (Ni + 255) >> 8, tiles_n = tiles_n1 * ngroup;
    const int S = a.splits > 1 ? a.splits : 1;                 // workgroups per tile (split-K)
    const int nwg = tiles_m * tiles_n * S;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int split = S > 1 ? pid % S : 0, tile_id = S > 1 ? pid / S : pid;
    const int GM = 4, in_group = GM * tiles_n, group_id = tile_id / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (tile_id % in_group) % gsz;
    int tn = (tile_id % in_group) / gsz;
    if (ngroup > 1) {
        const int which = tn / tiles_n1;                                  // (wave-uniform: scalar loads from the argument block)
        tn -= which * tiles_n1;
        // (selects over constant indices: a runtime index would put the argument block in scratch memory)
#define V9_PICK(f) (which == 0 ? a_in.f[0] : which == 1 ? a_in.f[1] : a_in.f[2])
        a.wm = V9_PICK(g_wm); a.we = V9_PICK(g_we); a.bias = V9_PICK(g_bias); a.y = V9_PICK(g_y);
        sw = V9_PICK(g_sw); wlist = V9_PICK(g_wlist); wf = V9_PICK(g_wf);
#undef V9_PICK
    }
    const int m0 = tm * 256, n0 = tn * 256;
    const int kp = (int)(a.K >> 6);                             // 1-KiB pieces per 16 rows
    const int kstep0 = S > 1 ? (int)((long long)kp * split / S) : 0;       // this workgroup's slice of the K-steps
    // (MIXED: no split-K; nsteps0 int8 K-steps of 64 values, then kp1 bf16 K-steps of 32 values -- both even, kp1 >= 4)
    const int kp1 = MIXED ? (int)(a.K1 >> 5) : 0;
    const int nsteps0 = S > 1 ? (int)((long long)kp * (split + 1) / S) - kstep0 : kp;
    const int nsteps = nsteps0 + kp1;

    int* xb = reinterpret_cast<int*>(smem + V9_XB);
    int* wb = reinterpret_cast<int*>(smem + V9_WB);
    int* rowslot = reinterpret_cast<int*>(smem + V9_MAP);
    int* colslot = rowslot + 256;
    float* sxt = reinterpret_cast<float*>(smem + V9_SXT);
    float* swt = reinterpret_cast<float*>(smem + V9_SWT);
    float* bst = reinterpret_cast<float*>(smem + V9_BIAS);
    float* corr = reinterpret_cast<float*>(smem + V9_CORR);

#ifdef V9_GATED_TU
    // (the consumer quantiser's log2 thresholds, read behind the K loop only; loaded by plain code BEFORE any LDS-DMA is in flight:
    //  the compiler waits for its own loads with vmcnt(0), which would drain the operand stream anywhere later)
    unsigned* const glut = reinterpret_cast<unsigned*>(smem + V9_GLUT);
    if (tid < LUT_N) glut[tid] = mi355q_log2_ceil_thr[tid];
    const GatedQ gq{a.q_mbits, a.q_emin, a.q_emax, (float)((1 << a.q_mbits) - 1), (float)(1 << a.q_mbits), 1.0f / (float)(1 << a.q_mbits)};
#endif
    // ---- in front of the operand stream (same queue, so landed by the first counted wait): the tile's scale / bias
    //      slices, its two exception buckets and the lists' overflow words
    const int ring_lds = (int)(size_t)(lptr_t)ring, side_lds = (int)(size_t)(lptr_t)side;     // the objects' own LDS addresses
    if (wave == 0 || wave == 1) {
        if (FIX && !(wave == 0 && a.x_post)) {
            const int* b = wave == 0 ? row_bucket(xlist, m0) : row_bucket(wlist, n0);
            const int d = side_lds + (wave == 0 ? V9_XB : V9_WB);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q * 256 + lane * 4 < ROW_BUCKET_WORDS) V9_GLDS16(b + q * 256 + lane * 4, d + q * 1024);
        }
    } else if (!BF16 && wave == 2) {
        V9_GLDS16(sx + m0 + lane * 4, side_lds + V9_SXT);
    } else if (!BF16 && wave == 3) {
        V9_GLDS16(sw + n0 + lane * 4, side_lds + V9_SWT);
    } else if (wave == 4) {
        // (bounds-checked by the descriptor: columns past N read as zero, no address past the array is touched)
        if (a.bias) {
            const i32x4 rb = v9_desc(a.bias + n0, (Ni - n0) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) V9_BLDS4(lane * 4 + q * 256, rb, 0, side_lds + V9_BIAS + q * 256);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) bst[q * 64 + lane] = 0.f;
        }
    } else if (FIX && (wave == 5 || wave == 6)) {
        V9_GLDS4((wave == 5 ? xlist : wlist) + lane, side_lds + V9_OVF + (wave - 5) * 256);
    }
    // maps cleared, vectors beside the rings zero (every product is ADDED to its row's / column's vector)
    auto clear_maps_and_vectors = [&]() {
        rowslot[tid & 255] = -1;
        if (tid >= 256) colslot[tid & 255] = -1;
        if (tid == 0) reinterpret_cast<int*>(smem + V9_FLAGS)[16] = 0;      // (set by the bookkeeping: a row with several entries)
#pragma unroll
        for (int q = 0; q < (V9_FAST_MAX * 1024 + V9_NT * 16 - 1) / (V9_NT * 16); ++q)
            if ((q * V9_NT + tid) * 16 < V9_FAST_MAX * 1024)
                *reinterpret_cast<f32x4*>(smem + V9_CORR + (q * V9_NT + tid) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    if (FIX) clear_maps_and_vectors();

    // ---- the operand stream.  Piece p of a K-step: 16 rows of A (p < 16) or of B; this wave stages pieces wave + 8 q.
    //      One descriptor per operand, rooted at the tile's first piece row and ending with its last (rows past the
    //      operand read as zero); the lane's part of the address is fixed, the K-step rides in the scalar offset.
    const long long row_bytes = (long long)kp * 1024;           // one piece row (16 rows x K bytes)
    const int pa_rows = min(16, ((Mi + 127) >> 7) * 8 - (m0 >> 4)), pb_rows = min(16, ((Ni + 127) >> 7) * 8 - (n0 >> 4));
    const int8_t* xbase = a.xm + (long long)(m0 >> 4) * row_bytes + (long long)kstep0 * 1024;
    const int8_t* wbase = a.wm + (long long)(n0 >> 4) * row_bytes + (long long)kstep0 * 1024;
    // (the range check covers the lane's offset only, not the scalar one: a K-step past the end of the slice is requested
    //  through a descriptor of zero bytes -- no memory traffic, zeros land in the slot)
    const int x_nrec = (int)(pa_rows * row_bytes), w_nrec = (int)(pb_rows * row_bytes);
    // MIXED: the class-1 operands (tiled bf16: kp1 pieces of 16 rows x 32 values per piece row)
    const long long row_bytes1 = (long long)kp1 * 1024;
    const int8_t* xbase1 = MIXED ? a.xm1 + (long long)(m0 >> 4) * row_bytes1 : nullptr;
    const int8_t* wbase1 = MIXED ? a.wm1 + (long long)(n0 >> 4) * row_bytes1 : nullptr;
    const int x_nrec1 = (int)(pa_rows * row_bytes1), w_nrec1 = (int)(pb_rows * row_bytes1);
    int voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) voff[q] = (wave + 8 * (q & 1)) * (int)row_bytes + lane * 16;
    // (the lane offsets of a class-1 piece differ only in the piece row's length: one scalar add per piece)
    const int dvo0 = MIXED ? wave * (int)(row_bytes1 - row_bytes) : 0, dvo1 = MIXED ? (wave + 8) * (int)(row_bytes1 - row_bytes) : 0;
    // piece q (literal) of K-step `step` into ring slots at byte offsets sa (A) / sb (B)
#define V9_PIECE(q, rxd, rwd, soff, sa, sb)                                                                             \
    V9_BLDS16(MIXED ? voff[(q) & 3] + (c1_ ? (((q) & 1) ? dvo1 : dvo0) : 0) : voff[q], (q) < 2 ? rxd : rwd, soff,           \
              ring_lds + ((q) < 2 ? (sa) : V9_B0 + (sb)) + (wave + 8 * ((q) & 1)) * 1024)
#define V9_DESCS(step)                                                                                                  \
    const bool more_ = (step) < nsteps;                                                                                 \
    const bool c1_ = MIXED && (step) >= nsteps0;                                                                        \
    const i32x4 rxd_ = v9_desc(c1_ ? xbase1 : xbase, more_ ? (c1_ ? x_nrec1 : x_nrec) : 0),                             \
                rwd_ = v9_desc(c1_ ? wbase1 : wbase, more_ ? (c1_ ? w_nrec1 : w_nrec) : 0);                             \
    const int soff_ = (c1_ ? (step) - nsteps0 : (step)) * 1024;
#define V9_STAGE(step, sa, sb) { V9_DESCS(step) V9_PIECE(0, rxd_, rwd_, soff_, sa, sb); V9_PIECE(1, rxd_, rwd_, soff_, sa, sb); V9_PIECE(2, rxd_, rwd_, soff_, sa, sb); V9_PIECE(3, rxd_, rwd_, soff_, sa, sb); }
    V9_STAGE(0, 0, 0)
    V9_STAGE(1, V9_HALF, V9_HALF)
    V9_STAGE(2, 2 * V9_HALF, 2 * V9_HALF)
    if (STAMP) st_t[1] = __builtin_amdgcn_s_memrealtime();

    using acc_t = typename std::conditional<BF16, f32x4, i32x4>::type;

    int cx = 0, cw = 0, nent = 0, mode = 0;   // exception entries of this tile (set behind the K loop)

    // lane-constant part of the fragment addresses; fragment i is i KiB further (immediate offset)
    const int va = ring_lds + piece_lds_off(wm * 128 + l16, lq), vb = ring_lds + V9_B0 + piece_lds_off(wn * 64 + l16, lq);
    i32x4 fa[4], fb0[4], fb1[4];
    // (1) of the exception add-back, the bookkeeping: 16 lanes share an entry: slot = the list index of the first entry of the
    //     same tile row / column; -2 marks a void entry (also for the atomics pass of mode 3); rows / columns without a vector
    //     keep -1 in the maps.  Needs the two buckets in LDS; writes entry word 3 and the maps (all beside the rings).
    auto bookkeep = [&](int cx, int nent) {
        for (int i0 = 0; i0 < nent; i0 += V9_NT / 16) {    // uniform
            const int i = i0 + (tid >> 4), sub = tid & 15;
            const bool valid = i < nent, is_x = i < cx;
            int* e = v8_entry(xb, wb, cx, valid ? i : 0);
            const int r = e[0], base = is_x ? m0 : n0;
            const bool live = valid && (is_x ? (r >= m0 && r < m0 + 256 && r < Mi) : (r >= n0 && r < n0 + 256 && r < Ni));
            const int lo = is_x ? 0 : cx, hi = is_x ? cx : nent;
            // (slot = the list index of the row's entry with the SMALLEST BLOCK: a property of the data, not of the order in
            //  which rows reserved their list slots -- the sums below start from it and go on in ascending block order)
            int skey = (e[1] << 8) | i;
            if (live)
                for (int j = lo + sub; j < hi; j += 16) {
                    const int* ej = v8_entry(xb, wb, cx, j);
                    if (ej[0] == r) skey = min(skey, (ej[1] << 8) | j);
                }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) skey = min(skey, __shfl_xor(skey, o));
            const int slot = skey & 255;
            if (valid && sub == 0) {
                e[3] = live ? slot : -2;
                if (live && slot == i) (is_x ? rowslot : colslot)[r - base] = i;
                if (live && slot != i) reinterpret_cast<int*>(smem + V9_FLAGS)[16] = 1;     // (a row with several entries)
            }
        }
    };
    // TPF: the bookkeeping runs HERE, in the shadow of the first stages' flight: the buckets, scales and header words were
    // requested in front of the operand pieces, so twelve outstanding LDS-DMA instructions mean they have landed
    bool early_bk = false;
    if (TPF && !(a.dbg & 16)) {                                 // (dbg 16, A/B: bookkeeping behind the loop)
        V9_WAITV(12);
        V9_LGKM(0);                                             // (this thread's share of the cleared maps has landed)
        __builtin_amdgcn_s_barrier();
        const int ecx = a.x_post ? 0 : __builtin_amdgcn_readfirstlane(min(xb[0], ROW_BCAP));
        const int ent = ecx + __builtin_amdgcn_readfirstlane(min(wb[0], ROW_BCAP));
        const int* ovf0 = reinterpret_cast<const int*>(smem + V9_OVF);
        if (ent > 0 && ent <= min(128, V9_FAST_MAX + V9_SLOW_MAX) && __builtin_amdgcn_readfirstlane(ovf0[0] | ovf0[64]) == 0) {
            bookkeep(ecx, ent);
            early_bk = true;
        }
        V9_LGKM(0);
    }
    V9_WAITV(8);                                            // everything but the pieces of K-steps 1 and 2
    __builtin_amdgcn_s_barrier();
    if (FIX) {
        // a bucket overflowed somewhere (uniform over the grid): the row-scale product does not apply; the workgroups of
        // this launch share the blockwise-exact product instead (the operand loads in flight land in LDS only)
        const int* ovf = reinterpret_cast<const int*>(smem + V9_OVF);
        if (__builtin_amdgcn_readfirstlane(ovf[0] | ovf[64]) != 0) {
            V9_WAITV(0);
            __syncthreads();
            if constexpr (MIXED) {
                // the mixed contraction's own fallback, tile by tile (no other workgroup touches this tile): first the class-1
                // product of this 256 x 256 tile -- fragments straight from memory (the tiled pieces ARE fragments: lane = (row,
                // 16-byte group)), no ring, slow and rare -- stored as it is; then the class-0 product blockwise-exact on top of
                // it (bfp_gemm_v2_tile with the stored values as its residual) and the tile's exception blocks (tile_fix_body)
                f32x4 fac[8][4];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) fac[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int lo_ = lq * 256 + l16 * 16;
                for (int t = 0; t < kp1; ++t) {
                    i32x4 fx[8], fw[4];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int pr = wm * 8 + i;                 // piece row of the tile (rows past the operand: zero)
                        fx[i] = pr < pa_rows ? *reinterpret_cast<const i32x4*>(xbase1 + (long long)pr * row_bytes1 + (long long)t * 1024 + lo_) : i32x4{0, 0, 0, 0};
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pr = wn * 4 + j;
                        fw[j] = pr < pb_rows ? *reinterpret_cast<const i32x4*>(wbase1 + (long long)pr * row_bytes1 + (long long)t * 1024 + lo_) : i32x4{0, 0, 0, 0};
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) fac[i][j] = v9_mma(fw[j], fx[i], fac[i][j]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const long long row = (long long)m0 + wm * 128 + i * 16 + l16;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int col = n0 + wn * 64 + j * 16 + lq * 4 + r;
                            if (row < a.M && col < Ni) a.y[row * a.ldy + col] = fac[i][j][r];
                        }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __threadfence();
                __syncthreads();
                if (threadIdx.x >= 256) return;            // (terminated waves do not take part in the barriers below)
                GemmArgs a2 = a;
                a2.resid = a.y;
                a2.ldr = a.ldy;
                for (int sub = 0; sub < 4; ++sub) {
                    const long long sm0 = m0 + (sub >> 1) * V2_BM, sn0 = n0 + (sub & 1) * V2_BN;
                    if (sm0 >= a.M || sn0 >= a.N) continue;        // (uniform)
                    bfp_gemm_v2_tile(a2, xf, wf, *reinterpret_cast<V2Smem*>(ring), sm0, sn0, (int)threadIdx.x);
                    __threadfence();
                    __syncthreads();
                    tile_fix_body(a, row_bucket(xlist, sm0, a.x_bcap), row_bucket(wlist, sn0, a.w_bcap), a.x_bcap, a.w_bcap, sm0, sn0,
                                  (int)threadIdx.x, 256);
                    __syncthreads();
                }
                return;
            }
#ifdef V9_GATED_TU
            // (the blockwise-exact product into the fp32 scratch, tile by tile as v8_fallback does it, each tile turned into the
            //  consumer's operand right behind its exception blocks)
            if (xlist[0] == 0 && wlist[0] == 0) return;
            if (threadIdx.x >= 256) return;
            {
                const int ntiles_ = (int)(((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN));
                for (int tile = (int)blockIdx.x; tile < ntiles_; tile += nwg) {
                    bfp_gemm_v2_body(a, xf, wf, *reinterpret_cast<V2Smem*>(ring), tile);
                    long long fm0, fn0;
                    v2_tile_origin(a, tile, fm0, fn0);
                    __threadfence();
                    __syncthreads();
                    tile_fix_body(a, row_bucket(xlist, fm0, a.x_bcap), row_bucket(wlist, fn0, a.w_bcap), a.x_bcap, a.w_bcap, fm0, fn0,
                                  (int)threadIdx.x, 256);
                    __threadfence();
                    __syncthreads();
                    gated_post_tile(a, glut, gq, fm0, fn0, V2_BM, V2_BN, (int)threadIdx.x, 256);
                    __syncthreads();
                }
            }
            return;
#else
            v8_fallback(a, xf, wf, xlist, wlist, ring, ngroup > 1 ? (tm * tiles_n1 + tn) * S + split : (int)blockIdx.x,
                        ngroup > 1 ? tiles_m * tiles_n1 * S : nwg);
            return;
#endif
        }
    }
    V9_DSR(fb0[0], vb, 0); V9_DSR(fb0[1], vb, 1024); V9_DSR(fb0[2], vb, 2048); V9_DSR(fb0[3], vb, 3072);
    V9_DSR(fa[0], va, 0); V9_DSR(fa[1], va, 1024);
    V9_SB();
    acc_t acc[8][4];
    f32x4 accf[MIXED ? 8 : 1][MIXED ? 4 : 1];               // MIXED: the same registers after the class boundary
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
    unsigned long long c_loop = 0;
    if (STAMP) { st_t[2] = __builtin_amdgcn_s_memrealtime(); c_loop = __builtin_amdgcn_s_memtime(); }

    // K-step t between barrier(t) and barrier(t + 1): MFMA group i (A fragment i against the four B fragments of the step,
    // held in registers) with, in front of its MFMAs: the read of A fragment i + 2 (the last two groups: fragments 0 and 1
    // of step t + 1) and the counted wait for fragment i; the read of B fragment i - 2 of step t + 1 (groups 2-5); LDS-DMA
    // piece i of step t + 3 (groups 0-3).  At barrier(t) every wave has waited for its own pieces of step t + 1 and has
    // retired every read of the A half of step t - 1 and of the B half of step t: those are the slots step t + 3 goes to.
    // FIX_ 1, round 4: the GATHERS of the tile's first V9_TPRE exception entries ride in the LDS-DMA slots of the THREE K-steps
    // past the end (tails 3, 4, 5; 96 pieces = 24 entries x 4 quarters of 64 rows): piece P = 32 j + p of tail step j carries
    // quarter P & 3 = wave & 3 of entry P >> 2 = 8 j + 2 q + wave / 4 -- the other operand's blocks at the entry's K position,
    // what `gather` below fetches behind the loop (5.5 us exposed, 8.8 for the fullest tile: profiles/r04_corr_breakdown.txt).
    // Only the entry's K position is needed to request it; the bookkeeping stays behind the loop.
    const int glane = (lane >> 4) * (int)row_bytes + (lane & 15) * 16;     // row `lane` of a quarter inside the tile's piece rows
    // (the descriptors are rebuilt where they are used: eight scalar registers less across the K loop)
#define V9_XG() v9_desc(a.xm + (long long)(m0 >> 4) * row_bytes, x_nrec)
#define V9_WG() v9_desc(a.wm + (long long)(n0 >> 4) * row_bytes, w_nrec)
    const int gv = glane + (wave & 3) * 4 * (int)row_bytes;
    int tcx = 0, tnent = 0;
    if (TPF) {
        tcx = a.x_post ? 0 : __builtin_amdgcn_readfirstlane(min(xb[0], ROW_BCAP));
        tnent = tcx + __builtin_amdgcn_readfirstlane(min(wb[0], ROW_BCAP));
    }
    auto body = [&](V9_ACCP i32x4 (&fb)[4], i32x4 (&fbn)[4], int t, int sa_c, int sa_n, int sb_n, int da, int db, const int tail) {
        V9_LGKM(2);                                             // (the B reads of the slot about to be refilled)
        V9_WAITV(4);
        __builtin_amdgcn_s_barrier();
        const int ac = va + sa_c, an = va + sa_n, bn = vb + sb_n;
        V9_DESCS(t + 3)
#define V9_GPIECE(q) { const int e_ = 8 * (tail - 3) + 2 * (q) + (wave >> 2);                                            \
        const int kb_ = __builtin_amdgcn_readfirstlane(v8_entry(xb, wb, tcx, min(e_, max(tnent - 1, 0)))[1]);            \
        const int koff_ = (kb_ >> 2) * 1024 + (kb_ & 3) * 256;                                                           \
        i32x4 gd_ = e_ < tcx ? V9_WG() : V9_XG();                                                                        \
        if (e_ >= tnent) gd_[2] = 0;                                                                                     \
        V9_BLDS16(gv, gd_, koff_, ring_lds + ((q) < 2 ? da : V9_B0 + db) + (wave + 8 * ((q) & 1)) * 1024); }
#define V9_GROUP(i, wait)                                                                                                \
        if (i < 6) V9_DSR(fa[(i + 2) & 3], ac, (i + 2) * 1024); else V9_DSR(fa[(i + 2) & 3], an, (i - 6) * 1024);          \
        V9_LGKM(wait);                                                                                                   \
        V9_SB();                                                                                                         \
        acc[i][0] = v9_mma(fb[0], fa[i & 3], acc[i][0]);                                                                 \
        V9_SB();                                                                                                         \
        if (i >= 2 && i < 6) V9_DSR(fbn[i - 2], bn, (i - 2) * 1024);                                                     \
        V9_SB();                                                                                                         \
        acc[i][1] = v9_mma(fb[1], fa[i & 3], acc[i][1]);                                                                 \
        V9_SB();                                                                                                         \
        if (i < 4) { if (TPF && tail >= 3) V9_GPIECE(i) else V9_PIECE(i, rxd_, rwd_, soff_, da, db); } \
        V9_SB();                                                                                                         \
        acc[i][2] = v9_mma(fb[2], fa[i & 3], acc[i][2]);                                                                 \
        V9_SB();                                                                                                         \
        acc[i][3] = v9_mma(fb[3], fa[i & 3], acc[i][3]);                                                                 \
        V9_SB();
        V9_GROUP(0, 2) V9_GROUP(1, 2) V9_GROUP(2, 2) V9_GROUP(3, 3)
        V9_GROUP(4, 4) V9_GROUP(5, 4) V9_GROUP(6, 4) V9_GROUP(7, 3)
    };
    // ring positions (byte offsets): A half of step t in a0, t + 1 in a1, ..., the slot step t + 3 goes to in a3;
    // B half of step t + 1 in b1, the slot step t + 3 goes to (= where step t's B half was) in b0
    int a0 = 0, a1 = V9_HALF, a2 = 2 * V9_HALF, a3 = 3 * V9_HALF, b0 = 0, b1 = V9_HALF, b2 = 2 * V9_HALF;
    // (the first two K-steps apart: their counted waits differ when a record is in flight)
#define V9_PAIR(ac_, t_, tl0, tl1)                                                                                       \
    body(V9_ACCA(ac_) fb0, fb1, t_, a0, a1, b1, a3, b0, tl0);                                                            \
    { const int o = a0; a0 = a1; a1 = a2; a2 = a3; a3 = o; }                                                             \
    { const int o = b0; b0 = b1; b1 = b2; b2 = o; }                                                                      \
    body(V9_ACCA(ac_) fb1, fb0, (t_) + 1, a0, a1, b1, a3, b0, tl1);                                                      \
    { const int o = a0; a0 = a1; a1 = a2; a2 = a3; a3 = o; }                                                             \
    { const int o = b0; b0 = b1; b1 = b2; b2 = o; }
    // (nsteps is even and >= 4: K % 128 == 0, even slices; TPF: the last two pairs request the gathers)
    if constexpr (MIXED) {
        // (the ring positions as functions of the step -- A half of step t in slot t % 4, B half in slot t % 3 -- instead of seven
        //  scalars rotated through three loops: with those the compiler lost track of their uniformity and handed the LDS-DMA
        //  statements vector registers for M0)
        int tb = 0;                                                 // t % 3
#define V9_PAIR_M(ac_, t_, tl0, tl1)                                                                                     \
        {                                                                                                                \
            const int ta_ = (t_);                                                                                        \
            const int A0_ = (ta_ & 3) * V9_HALF, A1_ = ((ta_ + 1) & 3) * V9_HALF, A2_ = ((ta_ + 2) & 3) * V9_HALF,       \
                      A3_ = ((ta_ + 3) & 3) * V9_HALF;                                                                   \
            const int tb1_ = tb == 2 ? 0 : tb + 1, tb2_ = tb1_ == 2 ? 0 : tb1_ + 1;                                      \
            body(V9_ACCA(ac_) fb0, fb1, ta_, A0_, A1_, tb1_ * V9_HALF, A3_, tb * V9_HALF, tl0);                          \
            body(V9_ACCA(ac_) fb1, fb0, ta_ + 1, A1_, A2_, tb2_ * V9_HALF, A0_, tb1_ * V9_HALF, tl1);                    \
            tb = tb2_;                                                                                                   \
        }
        // class 0 on the int8 MFMA (the requests three steps ahead run on into class 1's pieces by themselves) ...
        for (int t = 0; t < nsteps0; t += 2) {
            V9_PAIR_M(acc, t, 0, 0)
        }
        // ... the int32 sums become fp32 values where they stand (the fragments of class 1's first step are already on their
        // way into registers: let them land before the compiler's own code runs) ...
        V9_LGKM(0);
        V9_SB();
        {
            float sxr_[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) sxr_[i] = sxt[wm * 128 + i * 16 + l16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 swv = *reinterpret_cast<const f32x4*>(&swt[wn * 64 + j * 16 + lq * 4]);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accf[i][j][r] = (float)acc[i][j][r] * sxr_[i] * swv[r];
            }
        }
        V9_SB();
        // ... and class 1 goes on in the same registers on the bf16 MFMA, every block with its own exponent
        for (int t = nsteps0; t < nsteps - 4; t += 2) {
            V9_PAIR_M(accf, t, 0, 0)
        }
        V9_PAIR_M(accf, nsteps - 4, 0, 3)
        V9_PAIR_M(accf, nsteps - 2, 4, 5)
#undef V9_PAIR_M
        // (what the code behind the loop reads of the rotation: the slots of the three steps past the end and of the last step)
        a0 = (nsteps & 3) * V9_HALF; a1 = ((nsteps + 1) & 3) * V9_HALF; a2 = ((nsteps + 2) & 3) * V9_HALF; a3 = ((nsteps + 3) & 3) * V9_HALF;
        b0 = tb * V9_HALF; b1 = (tb == 2 ? 0 : tb + 1) * V9_HALF; b2 = (tb == 0 ? 2 : tb - 1) * V9_HALF;
    } else {
    for (int t = 0; t < (TPF ? nsteps - 4 : nsteps); t += 2) {
        V9_PAIR(acc, t, 0, 0)
    }
    if (TPF) {
        V9_PAIR(acc, nsteps - 4, 0, 3)
        V9_PAIR(acc, nsteps - 2, 4, 5)
    }
    }
    const int dead_a = a3;            // (the A half of the last K-step: nothing was requested into it, dead behind the loop)
    // ring slots (byte offsets) of the steps past the end: step nsteps + j went to a_j / b_j (the rotation above)
    const int ga_[3] = {a0, a1, a2}, gb_[3] = {b0, b1, b2};
    const i32x4 xg = V9_XG(), wg = V9_WG();
#undef V9_PAIR
    V9_WAITV(0);
    V9_LGKM(0);                                                 // (the compiler does not know these reads are in flight)
    V9_SB();
    if (STAMP) { st_t[3] = __builtin_amdgcn_s_memrealtime(); c_loop = __builtin_amdgcn_s_memtime() - c_loop; }
    __builtin_amdgcn_s_barrier();                               // (every wave is out of the rings)
    if constexpr (!MIXED)           // (the mixed contraction is never split: its int32 accumulators are dead behind the class boundary)
    if (S > 1) {
        // ---- split-K: every slice leaves its raw accumulators in its slab (16 bytes a lane, 1 KiB a wave instruction);
        //      the slice that arrives last at the tile's ticket sums all slabs IN SLICE ORDER (reproducible for the fp32
        //      flavour too; the int32 sums are exact in any order) and goes on to the epilogue, the others leave.
        //      Hand-off: plain stores, every wave's vmcnt(0), workgroup barrier, agent-scope release by one lane, relaxed
        //      ticket; the reducer acquires once, then loads plainly (cdna guide, Guideline 16).
        constexpr long long SLAB = 256ll * 256 * 4;
        acc_t* slab = reinterpret_cast<acc_t*>(static_cast<unsigned char*>(a.slabs) + ((long long)tile_id * S + split) * SLAB);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) slab[((wave * 8 + i) * 4 + j) * 64 + lane] = acc[i][j];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flagw = reinterpret_cast<int*>(smem + V9_FLAGS) + 8;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int t = __hip_atomic_fetch_add(&a.tickets[tile_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == S - 1 ? 1 : 0;
            if (last) {
                __hip_atomic_store(&a.tickets[tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // idle again
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flagw = last;
        }
        __syncthreads();
        if (*flagw == 0) return;
        const acc_t* tslabs = reinterpret_cast<const acc_t*>(static_cast<unsigned char*>(a.slabs) + (long long)tile_id * S * SLAB);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
        for (int sl = 0; sl < S; ++sl) {
            const acc_t* sp = tslabs + (long long)sl * (SLAB / 16);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += sp[((wave * 8 + i) * 4 + j) * 64 + lane];
        }
    }

    // ---- epilogue: y = float(acc) * sx[m] * sw[n] + bias[n] (+ exception blocks).  The lane holds, for tile (i, j) of its
    //      wave, row wm * 128 + 16 i + l16 and the four columns wn * 64 + 16 j + 4 lq + 0..3: one 16-byte store.
    //
    //      Exception blocks (blocks outside their row's exponent window, zero in the operand, listed exactly in the tile's
    //      two buckets; a few dozen per tile in the usual case) are added back here, behind the K loop, which does not know
    //      of them at all: every entry contributes one VECTOR of 256 products (x entry (row r, block kb): 2^(code - x_off) *
    //      sw[n] * dot16(entry, w'[n, kb]) for the tile's 256 columns n; w entries the mirror image over the rows), all
    //      entries of a tile row / column add into one vector (`slot` = the smallest list index of that row / column), and
    //      the stores of the rows / columns that have a vector add it.  Order of events: (1) bookkeeping -- slots and the
    //      row / column maps, one pass, one barrier; (2) every wave requests the blocks its entries need (wave = slot % 8;
    //      its entries in ascending (slot, block, index) order -- a property of the data, not of the order in which rows
    //      reserved their list slots: reproducible; LDS-DMA gathers of 4 KiB per entry into the ring area, free now); (3)
    //      vectors formed, one barrier; (4) the stores.  (Tried and not kept, profiles/r03_v9_exception_designs.txt: serving
    //      the entries while the K loop runs -- by gathers into scratch, or picking the blocks up from the operand rings
    //      as they stream by: every served entry stalls one wave for an LDS round trip and with it, at the next barrier,
    //      the workgroup; and storing the untouched tiles while the gathers fly: the gathers queue behind the stores.)
    float* rvec = reinterpret_cast<float*>(ring) + (V9_GSCR - V9_FAST_MAX * 1024) / 4;   // (vector of slot s >= V9_FAST_MAX)
    bool tmode = false;               // TPF: this tile's first V9_TPRE gathers are in the ring already
#define V9_VEC(s_) (((s_) < V9_FAST_MAX ? corr : rvec) + (s_) * 256)
    // (the two exponent offsets as opaque scalars: `is_x ? a.x_off : a.w_off` on the by-value argument struct becomes an indexed
    //  load of two adjacent fields, for which the compiler parks them in 16 bytes of scratch memory)
    int x_off_s = a.x_off, w_off_s = a.w_off;
    asm volatile("" : "+s"(x_off_s), "+s"(w_off_s));
    bool look = false;
    int mykeys0 = 0x7fffffff, mykeys1 = 0x7fffffff;                   // (slot << 18 | block << 8 | index) of the entries at list
                                                                // positions lane, lane + 64 that this wave serves
    if (FIX) {
        cx = a.x_post ? 0 : __builtin_amdgcn_readfirstlane(min(xb[0], ROW_BCAP));
        cw = __builtin_amdgcn_readfirstlane(min(wb[0], ROW_BCAP));
        nent = cx + cw;
        mode = nent == 0 ? 0 : (nent <= min(128, V9_FAST_MAX + V9_SLOW_MAX) ? 1 : 3);
        if (mode) {                                    // (uniform over the workgroup: the barrier below is met by all)
            if (!early_bk) bookkeep(cx, nent);
            V9_LGKM(0);
            __builtin_amdgcn_s_barrier();
            look = mode == 1 && !(a.dbg & 2);                   // (dbg 2, diagnostic: no add-back, results invalid)
            if (STAMP) st_x[0] = __builtin_amdgcn_s_memrealtime();
            // (the entries beyond the prefetched ones gather into 4-KiB chunks of the ring's six live halves: <= 24 of them)
            tmode = TPF && look && nent <= min(V9_FAST_MAX + V9_TVEC, V9_TPRE + 24) && !(a.dbg & 4);      // (dbg 4, A/B: gathers behind the loop)
            if (tmode && nent > V9_FAST_MAX) {
                // vectors of slots >= V9_FAST_MAX: the dead ring half (the prefetched gathers fill the others), zeroed --
                // their entries may be served in two phases, so every product is ADDED
                rvec = reinterpret_cast<float*>(ring + dead_a) - V9_FAST_MAX * 256;
                for (int o = tid * 16; o < V9_TVEC * 1024; o += V9_NT * 16)
                    *reinterpret_cast<f32x4*>(ring + dead_a + o) = f32x4{0.f, 0.f, 0.f, 0.f};
                V9_LGKM(0);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    // gathers of one entry: the other operand's 16-byte blocks at the entry's K position for the tile's 256 rows / columns,
    // four LDS-DMA quarters (rows past the operand read as zero) into 4 KiB of this wave's scratch
    // scratch chunk b of this wave (4 KiB: one entry's gathers).  tmode: anywhere in the ring but its dead half, which holds
    // vectors -- by then (phase B) the prefetched gathers have been consumed
    auto scratch = [&](int b) {
        int o = (wave * V9_NB_ENT + b) * 4096;
        if (tmode && o >= dead_a) o += V9_HALF;
        return o;
    };
    // where quarter c of entry i's gathered blocks lies: prefetched (piece P = 4 i + c of the tail steps) or in scratch chunk b
    auto gathered = [&](int i, int b, int c) {
        if (tmode && i < V9_TPRE) {
            const int P = 4 * i + c, j = P >> 5, pp = P & 31;
            const int base = pp < 16 ? (j == 0 ? ga_[0] : j == 1 ? ga_[1] : ga_[2]) : V9_B0 + (j == 0 ? gb_[0] : j == 1 ? gb_[1] : gb_[2]);
            return base + (pp & 15) * 1024;
        }
        return scratch(b) + c * 1024;
    };
    auto gather = [&](int i, int b) {
        if (tmode && i < V9_TPRE) return;                      // (already in the ring)
        const int kb = __builtin_amdgcn_readfirstlane(v8_entry(xb, wb, cx, i)[1]);
        const int koff = (kb >> 2) * 1024 + (kb & 3) * 256;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int vo = glane + c * 4 * (int)row_bytes, dl = ring_lds + scratch(b) + c * 1024;
            if (i < cx) { V9_BLDS16(vo, wg, koff, dl); } else { V9_BLDS16(vo, xg, koff, dl); }
        }
    };
    // one entry: multiply, add to its slot's vector (`first`: the vector is in the ring area and this is its first
    // entry); an x entry also takes the exception x exception terms (same K position in both lists)
    auto finish = [&](int i, int b, int slot, bool first) {
        const bool is_x = i < cx;
        const int* e = v8_entry(xb, wb, cx, i);
        const int kb = e[1], code = e[2];
        const int4 pv = *reinterpret_cast<const int4*>(e + 4);
        const int sh = code - (is_x ? x_off_s : w_off_s);
        const float* sc = is_x ? swt : sxt;
        float* v = V9_VEC(slot);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int4 q = *reinterpret_cast<const int4*>(ring + gathered(i, b, c) + lane * 16);
            const float p = __builtin_ldexpf((float)dot16(pv, q), sh) * sc[c * 64 + lane];
            v[c * 64 + lane] = first ? p : v[c * 64 + lane] + p;
        }
        if (is_x)
            for (int f0 = 0; f0 < cw; f0 += 64) {               // uniform
                const int fi = f0 + lane;
                if (fi < cw) {
                    const int* f = wb + EXC_HEADER + EXC_ENTRY * fi;
                    if (f[3] != -2 && f[1] == kb) {
                        const int d = dot16(pv, *reinterpret_cast<const int4*>(f + 4));
                        v[f[0] - n0] += __builtin_ldexpf((float)d, code + f[2] - a.scale_bias);
                    }
                }
            }
    };
    // the wave's next entry: the smallest key above `last` among the two this lane holds, over the wave
    // (`phase`: -1 all entries; 0 / 1 those whose gathers were / were not prefetched, tmode)
    auto next_key = [&](int last, int phase) {
        const bool in0 = phase < 0 || ((mykeys0 & 255) >= V9_TPRE) == (phase == 1);
        const bool in1 = phase < 0 || ((mykeys1 & 255) >= V9_TPRE) == (phase == 1);
        int best = mykeys0 > last && in0 ? mykeys0 : 0x7fffffff;
        if (mykeys1 > last && in1) best = min(best, mykeys1);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) best = min(best, __shfl_xor(best, o));
        return __builtin_amdgcn_readfirstlane(best);
    };
    if (look && !tmode) {
        // (2) this wave's entries
        // (two named scalars, not an array: a q loop the compiler keeps rolled would index it in scratch memory)
        auto key_of = [&](int j) {
            if (j >= nent) return 0x7fffffff;
            const int* f = v8_entry(xb, wb, cx, j);
            const int s3 = f[3];
            return s3 != -2 && (s3 & 7) == wave ? (s3 << 18) | (f[1] << 8) | j : 0x7fffffff;
        };
        mykeys0 = key_of(lane);
        mykeys1 = key_of(lane + 64);
    }
    // (masks only -- scales, bias and slots are read again where they are used: the accumulators take half the registers)
    // the maps the stores consult
    const int* const rslot_r = rowslot;
    const int* const cslot_r = rslot_r + 256;
    unsigned cmask = 0, jmask = 0;      // bit 4 j + r: some lane of the wave has a vector for that column; bit j: tile column j has one
    unsigned rmask = 0;                 // bit i: some row of fragment i has a vector (set by the one-pass epilogue from its preloads)
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.y) | (uintptr_t)(a.ldy * 4)) & 15) == 0;
    // tiles of fragment i: pass 0 the ones no vector touches, pass 1 the others (with their vectors)
    auto store_rows = [&](int i, int pass) {
        const bool rowv = (rmask >> i) & 1;
        if (pass == 0 && rowv) return;
        if (pass == 1 && !rowv && jmask == 0) return;
        const int rl = wm * 128 + i * 16 + l16;
        const float sxv = BF16 ? 1.f : sxt[rl];
        const int rs = rowv ? rslot_r[rl] : -1;
        const long long row = (long long)m0 + rl;
        float* yrow = a.y + row * a.ldy + n0 + wn * 64 + lq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool touched = rowv || ((jmask >> j) & 1);
            if ((pass == 1) != touched) continue;               // (uniform)
            const int cl = wn * 64 + j * 16 + lq * 4;
            const f32x4 swv = BF16 ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(&swt[cl]);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&bst[cl]);
            f32x4 val;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (MIXED) val[r] = accf[i][j][r] + bv[r];
                else val[r] = BF16 ? (float)acc[i][j][r] + bv[r] : (float)acc[i][j][r] * sxv * swv[r] + bv[r];
            }
            if (pass == 1) {
                if (rowv) {                                      // the row's vector: 256 products, one per tile column
                    const int rs0 = max(rs, 0);
                    const float* rv = V9_VEC(rs0);
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(rv + cl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[r] += rs >= 0 ? c4[r] : 0.f;
                }
                if ((jmask >> j) & 1) {                          // the columns' vectors: one product per tile row
                    const int4 c = *reinterpret_cast<const int4*>(&cslot_r[cl]);
                    const int c4[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (cmask & (1u << (4 * j + r)))
                            val[r] += c4[r] >= 0 ? V9_VEC(max(c4[r], 0))[rl] : 0.f;
                }
            }
            if (row < a.M) {
                const int col = n0 + cl;
                if (a.resid) {                                   // (uniform) the caller's residual add, in the store
                    const float* rr = a.resid + row * a.ldr + col;
                    if (col + 3 < Ni) val += *reinterpret_cast<const f32x4*>(rr);
                    else
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < Ni) val[r] += rr[r];
                }
                if (vec_ok && col + 3 < Ni) {
                    // (experiment, MI355Q_V9_DBG bits 8 / 16 / 32: write-through / system-scope / non-temporal stores)
                    if (a.dbg & 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(yrow + j * 16), "v"(val) : "memory");
                    else if (a.dbg & 16) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(yrow + j * 16), "v"(val) : "memory");
                    else if (a.dbg & 32) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(yrow + j * 16), "v"(val) : "memory");
                    else *reinterpret_cast<f32x4*>(yrow + j * 16) = val;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < Ni) yrow[j * 16 + r] = val[r];
                }
            }
            V9_SB();        // (tile by tile: the scheduler would otherwise pull the reads of many tiles ahead and spill)
        }
    };
    // (2), (3): batches of this wave's entries -- blocks requested (one round trip, exposed: the stores of the whole chip
    // start together behind it, and a gather issued beside them would queue behind the compute unit's own stores), vectors
    // formed; a second batch is rare
    if (STAMP) st_x[1] = __builtin_amdgcn_s_memrealtime();
    if (tmode) {
        // ---- round 4: ONE VECTOR PER ENTRY, formed by all waves at once.  Quarter c of entry i is an independent item (64
        //      products from blocks that are in the ring already): no sorting, no read-modify-write, nothing to wait for
        //      between items -- the serial service below (one wave per slot, entries in key order, a shuffle search per entry)
        //      took 3-4 us for ~20 entries, of which the gathers were 0.5 (profiles/r04_v9_tail_prefetch.txt).  Then the
        //      exception x exception terms as one pass over the (x entry, w entry) pairs, then the rows with several entries
        //      folded into their first one (smallest block) in ascending block order by one wave: same sums, same order.
        // A wave takes entries wave, wave + 8, wave + 16: the operands of all three are read before any product is formed (one
        // LDS round trip a wave, not one per entry), and the wave that wrote an x entry's vector adds that entry's exception x
        // exception terms right behind it (LDS operations of a wave complete in order: no barrier in between).
        struct Ent { int4 pv, q0, q1, q2, q3; float s0, s1, s2, s3; int e3, sh, kb, code; };
        auto load_entry = [&](Ent& t, int i, bool on, int o0, int o1, int o2, int o3) {
            const bool is_x = i < cx;
            const int* e = v8_entry(xb, wb, cx, on ? i : 0);
            t.e3 = on ? e[3] : -2;
            t.kb = e[1];
            t.code = e[2];
            t.sh = e[2] - (is_x ? x_off_s : w_off_s);
            t.pv = *reinterpret_cast<const int4*>(e + 4);
            const float* sc = is_x ? swt : sxt;
            t.q0 = *reinterpret_cast<const int4*>(ring + o0 + lane * 16); t.q1 = *reinterpret_cast<const int4*>(ring + o1 + lane * 16);
            t.q2 = *reinterpret_cast<const int4*>(ring + o2 + lane * 16); t.q3 = *reinterpret_cast<const int4*>(ring + o3 + lane * 16);
            t.s0 = sc[lane]; t.s1 = sc[64 + lane]; t.s2 = sc[128 + lane]; t.s3 = sc[192 + lane];
        };
        auto form_entry = [&](const Ent& t, int i) {
            if (t.e3 == -2 || (a.dbg & 64)) return;             // (uniform: void / outside the tile -- no vector)
            float* v = i < V9_FAST_MAX ? corr + i * 256 : rvec + i * 256;
            v[lane] = __builtin_ldexpf((float)dot16(t.pv, t.q0), t.sh) * t.s0;
            v[64 + lane] = __builtin_ldexpf((float)dot16(t.pv, t.q1), t.sh) * t.s1;
            v[128 + lane] = __builtin_ldexpf((float)dot16(t.pv, t.q2), t.sh) * t.s2;
            v[192 + lane] = __builtin_ldexpf((float)dot16(t.pv, t.q3), t.sh) * t.s3;
            if (i < cx && !(a.dbg & 32))                        // exception x exception: w entries at the same K position
                for (int f0 = 0; f0 < cw; f0 += 64) {           // (uniform)
                    const int fi = f0 + lane;
                    if (fi < cw) {
                        const int* f = wb + EXC_HEADER + EXC_ENTRY * fi;
                        if (f[3] != -2 && f[1] == t.kb)
                            v[f[0] - n0] += __builtin_ldexpf((float)dot16(t.pv, *reinterpret_cast<const int4*>(f + 4)), t.code + f[2] - a.scale_bias);
                    }
                }
        };
        const int npre = min(nent, V9_TPRE);
        // ONE copy of the code, looped (three inlined copies with all their reads in flight took longer: behind the K loop the
        // instruction stream is what costs), and no address arithmetic to speak of: the four quarters of entry wave + 8 k are
        // pieces 4 wave .. 4 wave + 3 of tail step k -- 4 KiB in a row in that step's A half (waves 0-3) or B half.
        {
            const int blk = ring_lds * 0 + (wave & 3) * 4096;
#pragma unroll 1
            for (int k = 0; k < 3; ++k) {
                const int i = wave + V9_NW * k;
                if (i >= npre) break;
                const int half = wave < 4 ? (k == 0 ? ga_[0] : k == 1 ? ga_[1] : ga_[2]) : V9_B0 + (k == 0 ? gb_[0] : k == 1 ? gb_[1] : gb_[2]);
                Ent t;
                load_entry(t, i, true, half + blk, half + blk + 1024, half + blk + 2048, half + blk + 3072);
                form_entry(t, i);
            }
        }
        if (STAMP) st_y[0] = __builtin_amdgcn_s_memrealtime();
        if (nent > V9_TPRE) {
            // the entries beyond the prefetched ones: their gathers now, all in one round trip; chunk k of the scratch (4 KiB,
            // anywhere in the ring but its dead half) takes entry V9_TPRE + k
            V9_LGKM(0);
            __builtin_amdgcn_s_barrier();                       // (the prefetched blocks are consumed: the ring is scratch now)
            auto chunk = [&](int k) { const int o = k * 4096; return o >= dead_a ? o + V9_HALF : o; };
            for (int it = wave; it < (nent - V9_TPRE) * 4; it += V9_NW) {
                const int i = V9_TPRE + (it >> 2), c = it & 3;
                const int kb = __builtin_amdgcn_readfirstlane(v8_entry(xb, wb, cx, i)[1]);
                const int koff = (kb >> 2) * 1024 + (kb & 3) * 256;
                const int vo = glane + c * 4 * (int)row_bytes, dl = ring_lds + chunk(it >> 2) + c * 1024;
                if (i < cx) { V9_BLDS16(vo, wg, koff, dl); } else { V9_BLDS16(vo, xg, koff, dl); }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            Ent t0, t1, t2;
            const int k0 = wave, k1 = wave + V9_NW, k2 = wave + 2 * V9_NW, nx = nent - V9_TPRE;
            load_entry(t0, V9_TPRE + k0, k0 < nx, chunk(k0), chunk(k0) + 1024, chunk(k0) + 2048, chunk(k0) + 3072);
            load_entry(t1, V9_TPRE + k1, k1 < nx, chunk(k1), chunk(k1) + 1024, chunk(k1) + 2048, chunk(k1) + 3072);
            load_entry(t2, V9_TPRE + k2, k2 < nx, chunk(k2), chunk(k2) + 1024, chunk(k2) + 2048, chunk(k2) + 3072);
            form_entry(t0, V9_TPRE + k0);
            form_entry(t1, V9_TPRE + k1);
            form_entry(t2, V9_TPRE + k2);
        }
        if (STAMP) st_y[1] = __builtin_amdgcn_s_memrealtime();
        // rows / columns with several entries (flagged by the bookkeeping): the others added to the first (wave = slot % 8,
        // ascending block) once every vector is complete
        if (reinterpret_cast<const int*>(smem + V9_FLAGS)[16] != 0) {       // (uniform)
            V9_LGKM(0);
            __builtin_amdgcn_s_barrier();
            auto follower_key = [&](int j) {
                if (j >= nent) return 0x7fffffff;
                const int* f = v8_entry(xb, wb, cx, j);
                const int s3 = f[3];
                return s3 >= 0 && s3 != j && (s3 & 7) == wave ? (s3 << 18) | (f[1] << 8) | j : 0x7fffffff;
            };
            mykeys0 = follower_key(lane);
            mykeys1 = follower_key(lane + 64);
            if (__any(mykeys0 != 0x7fffffff || mykeys1 != 0x7fffffff)) {
                int last = -1;
                for (;;) {
                    const int key = next_key(last, -1);
                    if (key == 0x7fffffff) break;
                    last = key;
                    float* h = V9_VEC(key >> 18);
                    const float* o = V9_VEC(key & 255);
#pragma unroll
                    for (int c = 0; c < 4; ++c) h[c * 64 + lane] += o[c * 64 + lane];
                }
            }
        }
        V9_LGKM(0);
        mykeys0 = mykeys1 = 0x7fffffff;                     // (the serial service below has nothing left to do)
    }
    int lastkey = -1, lastslot = -1;
    // the serial service (tiles whose entries do not fit the parallel one above; FIX_ 2 falling back)
    for (int phase = -1; phase < 0; ++phase) {
        bool more = look && !tmode;
        while (more) {
            int bkey[V9_NB_ENT];
#pragma unroll
            for (int b = 0; b < V9_NB_ENT; ++b) {
                bkey[b] = next_key(lastkey, phase);
                if (bkey[b] != 0x7fffffff) { lastkey = bkey[b]; gather(bkey[b] & 255, b); }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < V9_NB_ENT; ++b)
                if (bkey[b] != 0x7fffffff) {
                    const int slot = bkey[b] >> 18;
                    finish(bkey[b] & 255, b, slot, !tmode && slot >= V9_FAST_MAX && slot != lastslot);
                    lastslot = slot;
                }
            V9_LGKM(0);                                             // (this wave's reads of its scratch have returned)
            more = bkey[V9_NB_ENT - 1] != 0x7fffffff;
        }
    }
    if (STAMP) st_x[2] = __builtin_amdgcn_s_memrealtime();
    if (look) __builtin_amdgcn_s_barrier();
    if (STAMP) st_t[4] = __builtin_amdgcn_s_memrealtime();
#ifdef V9_GATED_TU
    // (4g) the gated epilogue: every fragment row's values as the plain epilogue forms them (scales, bias, the tile's correction
    //      vectors), then silu(gate) * up on the fragment pairs (2 jp, 2 jp + 1) -- the lane's four columns of both --, the
    //      consumer's quantiser over the pair's 16 h columns (= one [1,16] block of row l16: the four lanes l16 + 16 lq) and 8 bytes
    //      of bf16 a lane into the consumer's tiled operand.  Nothing is stored as fp32.
    if (!(FIX && mode == 3)) {
        int rs[8];
        float sxr[8];
        int4 cs[4];
        f32x4 swr[4], bvr[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = wm * 128 + i * 16 + l16;
            rs[i] = look ? rslot_r[rl] : -1;
            sxr[i] = sxt[rl];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cl = wn * 64 + j * 16 + lq * 4;
            cs[j] = look ? *reinterpret_cast<const int4*>(&cslot_r[cl]) : int4{-1, -1, -1, -1};
            swr[j] = *reinterpret_cast<const f32x4*>(&swt[cl]);
            bvr[j] = *reinterpret_cast<const f32x4*>(&bst[cl]);
        }
        const unsigned kpI = (unsigned)(Ni >> 6);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = wm * 128 + i * 16 + l16;
            const long long row = (long long)m0 + rl;
            f32x4 val[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) val[j][r] = (float)acc[i][j][r] * sxr[i] * swr[j][r] + bvr[j][r];
            }
            if (look) {
                const float* rv = V9_VEC(max(rs[i], 0)) + wn * 64 + lq * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(rv + j * 16);
                    const int cc[4] = {cs[j].x, cs[j].y, cs[j].z, cs[j].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        val[j][r] += rs[i] >= 0 ? c4[r] : 0.f;
                        val[j][r] += cc[r] >= 0 ? V9_VEC(max(cc[r], 0))[rl] : 0.f;
                    }
                }
            }
            if (a.epi_op == 2) {                                 // (uniform) relu: every fragment's 16 columns are a block of their own
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float h[4];
                    unsigned m = 0u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        h[r] = pre_relu(val[j][r]);
                        m = max(m, __float_as_uint(h[r]) & 0x7FFFFFFFu);
                    }
                    {
                        auto sw = __builtin_amdgcn_permlane32_swap(m, m, false, false);
                        m = max(sw[0], sw[1]);
                        sw = __builtin_amdgcn_permlane16_swap(m, m, false, false);
                        m = max(sw[0], sw[1]);
                    }
                    gated_quant<4>(h, m, glut, gq);
                    const int hc = n0 + wn * 64 + j * 16 + lq * 4;
                    if (row < a.M && hc < Ni)
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(a.yb) + gated_offset((unsigned)row, (unsigned)hc, (unsigned)(Ni >> 5))) =
                            make_uint2(pack_bf16(h[0], h[1]), pack_bf16(h[2], h[3]));
                }
                V9_SB();
                continue;
            }
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                float h[4];
                unsigned m = 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h[r] = pre_silu_mul(val[2 * jp][r], val[2 * jp + 1][r]);
                    m = max(m, __float_as_uint(h[r]) & 0x7FFFFFFFu);
                }
                {   // the block's four lanes (l16 + 16 lq) by lane swaps on the VALU (gfx950; no LDS crossbar round trip)
                    auto sw = __builtin_amdgcn_permlane32_swap(m, m, false, false);
                    m = max(sw[0], sw[1]);
                    sw = __builtin_amdgcn_permlane16_swap(m, m, false, false);
                    m = max(sw[0], sw[1]);
                }
                gated_quant<4>(h, m, glut, gq);
                const int hc = ((n0 + wn * 64) >> 1) + jp * 16 + lq * 4;
                if (row < a.M && 2 * hc < Ni)
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(a.yb) + gated_offset((unsigned)row, (unsigned)hc, kpI)) =
                        make_uint2(pack_bf16(h[0], h[1]), pack_bf16(h[2], h[3]));
            }
            V9_SB();
        }
        return;
    }
#endif
    // (4) the stores: fragment by fragment, first its tiles that no vector touches, then the others
    if (FIX && look && mode != 3) {
        // with corrections (the tile's own vectors, or the producers'): ONE pass, a row of four fragments at a time.  The wave's slots, scales and bias go to
        // registers first (the fragment registers of the K loop are free), so that the only LDS reads between a fragment's
        // accumulators and its store are the corrections themselves, four fragments' worth in flight together.  (The
        // two-pass form below, with its per-fragment look-ups, took 11-12 us against the plain epilogue's 7: stamps,
        // profiles/r04_v9_corr_stamps.txt; it remains for the tiles without any correction and for mode 3.)
        int rs[8];
        float sxr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = wm * 128 + i * 16 + l16;
            rs[i] = rslot_r[rl];
            sxr[i] = sxt[rl];
        }
        int4 cs[4];
        f32x4 swr[4], bvr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cl = wn * 64 + j * 16 + lq * 4;
            cs[j] = *reinterpret_cast<const int4*>(&cslot_r[cl]);
            swr[j] = *reinterpret_cast<const f32x4*>(&swt[cl]);
            bvr[j] = *reinterpret_cast<const f32x4*>(&bst[cl]);
        }
        // which fragments a vector touches at all (uniform masks, from the registers just loaded)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (__any(cs[j].x >= 0)) cmask |= 1u << (4 * j);
            if (__any(cs[j].y >= 0)) cmask |= 2u << (4 * j);
            if (__any(cs[j].z >= 0)) cmask |= 4u << (4 * j);
            if (__any(cs[j].w >= 0)) cmask |= 8u << (4 * j);
            if (cmask >> (4 * j) & 15) jmask |= 1u << j;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (__any(rs[i] >= 0)) rmask |= 1u << i;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool rowv = (rmask >> i) & 1;
            const int rl = wm * 128 + i * 16 + l16;
            const long long row = (long long)m0 + rl;
            float* yrow = a.y + row * a.ldy + n0 + wn * 64 + lq * 4;
            const int rs0 = max(rs[i], 0);
            const float* rv = V9_VEC(rs0) + wn * 64 + lq * 4;
            f32x4 val[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (MIXED) val[j][r] = accf[i][j][r] + bvr[j][r];
                    else val[j][r] = (float)acc[i][j][r] * sxr[i] * swr[j][r] + bvr[j][r];
                }
            }
            if (rowv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(rv + j * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[j][r] += rs[i] >= 0 ? c4[r] : 0.f;
                }
            }
            if (jmask) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c4[4] = {cs[j].x, cs[j].y, cs[j].z, cs[j].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (cmask & (1u << (4 * j + r)))
                            val[j][r] += c4[r] >= 0 ? V9_VEC(max(c4[r], 0))[rl] : 0.f;
                }
            }
            if (row < a.M) {
#ifdef V9_RESID_TU
                if (a.resid) {                                   // (uniform) the caller's residual add, in the store (see V9_RESID_TU)
                    const float* rr = a.resid + row * a.ldr + n0 + wn * 64 + lq * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = n0 + wn * 64 + j * 16 + lq * 4;
                        if (col + 3 < Ni) val[j] += *reinterpret_cast<const f32x4*>(rr + j * 16);
                        else
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (col + r < Ni) val[j][r] += rr[j * 16 + r];
                    }
                }
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + wn * 64 + j * 16 + lq * 4;
                    if (vec_ok && col + 3 < Ni) {
                        if (a.dbg & 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(yrow + j * 16), "v"(val[j]) : "memory");
                        else *reinterpret_cast<f32x4*>(yrow + j * 16) = val[j];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < Ni) yrow[j * 16 + r] = val[j][r];
                    }
                }
            }
            V9_SB();
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            store_rows(i, 0);
            if (look) store_rows(i, 1);
        }
    }
    if (STAMP) {
        st_t[5] = __builtin_amdgcn_s_memrealtime();
        if (a.stamps && (tid & 63) == 0 && (wave == 0 || wave == 7)) {
            unsigned long long* d = a.stamps + ((long long)blockIdx.x * 2 + (wave == 7)) * 8;
#pragma unroll
            for (int q = 0; q < 6; ++q) d[q] = st_t[q];
            d[6] = c_loop;
            // (bits 8-19 / 20-31: the parallel service's first round and its second round, x 10 ns)
            d[7] = ((unsigned long long)(unsigned)nent << 32) | (unsigned)mode | (((st_y[0] - st_x[1]) & 0xfff) << 8) | (((st_y[1] - st_y[0]) & 0xfff) << 20);
            // (the post-loop phases, x 10 ns, 12 bits each, instead of the stage-request stamp: bookkeeping, first-pass stores, vectors)
            d[1] = ((st_x[0] - st_t[3]) & 0xfff) | (((st_x[1] - st_x[0]) & 0xfff) << 12) | (((st_x[2] - st_x[1]) & 0xfff) << 24);
        }
    }
    if (FIX && mode == 3) {
        V9_WAITV(0);
        __syncthreads();
        v8_fix_atomic(a, xb, wb, cx, cw, sxt, swt, m0, n0, 256);
#ifdef V9_GATED_TU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __threadfence();
        __syncthreads();
        gated_post_tile(a, glut, gq, m0, n0, 256, 256, tid, V9_NT);
#endif
    }
}

#if !defined(V9_MIXED_TU) && !defined(V9_GATED_TU) && !defined(V9_RESID_TU)
static unsigned long long* g_v9_stamps = nullptr;       // diagnostic (tools/dbg/v9_stamps.py): where the stamps build writes
#endif

// 256 x 256 tiles, K % 128 == 0, at least four K-steps per slice (even slices under split-K).
#ifdef V9_GATED_TU
// the gated epilogue: a.wm / sw / wlist = the INTERLEAVED gate / up operand (a.N = 2 I rows), a.y = fp32 scratch [M, 2 I] (slow paths
// only), a.yb = the consumer's tiled bf16 operand [M, I], a.q_* = its quantiser
int launch_bfp_gemm_v9_gated(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist,
                             hipStream_t st, const uint8_t* xf, const uint8_t* wf) {
    if (!xlist || !wlist || !xf || !wf || !a.yb || !a.y) return MI355Q_E_BADARG;
    if (a.K % 128 != 0 || a.K < 256 || a.N % (a.epi_op == 2 ? 32 : 64) != 0 || a.splits > 1 || a.ngroup > 1 || a.x_post || (a.epi_op != 1 && a.epi_op != 2))
        return MI355Q_E_UNSUPPORTED;
    const unsigned grid = (unsigned)((a.M + 255) / 256 * ((a.N + 255) / 256));
    hipLaunchKernelGGL((bfp_gemm_v9<1, false, false>), grid, V9_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    return (int)hipGetLastError();
}
#elif defined(V9_MIXED_TU)
// the mixed contraction (MIXED above): a.K / a.xm / a.wm = class 0 (row-aligned int8, K % 128 == 0), a.K1 / a.xm1 / a.wm1 = class 1
// (tiled bf16, K1 % 64 == 0, K1 >= 128)
int launch_bfp_gemm_v9_mixed(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist,
                             hipStream_t st, const uint8_t* xf, const uint8_t* wf) {
    if (!xlist || !wlist || !xf || !wf || !a.xm1 || !a.wm1) return MI355Q_E_BADARG;
    if (a.K % 128 != 0 || a.K < 256 || a.K1 % 64 != 0 || a.K1 < 128 || a.splits > 1 || a.ngroup > 1 || a.x_post) return MI355Q_E_UNSUPPORTED;
    const unsigned grid = (unsigned)((a.M + 255) / 256 * ((a.N + 255) / 256));
    hipLaunchKernelGGL((bfp_gemm_v9<1, false, false, true>), grid, V9_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    return (int)hipGetLastError();
}

#elif defined(V9_RESID_TU)
// the row-scale int8 product with exception lists and a.resid set: the shipping kernel + the residual add in its one-pass stores
int launch_bfp_gemm_v9_resid(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist,
                             hipStream_t st, const uint8_t* xf, const uint8_t* wf) {
    if (!xlist || !wlist || !xf || !wf || !a.resid) return MI355Q_E_BADARG;
    const unsigned tiles = (unsigned)((a.M + 255) / 256 * ((a.N + 255) / 256));
    const unsigned grid = tiles * (a.ngroup > 1 ? a.ngroup : 1) * (a.splits > 1 ? a.splits : 1);
    hipLaunchKernelGGL((bfp_gemm_v9<1, false, false>), grid, V9_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    return (int)hipGetLastError();
}
#else
int launch_bfp_gemm_v9(const GemmArgs& a_in, const float* sx, const float* sw, const int* xlist, const int* wlist,
                       hipStream_t st, const uint8_t* xf, const uint8_t* wf, bool bf16) {
    if (a_in.resid && !bf16 && xlist && wlist) return launch_bfp_gemm_v9_resid(a_in, sx, sw, xlist, wlist, st, xf, wf);
    GemmArgs a = a_in;
    // (stamps build: while a buffer is registered through mi355q_debug_v9_stamps -- bench.py's `roofline.loop_clock_GHz`,
    //  tools/dbg/v9_stamps.py; no stamp executes in the kernel every other launch runs)
    const bool want_stamps = g_v9_stamps != nullptr;
    if (want_stamps) a.stamps = g_v9_stamps;
    const bool fix = xlist && wlist;
    if (fix && (!xf || !wf)) return MI355Q_E_BADARG;
    const unsigned tiles = (unsigned)((a.M + 255) / 256 * ((a.N + 255) / 256));
    const unsigned grid = tiles * (a.ngroup > 1 ? a.ngroup : 1) * (a.splits > 1 ? a.splits : 1);
    if (bf16) hipLaunchKernelGGL((bfp_gemm_v9<0, true, false>), grid, V9_NT, 0, st, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    else if (!fix) hipLaunchKernelGGL((bfp_gemm_v9<0, false, false>), grid, V9_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else if (want_stamps && a.stamps) hipLaunchKernelGGL((bfp_gemm_v9<1, false, true>), grid, V9_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else if (g_kernel_events.start) {
        // (benchmark timing: the events carry this dispatch's own start and end -- mi355q_internal.h)
        hipExtLaunchKernelGGL((bfp_gemm_v9<1, false, false>), dim3(grid), dim3(V9_NT), 0, st, g_kernel_events.start, g_kernel_events.stop, 0, a, sx, sw,
                              xlist, wlist, xf, wf);
        g_kernel_events.start = nullptr;
    } else hipLaunchKernelGGL((bfp_gemm_v9<1, false, false>), grid, V9_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    return (int)hipGetLastError();
}
#endif

}  // namespace mi355q

#if !defined(V9_MIXED_TU) && !defined(V9_GATED_TU) && !defined(V9_RESID_TU)
// diagnostic hook, not part of include/mi355q.h: the buffer ([workgroups][2][8] 64-bit words) the MI355Q_V9_STAMPS build fills
extern "C" __attribute__((visibility("default"))) void mi355q_debug_v9_stamps(void* buf) { mi355q::g_v9_stamps = static_cast<unsigned long long*>(buf); }
#endif
