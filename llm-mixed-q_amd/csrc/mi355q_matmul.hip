// mi355q_matmul.hip -- block-quantised batched matmul  out[b] = Qx(x[b]) @ Qy(y[b])  for block_fp, block_minifloat and (x
// only: the reference leaves y unquantised there, matmul.py:252-297) block_log operands  (reference
// quantized_functions/matmul.py:146-297: x quantised along its last dim = the contraction, y along ITS last dim = the
// output columns, then torch.matmul / torch.bmm on the fake-quantised fp32 tensors).
//
// The large operand is x (attention probabilities [heads, T, T], 4 B per element): the reference writes its
// fake-quantised copy and reads it again for the product.  Here it is read ONCE:
//   kernel 1 (small operand)  y fp32 [B, K, N] -> fake-quantise along N ([1,16] blocks) -> yt bf16 in FRAGMENT ORDER
//                             (what an MFMA operand of kernel 2 holds, one contiguous KiB per fragment load);
//   kernel 2                  one workgroup per 16 rows of x: lanes load float4s of their rows (one instruction = 64
//                             contiguous bytes of every row), complete the block maxima with two shuffles, quantise in
//                             registers -- the same decisions as the streaming quantiser, bit for bit -- convert the
//                             results to bf16 (exact: a block_fp value of width <= 9 has <= 8 significant bits) and
//                             feed v_mfma_f32_16x16x32_bf16 against the yt fragments.
// The K index inside a 64-step is permuted (mm_kperm); both operands use the same permutation, so the sum is the same.
// Products of two such values are exact in fp32; accumulation is fp32 like the reference's GEMM (order differs:
// tolerance of the matmul tests, 1e-3).  All-zero blocks quantise to zeros whatever their exponent (block_fp.py:54-58
// only changes the stored code).  Elements |x| <= 1e-8, which the reference passes through unquantised, enter the
// product rounded to bf16 (<= 2e-11 absolute each).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_quant_dev.h"

namespace mi355q {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int MM_NCHUNK = 64;            // output columns per accumulator set (4 MFMA tiles)

// position of contraction index k (inside its 64-group) in yt and in the MFMA operands: lane group g of kernel 2 holds
// elements 4 g .. 4 g + 3 of each of the four [1,16] blocks (so that one load instruction reads 64 contiguous bytes of
// every row); MFMA t takes blocks 2 t and 2 t + 1.  k = 16 (2 t + h) + 4 g + e  ->  32 t + 8 g + 4 h + e.
__host__ __device__ constexpr int mm_kperm(int k) {
    return ((k >> 5) & 1) * 32 + ((k >> 2) & 3) * 8 + ((k >> 4) & 1) * 4 + (k & 3);
}
// the tile product kernel's (kernel 3): lane group g holds the WHOLE block g -- the block maximum and its parameter need no
// cross-lane traffic -- and MFMA t takes its elements 8 t .. 8 t + 7.  k = 16 g + 8 t + e  ->  32 t + 8 g + e.
__host__ __device__ constexpr int mm_kperm_lane(int k) {
    return ((k >> 3) & 1) * 32 + ((k >> 4) & 3) * 8 + (k & 7);
}

// ---- kernel 1: y [B, K, N] fp32 -> fake-quantise along N -> yt [B, N, K] bf16 ------------------------------------
// FMT: the block format y is fake-quantised in (round 4: block_minifloat beside block_fp -- its values have <= 7 mantissa bits,
// exact in bf16 like block_fp's).  FMT_RAW: y is NOT quantised (block_log products, reference quirk: matmul.py:252-297 passes
// the second operand through) and goes out as THREE bf16 planes hi / mid / lo with y = hi + mid + lo exactly (8 + 8 + 8
// significant bits by truncation): x is a signed power of two there, so every x * plane product is exact in fp32 and the sum
// of the three accumulations is the fp32 product's, up to summation order.  `plane_stride`: elements between planes.
constexpr int FMT_RAW = 3;
template <int FMT, bool LANE = false>                      // LANE: the K order of the tile product kernel (mm_kperm_lane)
__global__ __launch_bounds__(256) void bfp_quant_pack_t_kernel(const QuantArgs a, const float* __restrict__ y,
                                                               uint16_t* __restrict__ yt, long long K, long long Kp, long long N,
                                                               long long plane_stride) {
    __shared__ Lut lut;
    constexpr int NPL = FMT == FMT_RAW ? 3 : 1;
    __shared__ uint16_t tile[NPL][64][64 + 8];            // [plane][n][k], row padded against bank conflicts
    if (FMT != FMT_RAW) load_lut<(FMT == FMT_RAW ? FMT_BFP : FMT)>(lut);
    const int tid = threadIdx.x;
    const long long b = blockIdx.z, k0 = (long long)blockIdx.y * 64, n0 = (long long)blockIdx.x * 64;
    const int n4 = tid & 15, kr = tid >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long k = k0 + kr + 16 * i, n = n0 + n4 * 4;
        const bool ok = k < K && n < N;                   // (N % 16 == 0: a float4 is inside or outside as a whole)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(y + (b * K + k) * N + n);
        float bmax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        bmax = group_max<4>(bmax);                        // 4 adjacent lanes = one [1,16] block along N
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        if (FMT == FMT_RAW) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = vv[j];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {         // truncation: every plane exact, the remainder exact
                    const unsigned hb = __float_as_uint(r) & 0xFFFF0000u;
                    tile[pl][n4 * 4 + j][LANE ? mm_kperm_lane(kr + 16 * i) : mm_kperm(kr + 16 * i)] = (uint16_t)(hb >> 16);
                    r -= __uint_as_float(hb);
                }
            }
        } else {
            constexpr int F = FMT == FMT_RAW ? FMT_BFP : FMT;
            if (bmax != 0.f) {
                unsigned code;
                const BlockParam bp = block_param<F>(bmax, a, lut, code);
                int mant;
                q[0] = quant_elem<F>(v.x, bp, a, lut, mant);
                q[1] = quant_elem<F>(v.y, bp, a, lut, mant);
                q[2] = quant_elem<F>(v.z, bp, a, lut, mant);
                q[3] = quant_elem<F>(v.w, bp, a, lut, mant);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[0][n4 * 4 + j][LANE ? mm_kperm_lane(kr + 16 * i) : mm_kperm(kr + 16 * i)] = (uint16_t)(pack_bf16(q[j], 0.f) & 0xFFFFu);
        }
    }
    __syncthreads();
    // 64 rows (n) x 64 k: thread writes 16 bytes (8 k) of one row
    // FRAGMENT ORDER: the 16 bytes lane (c, g) of kernel 2 takes for (16-column tile, 64-step, MFMA t) lie at lane * 16
    // inside a 1-KiB piece, pieces ordered [tile][step][t]: a wave's fragment load is one contiguous KiB (8 cache lines
    // instead of 16 half lines -- the fragment loads are L1-access bound).  Kp = number of 64-steps here.
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = tid + 256 * i, ntl = slot >> 7, t = (slot >> 6) & 1, ln = slot & 63;
        if (n0 + 16 * ntl < N) {                          // (whole 64-groups: zeros behind K)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                *reinterpret_cast<uint4*>(yt + pl * plane_stride + ((((b * (N >> 4) + (n0 >> 4) + ntl) * Kp + blockIdx.y) * 2 + t) * 64 + ln) * 8) =
                    *reinterpret_cast<const uint4*>(&tile[pl][16 * ntl + (ln & 15)][32 * t + 8 * (ln >> 4)]);
        }
    }
}

// ---- kernel 2 -----------------------------------------------------------------------------------------------------
// Workgroup = 16 rows of x, 4 waves.  A wave step covers 64 contraction elements = four [1,16] blocks of each row: lane
// (r = lane % 16, g = lane / 16) loads float4 number g of every block (one load instruction = 64 contiguous bytes of
// each of the 16 rows), block maxima are completed across the four lane groups with two shuffles, every lane quantises
// its 16 values; MFMA t takes the lane's values of blocks 2 t and 2 t + 1 (mm_kperm).  The roles of the MFMA operands
// are swapped (yt fragment as A, x fragment as B) so that a lane ends up with FOUR CONSECUTIVE columns of one output
// row: 16-byte stores.  Long contractions (probs x V) are SPLIT over the waves (each streams a quarter of K; partial
// tiles are summed through LDS in wave order); short ones (Q x K^T) keep the quantised row block in registers and
// split the column chunks instead.
struct XBlk { float4 v[4]; };

// (loads are UNCONDITIONAL -- a block behind K is read from the row's first block and zeroed afterwards: a load inside a
// branch makes the compiler's vmcnt bookkeeping drain everything in flight, the prefetch included)
__device__ __forceinline__ void load_xblk(XBlk& s, const float* __restrict__ row, long long k0, int g, long long K) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                         // (K % 16 == 0: a block is inside or outside as a whole)
        const long long k = k0 + 16 * i;
        s.v[i] = *reinterpret_cast<const float4*>(row + (k < K ? k : 0) + 4 * g);
    }
}
__device__ __forceinline__ void mask_xblk(XBlk& s, long long k0, long long K) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (k0 + 16 * i >= K) s.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// (quant_elem_fused / bm_elem_fused / bl_elem_fused: mi355q_quant_dev.h -- the streaming quantisers use them too since round 5)

// quantise the wave's step: lane (r, g) holds float4 g of blocks 0..3 of row r.  The shared exponent of block i is
// worked out by lane group i only (threshold lookup and clamp once per block, not four times) and fetched by the others.
template <int FMT>
__device__ __forceinline__ void quantise_xblk(const XBlk& s, const QuantArgs& a, const Lut& lut, int mbits, int lane,
                                              bf16x8 (&afr)[2], float zfill = 1.0f) {
    const int g = lane >> 4;
    float bm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = s.v[i];
        float bmax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        bmax = fmaxf(bmax, __shfl_xor(bmax, 16));         // the block's other three float4s live in lanes ^16, ^32, ^48
        bm[i] = fmaxf(bmax, __shfl_xor(bmax, 32));
    }
    const float mine = g == 0 ? bm[0] : (g == 1 ? bm[1] : (g == 2 ? bm[2] : bm[3]));
    unsigned code;
    // (an all-zero block: block_fp / block_minifloat elements <= 1e-8 pass through as zeros whatever the parameter; block_log
    //  has no pass-through -- its zeros become +2^-bias with the bias of the reference's tensor-wide fill, the smallest
    //  non-zero block maximum of the whole tensor (block_fp.py:54-58 via block_log.py:48-58): `zfill`, from the statistics pass)
    const int pmine = block_param<FMT>(mine > 0.f ? mine : zfill, a, lut, code).p;
    float q[16];
    bool near = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = __shfl(pmine, (lane & 15) + 16 * i);
        const float4 v = s.v[i];
        if (FMT == FMT_BFP) {
            // (elements of an all-zero block are <= 1e-8 and pass through as zeros)
            q[4 * i + 0] = quant_elem_fused(v.x, mbits - p, p - mbits, a.mant_max);
            q[4 * i + 1] = quant_elem_fused(v.y, mbits - p, p - mbits, a.mant_max);
            q[4 * i + 2] = quant_elem_fused(v.z, mbits - p, p - mbits, a.mant_max);
            q[4 * i + 3] = quant_elem_fused(v.w, mbits - p, p - mbits, a.mant_max);
        } else if (FMT == FMT_BM) {
            q[4 * i + 0] = bm_elem_fused(v.x, 127 - p, 127 + a.span - p, mbits, a.shift, a.mant_max, near);
            q[4 * i + 1] = bm_elem_fused(v.y, 127 - p, 127 + a.span - p, mbits, a.shift, a.mant_max, near);
            q[4 * i + 2] = bm_elem_fused(v.z, 127 - p, 127 + a.span - p, mbits, a.shift, a.mant_max, near);
            q[4 * i + 3] = bm_elem_fused(v.w, 127 - p, 127 + a.span - p, mbits, a.shift, a.mant_max, near);
        } else {
            const float eps = __builtin_ldexpf(0.1f, -p);
            q[4 * i + 0] = bl_elem_fused(v.x, eps, 127 - p, 127 + a.span - p, near);
            q[4 * i + 1] = bl_elem_fused(v.y, eps, 127 - p, 127 + a.span - p, near);
            q[4 * i + 2] = bl_elem_fused(v.z, eps, 127 - p, 127 + a.span - p, near);
            q[4 * i + 3] = bl_elem_fused(v.w, eps, 127 - p, 127 + a.span - p, near);
        }
    }
    if (FMT != FMT_BFP && __any(near)) {                  // (rare, wave-uniform: the streaming quantisers' element functions)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            BlockParam bp;
            bp.p = __shfl(pmine, (lane & 15) + 16 * i);
            bp.eps = FMT == FMT_BL ? __builtin_ldexpf(0.1f, -bp.p) : 0.f;
            const float4 v = s.v[i];
            int mant;
            constexpr int F = FMT == FMT_BFP ? FMT_BM : FMT;
            q[4 * i + 0] = quant_elem<F>(v.x, bp, a, lut, mant);
            q[4 * i + 1] = quant_elem<F>(v.y, bp, a, lut, mant);
            q[4 * i + 2] = quant_elem<F>(v.z, bp, a, lut, mant);
            q[4 * i + 3] = quant_elem<F>(v.w, bp, a, lut, mant);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        uint4 pk;
        pk.x = pack_bf16(q[8 * t + 0], q[8 * t + 1]);
        pk.y = pack_bf16(q[8 * t + 2], q[8 * t + 3]);
        pk.z = pack_bf16(q[8 * t + 4], q[8 * t + 5]);
        pk.w = pack_bf16(q[8 * t + 6], q[8 * t + 7]);
        afr[t] = __builtin_bit_cast(bf16x8, pk);
    }
}

// The same for kernel 3's layout: the lane holds the 16 values of ONE block (v[0..3] = its four float4s).  Block maximum and
// parameter in the lane -- no shuffles, no LDS round trips (kernel 3 runs two waves a SIMD: nothing would hide them); the log2
// tables are consulted only where they can matter (a maximum within 45 ulps above / 88 below a power of two, a subnormal one:
// wave-uniform, rare), the elements go through the fused element functions above with the same rare way out.
template <int FMT>
__device__ __forceinline__ void quantise_lane(const f32x4 (&v)[4], const QuantArgs& a, const Lut& lut, int mbits, float zfill,
                                              bf16x8 (&afr)[2]) {
    float bmax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) bmax = fmaxf(fmaxf(bmax, fmaxf(fabsf(v[i][0]), fabsf(v[i][1]))), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
    const float bm = bmax > 0.f ? bmax : zfill;           // (an all-zero block: see quantise_xblk)
    const unsigned bb = __float_as_uint(bm), m = bb & 0x7FFFFFu;
    const int k = (int)(bb >> 23) - 127;
    int p;
    bool near = bb < 0x00800000u;
    if (FMT == FMT_BFP) {
        near |= (m - 1u) < MI355Q_LOG2_CEIL_THR_MAX;
        p = clampi(k + (m != 0u ? 1 : 0), a.e_min, a.e_max);
    } else if (FMT == FMT_BM) {
        near |= m >= MI355Q_LOG2_FLOOR_THR_MIN;
        p = clampi(k, 0, a.bias_max);
    } else {
        near |= (m - 1u) < MI355Q_LOG2_CEIL_THR_MAX;
        p = clampi(a.span - (k + (m != 0u ? 1 : 0)), 0, a.bias_max);
    }
    if (__any(near)) {
        unsigned code;
        p = block_param<FMT>(bm, a, lut, code).p;
    }
    float q[16];
    bool near_e = false;
    const float eps = FMT == FMT_BL ? __builtin_ldexpf(0.1f, -p) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float e4[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (FMT == FMT_BFP) q[4 * i + j] = quant_elem_fused(e4[j], mbits - p, p - mbits, a.mant_max);
            else if (FMT == FMT_BM) q[4 * i + j] = bm_elem_fused(e4[j], 127 - p, 127 + a.span - p, mbits, a.shift, a.mant_max, near_e);
            else q[4 * i + j] = bl_elem_fused(e4[j], eps, 127 - p, 127 + a.span - p, near_e);
        }
    }
    if (FMT != FMT_BFP && __any(near_e)) {                // (rare, wave-uniform: the streaming quantisers' element functions)
        BlockParam bp;
        bp.p = p;
        bp.eps = eps;
        constexpr int F = FMT == FMT_BFP ? FMT_BM : FMT;
        int mant;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q[4 * i + 0] = quant_elem<F>(v[i][0], bp, a, lut, mant);
            q[4 * i + 1] = quant_elem<F>(v[i][1], bp, a, lut, mant);
            q[4 * i + 2] = quant_elem<F>(v[i][2], bp, a, lut, mant);
            q[4 * i + 3] = quant_elem<F>(v[i][3], bp, a, lut, mant);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        uint4 pk;
        pk.x = pack_bf16(q[8 * t + 0], q[8 * t + 1]);
        pk.y = pack_bf16(q[8 * t + 2], q[8 * t + 3]);
        pk.z = pack_bf16(q[8 * t + 4], q[8 * t + 5]);
        pk.w = pack_bf16(q[8 * t + 6], q[8 * t + 7]);
        afr[t] = __builtin_bit_cast(bf16x8, pk);
    }
}

// yt fragments of one step for NT column tiles (straight from L1 / L2; positions inside a 64-group are permuted and the
// group is whole in yt), and their use: acc[tile] += Y(step, columns n0 + 16 tile ...) * X(step)^T, i.e.
// acc[tile][i] = out[row lane % 16][column 4 (lane / 16) + i].  Requested BEFORE the next step's x: vmcnt counts in
// order, so fragments requested behind the x prefetch could only be consumed once that whole HBM round trip is back.
template <int NT> struct BFrag { uint4 v[NT][2]; };
template <int NT>
__device__ __forceinline__ void load_bfrag(BFrag<NT>& bf, const uint16_t* __restrict__ ytb, long long n0, long long k0, long long K,
                                           long long N, int lane) {
#pragma unroll
    for (int tile = 0; tile < NT; ++tile) {
        const long long nt = min((n0 >> 4) + tile, (N >> 4) - 1);
#pragma unroll
        for (int t = 0; t < 2; ++t)                       // (K = number of 64-steps; k0 = 64 * step)
            bf.v[tile][t] = *reinterpret_cast<const uint4*>(ytb + (((nt * K + (k0 >> 6)) * 2 + t) * 64 + lane) * 8);
    }
}
template <int NT>
__device__ __forceinline__ void mma_step(const bf16x8 (&afr)[2], const BFrag<NT>& bf, long long n0, long long N, f32x4 (&acc)[NT]) {
#pragma unroll
    for (int tile = 0; tile < NT; ++tile) {
        if (n0 + 16 * tile >= N) break;                   // (uniform; N % 16 == 0)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            acc[tile] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf.v[tile][t]), afr[t], acc[tile], 0, 0, 0);
    }
}

__device__ __forceinline__ void store_tile(const f32x4& acc, float* __restrict__ outb, long long m0, long long n, long long M,
                                           long long N, int lane) {
    const long long m = m0 + (lane & 15);
    if (m < M) *reinterpret_cast<float4*>(outb + m * N + n + 4 * (lane >> 4)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}


// (two instantiations: the short-contraction one keeps quantised x in registers, the streaming one stays lean)
// SOFTMAX (streaming instantiations only): x holds attention SCORES; the rows' softmax (fp32: exp(x - max) / sum, the
// arithmetic of torch.softmax, exponentials to ~1 ulp) is formed on the way in, so the probability tensor [heads, T, T] never exists in memory:
// two statistics passes over the workgroup's 16 rows (max, then the sum of exponentials; the rows come from L2 the second
// and third time), then the main pass turns every score into its probability right before the block quantiser.
// exp(x) for x <= 0 to ~1 ulp in 6 operations: 2^(x log2e) with the product's rounding error carried along
// (t = fl(x L), r = x L - t exactly by FMA, plus x times the low part of log2e;  2^(t + r) = 2^t (1 + r ln2 + ...)).
__device__ __forceinline__ float exp_neg(float x) {
    x = fmaxf(x, -104.0f);               // (masked scores are finfo.min: the scaled argument would overflow; e^-104 = 0 in fp32)
    constexpr float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f, LN2 = 0.693147182464599609375f;
    const float t = x * L2E_HI;
    float r = __builtin_fmaf(x, L2E_HI, -t);
    r = __builtin_fmaf(x, L2E_LO, r);
    const float p = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(p, r * LN2, p);
}
// e / l with the quotient corrected once (q = e inv; q += (e - q l) inv): the correctly rounded quotient except for rare
// double-rounding cases, in 3 operations instead of the division's ~10
__device__ __forceinline__ float div_fast(float e, float l, float inv) {
    const float q = e * inv;
    return __builtin_fmaf(__builtin_fmaf(-q, l, e), inv, q);
}
__device__ __forceinline__ void softmax_xblk(XBlk& s, float m, float l, float inv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s.v[i].x = div_fast(exp_neg(s.v[i].x - m), l, inv);
        s.v[i].y = div_fast(exp_neg(s.v[i].y - m), l, inv);
        s.v[i].z = div_fast(exp_neg(s.v[i].z - m), l, inv);
        s.v[i].w = div_fast(exp_neg(s.v[i].w - m), l, inv);
    }
}

// scores as the reference's attention hands them to softmax (modeling_opt.py:262-276, modeling_llama.py:318-329):
// max(x + mask, finfo.min).  `mrow`: the additive mask's row [K] for this lane's query (nullable); causal: keys behind
// `kvis` are masked (what the causal mask's finfo.min entries do: x + finfo.min clamps to finfo.min for every finite x).
__device__ __forceinline__ void mask_scores(XBlk& s, const float* __restrict__ mrow, long long k0, int g, long long K, long long kvis) {
    constexpr float FMIN = -3.4028234663852886e38f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long k = k0 + 16 * i + 4 * g;
        if (mrow) {
            const float4 m = *reinterpret_cast<const float4*>(mrow + (k0 + 16 * i < K ? k : 4 * g));
            s.v[i].x = fmaxf(s.v[i].x + m.x, FMIN); s.v[i].y = fmaxf(s.v[i].y + m.y, FMIN);
            s.v[i].z = fmaxf(s.v[i].z + m.z, FMIN); s.v[i].w = fmaxf(s.v[i].w + m.w, FMIN);
        }
        if (k + 0 > kvis) s.v[i].x = FMIN;
        if (k + 1 > kvis) s.v[i].y = FMIN;
        if (k + 2 > kvis) s.v[i].z = FMIN;
        if (k + 3 > kvis) s.v[i].w = FMIN;
    }
}

// block_log's statistics over x (bl_block_stats_kernel below): BL_STATS_SLOTS words, one per workgroup of that pass -- the smallest
// non-zero [1,16]-block maximum it saw, as its bit pattern, all ones if none.  The product kernels take the minimum themselves (4 KiB
// from the L2 per workgroup): no atomics on one word, no memset launch in front, and the pass can run ahead of the y pack.
constexpr int BL_STATS_SLOTS = 1024;
template <int NTHREADS>
__device__ __forceinline__ float bl_zero_fill(const unsigned* __restrict__ stats) {
    __shared__ unsigned part[NTHREADS / 64];
    unsigned best = 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < BL_STATS_SLOTS; i += NTHREADS) best = min(best, stats[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    best = part[0];
#pragma unroll
    for (int w = 1; w < NTHREADS / 64; ++w) best = min(best, part[w]);
    const unsigned u = __builtin_amdgcn_readfirstlane(best);
    return u == 0xFFFFFFFFu ? 1.0f : __uint_as_float(u);       // (every block zero: the reference's fill is 1)
}

// FMT: x's block format (block_fp, block_minifloat, block_log); PLANES: bf16 planes of y (3 for block_log's raw y, else 1)
template <bool RESIDENT, int NT, bool SOFTMAX = false, int FMT = FMT_BFP, int PLANES = 1>
__global__ __launch_bounds__(256) void bfp_qmatmul_kernel(const QuantArgs a, const float* __restrict__ x,
                                                          const uint16_t* __restrict__ yt, float* __restrict__ out,
                                                          long long M, long long K, long long Kp, long long N,
                                                          const float* __restrict__ mask, long long causal_off,
                                                          long long plane_stride, const unsigned* __restrict__ xstats) {
    __shared__ Lut lut;
    __shared__ f32x4 red[4][NT][64];                      // [wave][tile][lane]: split-K partial tiles
    __shared__ float stat[4][16];
    load_lut<FMT>(lut);
    float zfill = 1.0f;                                   // (block_log: the fill of all-zero blocks, bl_block_stats_kernel)
    if (FMT == FMT_BL) zfill = bl_zero_fill<256>(xstats);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    // (causal softmax: the grid is (heads, row blocks) and the row blocks run from the last one down -- the row blocks with the most
    //  visible keys of EVERY head are dispatched first and the launch ends on its cheapest workgroups; round 5, see mi355q_attention.hip)
    const bool heavy_first = SOFTMAX && causal_off >= 0;
    const long long b = heavy_first ? blockIdx.x : blockIdx.y;
    const long long m0 = (heavy_first ? (long long)(gridDim.y - 1 - blockIdx.y) : (long long)blockIdx.x) * 16;
    const long long mrow = min(m0 + (lane & 15), M - 1);                 // (rows past M: loaded again, never stored)
    const float* __restrict__ row = x + (b * M + mrow) * K;
    const uint16_t* __restrict__ ytb = yt + b * (N >> 4) * Kp * 1024;   // (Kp = 64-steps per row of tiles)
    float* __restrict__ outb = out + b * M * N;
    long long nsteps = (K + 63) / 64;                      // (= Kp: yt is stored in fragment order)
    const int mbits = (int)__builtin_log2f(a.shift);
    bf16x8 afr[2];
    if (RESIDENT) {                      // short contraction (Q K^T): quantise the row block once, waves share the columns
        // (NT = the number of 64-steps here, 1..3: everything below is branch-free so that the counted waits stay exact)
        constexpr int NS = NT;
        bf16x8 res[NS][2];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            XBlk s;
            load_xblk(s, row, st * 64, g, K);
            mask_xblk(s, st * 64, K);
            quantise_xblk<FMT>(s, a, lut, mbits, lane, res[st], zfill);
        }
        // Two fragment sets used alternately: the next chunk's yt fragments are requested BEFORE this chunk's stores --
        // vmcnt counts loads and stores in one queue, so fragments requested behind the stores could only be used once
        // those writes were acknowledged.
        BFrag<4> b0[NS], b1[NS];
        const long long step = 4 * MM_NCHUNK, nlast = ((N - 1) / MM_NCHUNK) * MM_NCHUNK;
        long long n0 = (long long)wave * MM_NCHUNK;
        // (PLANES > 1: the planes of a chunk are taken one after the other through the same two fragment sets -- the
        //  sequence (chunk, plane) alternates between them -- into one accumulator that is stored behind the last plane)
        int pl = 0;
        f32x4 acc[4];
#pragma unroll
        for (int st = 0; st < NS; ++st) load_bfrag<4>(b0[st], ytb, min(n0, nlast), st * 64, Kp, N, lane);
#define MI355Q_MM_CHUNK(USE_, FILL_)                                                                         \
        {                                                                                                   \
            if (PLANES == 1 || pl == 0)                                                                     \
                _Pragma("unroll") for (int tile = 0; tile < 4; ++tile) acc[tile] = f32x4{0, 0, 0, 0};       \
            _Pragma("unroll") for (int st = 0; st < NS; ++st) mma_step<4>(res[st], USE_[st], n0, N, acc);   \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            const bool last_pl = PLANES == 1 || pl == PLANES - 1;                                           \
            const long long nn = last_pl ? n0 + step : n0;                                                  \
            const int npl = last_pl ? 0 : pl + 1;                                                           \
            _Pragma("unroll") for (int st = 0; st < NS; ++st)                                               \
                load_bfrag<4>(FILL_[st], ytb + npl * plane_stride, min(nn, nlast), st * 64, Kp, N, lane);   \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            if (last_pl) {                                                                                  \
                _Pragma("unroll") for (int tile = 0; tile < 4; ++tile)                                      \
                    if (n0 + 16 * tile < N) store_tile(acc[tile], outb, m0, n0 + 16 * tile, M, N, lane);   \
            }                                                                                               \
            n0 = nn;                                                                                        \
            pl = npl;                                                                                       \
        }
        while (n0 < N) {
            MI355Q_MM_CHUNK(b0, b1)
            if (n0 >= N) break;
            MI355Q_MM_CHUNK(b1, b0)
        }
#undef MI355Q_MM_CHUNK
        return;
    }
    float row_max = 0.f, row_sum = 1.f;
    // (SOFTMAX) causal: this lane's query sees keys 0 .. kvis; the workgroup's 16 rows need no step behind the last
    // row's horizon (their probabilities are exactly 0 there: all-zero blocks quantise to zeros)
    const long long kvis = SOFTMAX && causal_off >= 0 ? mrow + causal_off : K;
    const long long nsteps_all = nsteps;
    const float* __restrict__ mrowp = SOFTMAX && mask ? mask + mrow * K : nullptr;
    if (SOFTMAX && causal_off >= 0) {
        const long long last = min(m0 + 15, M - 1) + causal_off;       // horizon of the workgroup's last row
        const long long need = last / 64 + 1;
        if (need < nsteps) nsteps = need < 1 ? 1 : need;
    }
    (void)nsteps_all;
    if (SOFTMAX) {
        // statistics of this lane's row r = lane % 16: every (wave, lane group g) covers a quarter of a quarter of the
        // steps' values; combine over g by shuffles, over the waves through LDS.  Values behind K do not count.
        float mx = -INFINITY;
        for (long long st = wave; st < nsteps; st += 4) {
            XBlk s;
            load_xblk(s, row, st * 64, g, K);
            mask_scores(s, mrowp, st * 64, g, K, kvis);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (st * 64 + 16 * i < K) mx = fmaxf(mx, fmaxf(fmaxf(s.v[i].x, s.v[i].y), fmaxf(s.v[i].z, s.v[i].w)));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (g == 0) stat[wave][lane & 15] = mx;
        __syncthreads();
        row_max = fmaxf(fmaxf(stat[0][lane & 15], stat[1][lane & 15]), fmaxf(stat[2][lane & 15], stat[3][lane & 15]));
        __syncthreads();
        float sm = 0.f;
        for (long long st = wave; st < nsteps; st += 4) {
            XBlk s;
            load_xblk(s, row, st * 64, g, K);
            mask_scores(s, mrowp, st * 64, g, K, kvis);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (st * 64 + 16 * i < K)
                    sm += (exp_neg(s.v[i].x - row_max) + exp_neg(s.v[i].y - row_max)) + (exp_neg(s.v[i].z - row_max) + exp_neg(s.v[i].w - row_max));
        }
        sm += __shfl_xor(sm, 16);
        sm += __shfl_xor(sm, 32);
        if (g == 0) stat[wave][lane & 15] = sm;
        __syncthreads();
        row_sum = (stat[0][lane & 15] + stat[1][lane & 15]) + (stat[2][lane & 15] + stat[3][lane & 15]);
    }
    const float row_inv = 1.0f / row_sum;
    // long contraction: wave w streams steps w, w + 4, w + 8, ... of every column chunk (the four waves read adjacent
    // 256-byte pieces of each row: 1 KiB runs per row and DRAM page while they move in step)
    for (long long n0 = 0; n0 < N; n0 += 16 * NT) {                     // (probs x V: one chunk, x streamed once)
        f32x4 acc[NT];
#pragma unroll
        for (int tile = 0; tile < NT; ++tile) acc[tile] = f32x4{0, 0, 0, 0};
        // two x buffers used alternately (a register copy would wait for the prefetch at the end of every step); the
        // scheduling barriers keep the request order yt fragments -> next x, which the counted waits rely on
        XBlk xa, xb;
        load_xblk(xa, row, (long long)wave * 64, g, K);
        for (long long st = wave; st < nsteps; st += 8) {
            {
                BFrag<NT> bf[PLANES];
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) load_bfrag<NT>(bf[pl], ytb + pl * plane_stride, n0, st * 64, Kp, N, lane);
                __builtin_amdgcn_sched_barrier(0);
                load_xblk(xb, row, (st + 4) * 64, g, K);   // next step's x in flight under this one's work (behind K: a re-read)
                __builtin_amdgcn_sched_barrier(0);
                if (SOFTMAX) { mask_scores(xa, mrowp, st * 64, g, K, kvis); softmax_xblk(xa, row_max, row_sum, row_inv); }
                if ((st + 1) * 64 > K) mask_xblk(xa, st * 64, K);          // (uniform: the last, partial step only)
                quantise_xblk<FMT>(xa, a, lut, mbits, lane, afr, zfill);
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) mma_step<NT>(afr, bf[pl], n0, N, acc);
            }
            if (st + 4 >= nsteps) break;
            {
                BFrag<NT> bf[PLANES];
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) load_bfrag<NT>(bf[pl], ytb + pl * plane_stride, n0, (st + 4) * 64, Kp, N, lane);
                __builtin_amdgcn_sched_barrier(0);
                load_xblk(xa, row, (st + 8) * 64, g, K);
                __builtin_amdgcn_sched_barrier(0);
                if (SOFTMAX) { mask_scores(xb, mrowp, (st + 4) * 64, g, K, kvis); softmax_xblk(xb, row_max, row_sum, row_inv); }
                if ((st + 5) * 64 > K) mask_xblk(xb, (st + 4) * 64, K);
                quantise_xblk<FMT>(xb, a, lut, mbits, lane, afr, zfill);
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) mma_step<NT>(afr, bf[pl], n0, N, acc);
            }
        }
        __syncthreads();                                   // (previous chunk's partials have been read)
#pragma unroll
        for (int tile = 0; tile < NT; ++tile) red[wave][tile][lane] = acc[tile];
        __syncthreads();
#pragma unroll
        for (int tile = wave; tile < NT; tile += 4) {      // wave w sums tiles w, w + 4 of the four partials, in wave order
            if (n0 + 16 * tile >= N) break;
            f32x4 sum = red[0][tile][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 p = red[w][tile][lane];
                sum += p;
            }
            store_tile(sum, outb, m0, n0 + 16 * tile, M, N, lane);
        }
    }
}

// ---- kernel 3: the tile product (round 4) -----------------------------------------------------------------------------
// What bounded kernel 2 (profiles/r04_values_matmul.txt): a wave's yt fragments and its x prefetch share ONE in-order vector-
// memory queue, so a fragment asked for "now" comes back behind every x prefetch issued before it -- a K-step cost an HBM
// round trip however deep the prefetch, and the short-contraction form pulled the whole of yt[b] from L2 once per 16 rows
// (2 GB of L2 traffic at [32, 2048, 128] x [32, 128, 2048]).  Here a workgroup is 8 waves x 16 rows of x plus a NINTH wave that
// only feeds a ring of yt pieces in LDS (wave specialisation: no wave asks for a fragment "now", and every wave's queue holds one
// kind of traffic) --
//   STREAM (long contraction, P V):  piece = the yt fragments of one 64-step for NT column tiles (NT x 2 KiB): the feeder loads it
//     into registers a step ahead and ds_writes it (LDS-DMA moved no more than ~22 bytes a clock and compute unit); every compute
//     wave loads its own x slab (16 rows x 256 bytes) XD - 1 steps ahead into registers, passes it through its 4-KiB LDS patch
//     into a block-per-lane order and quantises in registers (quantise_lane); accumulators live across the steps;
//   RESIDENT (short contraction, Q K^T): the wave's x is quantised once into registers; piece = the yt fragments of one
//     64-column chunk for all NS steps (NS x 8 KiB, contiguous in yt), fetched by the feeder with LDS-DMA (hand-counted
//     s_waitcnt vmcnt; the compiler must not see the DMA: it would drain it at every LDS read); the chunk's 16 x 64 outputs leave
//     through the wave's LDS patch as four stores of 4 rows x 256 bytes;
// one s_barrier per piece.  Counting vmcnt only works where the number of LOADS per piece is fixed -- pieces and tiles beyond
// the end are clamped to the last valid one (the same data lands on the same place twice), never skipped -- and where no
// STORES share the wave's counter: loads return in order among themselves, stores among themselves, but not with respect to
// each other (a count that budgeted for the chunk stores let pieces through that had not landed).  Hence the feeder.
// PLANES > 1 (block_log's raw y as three bf16 planes): the planes of a step / chunk are consecutive pieces over the same x.
constexpr int TP_WAVES = 8;
#define MM_GLDS16(gp, lds) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds) : "memory")
#define MM_WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <bool STREAM, int NTNS, int FMT, int PLANES, int RW = 1, int CW = TP_WAVES>
__global__ __launch_bounds__((CW + 1) * 64) void bfp_qmatmul_tile_kernel(const QuantArgs a, const float* __restrict__ x,
                                                                         const uint16_t* __restrict__ yt, float* __restrict__ out,
                                                                         long long M, long long K, long long Kp, long long N,
                                                                         long long plane_stride, const unsigned* __restrict__ xstats,
                                                                         unsigned long long* __restrict__ stamps) {
    constexpr int NT = STREAM ? NTNS : 4;                 // column tiles of a piece
    constexpr int NS = STREAM ? 1 : NTNS;                 // 64-steps of a piece
    constexpr int SUB = NT * NS * 2;                      // 1-KiB sub-pieces of a piece
    constexpr int BW = STREAM ? SUB / CW : SUB;     // DMA instructions per issuing wave and piece
    static_assert(!STREAM || SUB % CW == 0, "a piece is shared evenly by the waves");
    constexpr int DB = STREAM ? 3 : (RW == 1 ? 2 : (CW == 4 ? 3 : 4));   // ring depth in pieces (RESIDENT: two -- with the patches two
                                                          // workgroups still share a compute unit's LDS up to 128 columns of x; two row
                                                          // groups a wave: the compute unit's one or two workgroups take a deeper ring --
                                                          // with one piece in flight the waves spent 60 % of a piece at its barrier)
    constexpr int XD = FMT == FMT_BFP ? 3 : 2;            // STREAM: x slabs in rotation, XD - 1 steps ahead (the other formats' element
                                                          // functions leave no registers for a third)
    constexpr int PIECE = SUB * 1024;
    // RW (RESIDENT): 16-row groups of x per compute wave.  Two: every yt fragment read from LDS feeds two MFMAs and a workgroup's
    // pass over yt[b] serves 256 rows -- with one group a piece's fragment reads (8 waves x the whole piece) take twice the LDS
    // time of the MFMAs they feed, three planes three times that (round 5)
    // CW: compute waves of a workgroup (RESIDENT with RW = 2 and three planes: four, so that two workgroups share a compute unit
    // and one's MFMAs run under the other's output stores)
    static_assert(RW == 1 || !STREAM, "the streamed form keeps one row group per wave");
    static_assert(CW == TP_WAVES || !STREAM, "the streamed form shares its pieces among eight waves");
    __shared__ Lut lut;
    __shared__ __attribute__((aligned(16))) unsigned char bring[DB * PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char xring[CW * 4096];     // the waves' transpose patches (STREAM: x in, RESIDENT: out)
    load_lut<FMT>(lut);
    float zfill = 1.0f;
    if (FMT == FMT_BL) zfill = bl_zero_fill<(CW + 1) * 64>(xstats);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const long long b = blockIdx.y, m0 = ((long long)blockIdx.x * CW + min(wave, CW - 1)) * 16 * RW;
    const long long mrow = min(m0 + (lane & 15), M - 1);                 // (rows past M: the last row again)
    const uint16_t* __restrict__ ytb = yt + b * (N >> 4) * Kp * 1024;
    float* __restrict__ orow = out + (b * M + mrow) * N + 4 * g;
    const int mbits = (int)__builtin_log2f(a.shift);
    const int nsteps = (int)Kp, ntiles = (int)(N >> 4);
    using lptr_t = __attribute__((address_space(3))) void*;
    const unsigned bring0 = (unsigned)(size_t)(lptr_t)bring;       // (the object's own LDS address)

    if (STREAM) {
        if (wave == CW) {
            // the feeder of the yt ring: plain loads into registers, ds_write into the slot.  (LDS-DMA moves ~22 bytes a clock
            // and compute unit -- measured: 805 MB of DMA, x and yt, took 70 us with every byte in cache -- so the DMA path is
            // left to the x slabs, which have no other way into LDS without passing the consumers' registers.)
            for (int t0 = 0; t0 < ntiles; t0 += NT) {
                const int niter = nsteps * PLANES;
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (a native vector: the HIP uint4 array stayed in scratch memory)
                u32x4 pc[SUB];
#define MM_FEED_LOAD(ITP_)                                                                                                   \
                {                                                                                                            \
                    const int itp_ = (ITP_) > niter - 1 ? niter - 1 : (ITP_);                                                \
                    const int kb_ = itp_ / PLANES, plb_ = itp_ - kb_ * PLANES;                                               \
                    _Pragma("unroll") for (int sub = 0; sub < SUB; ++sub) { /* (tile, t) = (sub / 2, sub % 2) */             \
                        const int tile_ = min(t0 + (sub >> 1), ntiles - 1);                                                  \
                        pc[sub] = *reinterpret_cast<const u32x4*>(ytb + plb_ * plane_stride +                                \
                                                                  (((long long)tile_ * Kp + kb_) * 2 + (sub & 1)) * 512 + lane * 8); \
                    }                                                                                                        \
                }
#define MM_FEED_STORE(ITP_)                                                                                                  \
                {                                                                                                            \
                    unsigned char* d_ = bring + ((ITP_) % DB) * PIECE + lane * 16;                                           \
                    _Pragma("unroll") for (int sub = 0; sub < SUB; ++sub) *reinterpret_cast<u32x4*>(d_ + sub * 1024) = pc[sub]; \
                }
                MM_FEED_LOAD(0) MM_FEED_STORE(0)
                MM_FEED_LOAD(1) MM_FEED_STORE(1)
                MM_FEED_LOAD(2)
                for (int it = 0; it < niter; ++it) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();               // piece `it` is there; the slot of piece it - 1 is free
                    MM_FEED_STORE(it + 2)
                    MM_FEED_LOAD(it + 3)
                }
#undef MM_FEED_LOAD
#undef MM_FEED_STORE
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            return;
        }
        for (int t0 = 0; t0 < ntiles; t0 += NT) {           // (more than NT column tiles: x is streamed again per group)
            // (x DMA: this lane's row inside each of the four row groups -- 4 i + lane / 16 -- rows past M: the last row again)
            const int xrow_lo = lane >> 4;                  // row = 4 i + xrow_lo, so row & 7 = 4 (i & 1) + xrow_lo
            // (addresses as ONE scalar base per step + four 32-bit lane offsets that never change: per-step 64-bit lane arithmetic
            //  had its temporaries share registers with slabs still in flight, and the compiler waited for those)
            const long long mbase = min(m0, M - 1);
            const char* xbase = reinterpret_cast<const char*>(x + (b * M + mbase) * K);
            unsigned xoff[4], xend[4];                      // (a workgroup's 128 rows: K < 2^23 keeps the offsets in 32 bits)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned row0 = (unsigned)(min((long long)(4 * i + xrow_lo), M - 1 - mbase) * K * 4);
                xoff[i] = row0 + 16u * ((lane & 15) ^ (4 * (i & 1) + xrow_lo));
                xend[i] = row0 + (unsigned)K * 4u - 16u;
            }
            // x: plain loads into registers, two steps ahead (three slabs of 16 VGPRs in rotation), then through the wave's own
            // 4-KiB patch of LDS into the block-per-lane order.  (As LDS-DMA the same bytes streamed at 3.7 TB/s whatever the
            // access shape and ring depth -- profiles/r04_values_matmul.txt.)  Instruction i loads rows 4 i .. 4 i + 3 of the
            // wave's 16, each as ONE 256-byte run over 16 lanes (whole cache lines); lane l writes its 16 bytes at l * 16, and
            // the run's 16-byte chunks are dealt to the lanes XOR (row & 7): the read-back (16 rows x the same chunk) then
            // meets no bank twice.
            auto xload = [&](f32x4 (&d)[4], int kx) {
                kx = kx > nsteps - 1 ? nsteps - 1 : kx;     // (behind the end: the last slab again, never used)
                // (no branch, no 64-bit lane arithmetic: chunks behind K -- the last, partial step -- read the row's last chunk
                //  instead and are zeroed with their block at the read-back)
                const unsigned koff = (unsigned)kx * 256u;
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const f32x4*>(xbase + min(xoff[i] + koff, xend[i]));
            };
            f32x4 acc[NT];
#pragma unroll
            for (int tile = 0; tile < NT; ++tile) acc[tile] = f32x4{0, 0, 0, 0};
            bf16x8 afr[2];
            unsigned long long t_wait = 0, t_bar = 0;
            const unsigned long long t_begin = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            unsigned char* xpatch = xring + wave * 4096;
            f32x4 xq[XD][4];                   // (native vectors: arrays of HIP's float4 were left in scratch memory)
#pragma unroll
            for (int u = 0; u < XD - 1; ++u) xload(xq[u], u);
#define MM_STREAM_STEP(K_, CUR_, FILL_)                                                                                      \
            {                                                                                                                \
                const int k = (K_);                                                                                          \
                xload(FILL_, k + XD - 1);                                                                                    \
                _Pragma("unroll") for (int pl = 0; pl < PLANES; ++pl) {                                                      \
                    const int it = k * PLANES + pl;                                                                          \
                    const unsigned long long tw1 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;                             \
                    __builtin_amdgcn_s_barrier();                /* piece `it` is in the ring */                             \
                    if (stamps) t_bar += __builtin_amdgcn_s_memtime() - tw1;                                                 \
                    if (pl == 0) {                                                                                           \
                        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
                            *reinterpret_cast<f32x4*>(xpatch + i * 1024 + lane * 16) = CUR_[i];                              \
                        f32x4 xv[4];                                                                                         \
                        const int r = lane & 15;                 /* lane (r, g) takes block g = chunks 4 g .. 4 g + 3 of row r */ \
                        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                            xv[j] = *reinterpret_cast<const f32x4*>(xpatch + r * 256 + (((4 * g + j) ^ (r & 7)) << 4));      \
                        if ((long long)k * 64 + 16 * g >= K) {   /* the last, partial step: blocks behind K */               \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};                 \
                        }                                                                                                    \
                        quantise_lane<FMT>(xv, a, lut, mbits, zfill, afr);                                                   \
                    }                                                                                                        \
                    const unsigned char* bs = bring + (it % DB) * PIECE + lane * 16;                                         \
                    _Pragma("unroll") for (int tile = 0; tile < NT; ++tile) {                                                \
                        if (t0 + tile >= ntiles) break;          /* (uniform) */                                             \
                        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                        \
                            acc[tile] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                             \
                                __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bs + (tile * 2 + t) * 1024)), afr[t], acc[tile], 0, 0, 0); \
                    }                                                                                                        \
                }                                                                                                            \
            }
            for (int k0 = 0; k0 < nsteps; k0 += XD) {           // (XD steps per trip: the slabs keep their registers)
#pragma unroll
                for (int u = 0; u < XD; ++u) {
                    if (k0 + u >= nsteps) break;
                    MM_STREAM_STEP(k0 + u, xq[u], xq[(u + XD - 1) % XD])
                }
            }
#undef MM_STREAM_STEP
            (void)t_wait;
            if (m0 + (lane & 15) < M) {
#pragma unroll
                for (int tile = 0; tile < NT; ++tile)
                    if (t0 + tile < ntiles)
                        *reinterpret_cast<float4*>(orow + (t0 + tile) * 16) = make_float4(acc[tile][0], acc[tile][1], acc[tile][2], acc[tile][3]);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // (the group's last reads of the ring are through)
            if (stamps && lane == 0 && (wave == 0 || wave == CW - 1)) {      // diagnostic: cycles in the loop, waiting for loads, at barriers
                unsigned long long* d = stamps + ((blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wave != 0)) * 4;
                d[0] = __builtin_amdgcn_s_memtime() - t_begin; d[1] = t_wait; d[2] = t_bar; d[3] = nsteps;
            }
        }
        return;
    }
    // RESIDENT: wave 8 feeds the ring, waves 0..7 quantise their 16 rows once and take the pieces as they land
    const int nchunks = (ntiles + 3) >> 2, niter = nchunks * PLANES;
    if (wave == CW) {
        auto issue = [&](int j) {
            int itp = j + DB - 1;
            itp = itp < 0 ? 0 : (itp > niter - 1 ? niter - 1 : itp);
            const int cb = itp / PLANES, plb = itp - cb * PLANES;
            const unsigned dst = bring0 + (itp % DB) * PIECE;
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {           // (tile, step, t) = (sub / (2 NS), ...): yt's own order
                const int tile = min(cb * 4 + sub / (2 * NS), ntiles - 1);
                const uint16_t* gp = ytb + plb * plane_stride + ((long long)tile * Kp * 2 + sub % (2 * NS)) * 512 + lane * 8;
                MM_GLDS16(gp, dst + sub * 1024);
            }
        };
        MM_WAITV(0);                                        // (the table loads: the counts below start from an empty queue)
        for (int j = -(DB - 1); j < 0; ++j) issue(j);
        for (int it = 0; it < niter; ++it) {
            MM_WAITV(((DB - 2) * BW));                      // (piece `it` has landed; the DB - 2 pieces requested behind it may fly)
            __builtin_amdgcn_s_barrier();
            issue(it);
        }
        MM_WAITV(0);
        return;
    }
    bf16x8 res[RW][NS][2];
#pragma unroll
    for (int rg = 0; rg < RW; ++rg) {
        const float* __restrict__ rowg = x + (b * M + min(m0 + 16 * rg + (lane & 15), M - 1)) * K;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            f32x4 xv[4];
            const long long kk = st * 64 + 16 * g;          // lane (r, g): block g of the step, whole
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const f32x4*>(rowg + (kk < K ? kk : 0) + 4 * j);
            if (kk >= K) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            quantise_lane<FMT>(xv, a, lut, mbits, zfill, res[rg][st]);
        }
    }
    f32x4 acc[RW][4];
    unsigned char* opatch = xring + wave * 4096;
    char* obase[RW];                                        // (scalar bases + constant 32-bit lane offsets)
#pragma unroll
    for (int rg = 0; rg < RW; ++rg) obase[rg] = reinterpret_cast<char*>(out + (b * M + min(m0 + 16 * rg, M - 1)) * N);
    unsigned ooff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ooff[j] = (unsigned)(((long long)(4 * j + g) * N + 4 * ((lane & 15) ^ (4 * j + g))) * 4);
    unsigned long long t_bar = 0;
    const unsigned long long t_begin = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            const int it = c * PLANES + pl;
            asm volatile("" ::: "memory");
            const unsigned long long tw1 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            __builtin_amdgcn_s_barrier();
            if (stamps) t_bar += __builtin_amdgcn_s_memtime() - tw1;
            asm volatile("" ::: "memory");
            if (pl == 0) {
#pragma unroll
                for (int rg = 0; rg < RW; ++rg)
#pragma unroll
                    for (int tile = 0; tile < 4; ++tile) acc[rg][tile] = f32x4{0, 0, 0, 0};
            }
            const unsigned char* bs = bring + (it % DB) * PIECE + lane * 16;
#pragma unroll
            for (int tile = 0; tile < 4; ++tile)
#pragma unroll
                for (int st = 0; st < NS; ++st)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bs + ((tile * NS + st) * 2 + t) * 1024));
#pragma unroll
                        for (int rg = 0; rg < RW; ++rg) acc[rg][tile] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, res[rg][st][t], acc[rg][tile], 0, 0, 0);
                    }
            if (pl == PLANES - 1) {
                // The chunk's 16 x 64 outputs through the wave's 4-KiB patch of LDS, so that a store instruction writes four
                // rows x 256 contiguous bytes instead of sixteen rows x 64 (the MFMA layout): whole cache lines per request.
                // Chunks of 16 bytes XOR (row & 15): no bank twice either way.  (Non-temporal stores: no difference.)
                // (row groups one after the other through the same patch: a wave's LDS operations complete in order)
                const int r = lane & 15;
#pragma unroll
                for (int rg = 0; rg < RW; ++rg) {
#pragma unroll
                    for (int tile = 0; tile < 4; ++tile)
                        *reinterpret_cast<f32x4*>(opatch + r * 256 + (((tile * 4 + g) ^ r) << 4)) = acc[rg][tile];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rw = 4 * j + g, ch = (lane & 15) ^ rw;          // this lane's row and (logical) chunk of it
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(opatch + rw * 256 + ((lane & 15) << 4));
                        if (m0 + 16 * rg + rw < M && c * 4 + (ch >> 2) < ntiles)
                            *reinterpret_cast<f32x4*>(obase[rg] + ooff[j] + (unsigned)c * 256u) = v4;
                    }
                }
            }
        }
    }
    if (stamps && lane == 0 && (wave == 0 || wave == CW - 1)) {
        unsigned long long* d = stamps + ((blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wave != 0)) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - t_begin; d[1] = 0; d[2] = t_bar; d[3] = niter;
    }
}

// block_log's statistics pass over x: the smallest non-zero [1,16]-block maximum of the whole tensor (as its bit pattern:
// positive floats order like unsigned integers), the fill the reference gives all-zero blocks (block_fp.py:54-58).
// Launched with BL_STATS_SLOTS workgroups, one slot each (bl_zero_fill).
__global__ __launch_bounds__(256) void bl_block_stats_kernel(const float4* __restrict__ x4, long long n4, unsigned* __restrict__ stats) {
    __shared__ unsigned part[4];
    unsigned best = 0xFFFFFFFFu;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {   // (n4 % 4 == 0: whole blocks per quad)
        const float4 v = x4[i];
        const float bmax = group_max<4>(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        if (bmax > 0.f) best = min(best, __float_as_uint(bmax));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) stats[blockIdx.x] = min(min(part[0], part[1]), min(part[2], part[3]));      // (this workgroup's slot)
}

static bool tile_route() {
    static const bool on = []() { const char* e = getenv("MI355Q_MATMUL_TILE"); return !(e && e[0] == '0'); }();
    return on;
}
static unsigned long long* g_tp_stamps = nullptr;      // diagnostic (tools/dbg/tp_stamps.py): per-workgroup cycle counts of the tile kernel

template <int FMT, int PLANES>
static int launch_qmatmul_fmt(const QuantArgs& ax, const float* x, const uint16_t* yt, float* out, long long B, long long M,
                              long long K, long long Kp, long long N, hipStream_t st, bool softmax, const float* mask,
                              long long causal_off, long long plane_stride, const unsigned* xstats) {
    // plain products: the tile kernel (MI355Q_MATMUL_TILE=0: kernel 2, for comparisons); the softmax forms stay on kernel 2
    // (diagnostic: MI355Q_MATMUL_DBG=1 stops behind the y pack [and block_log's statistics pass] -- tools/dbg/vm_breakdown.py)
    static const bool pack_only = []() { const char* e = getenv("MI355Q_MATMUL_DBG"); return e && e[0] == '1'; }();
    if (pack_only) return 0;
    if (!softmax && tile_route()) {
        // (short contraction: two row groups per wave -- see RW)
        static const int rw_env = []() { const char* e = getenv("MI355Q_MATMUL_RW"); return e ? atoi(e) : 0; }();
        const int rw = Kp <= 3 ? (rw_env ? rw_env : (PLANES > 1 ? 2 : 1)) : 1;
        const int rows = TP_WAVES * 16 * rw;
        dim3 g3((unsigned)((M + rows - 1) / rows), (unsigned)B);
#define MI355Q_TP_LAUNCH(STREAM_, NTNS_, RW_)                                                                              \
        hipLaunchKernelGGL((bfp_qmatmul_tile_kernel<STREAM_, NTNS_, FMT, PLANES, RW_>), g3, (TP_WAVES + 1) * 64, 0, st, ax, x, yt, out, M, K, \
                           Kp, N, plane_stride, xstats, g_tp_stamps)
        if (Kp == 1 && rw == 2) MI355Q_TP_LAUNCH(false, 1, 2);
        else if (Kp == 2 && rw == 2) MI355Q_TP_LAUNCH(false, 2, 2);
        else if (Kp == 3 && rw == 2) MI355Q_TP_LAUNCH(false, 3, 2);
        else if (Kp == 1) MI355Q_TP_LAUNCH(false, 1, 1);
        else if (Kp == 2) MI355Q_TP_LAUNCH(false, 2, 1);
        else if (Kp == 3) MI355Q_TP_LAUNCH(false, 3, 1);
        else if (N <= 64) MI355Q_TP_LAUNCH(true, 4, 1);
        else MI355Q_TP_LAUNCH(true, 8, 1);
#undef MI355Q_TP_LAUNCH
        return (int)hipGetLastError();
    }
    dim3 g2((unsigned)((M + 15) / 16), (unsigned)B);
    if (softmax && causal_off >= 0) g2 = dim3((unsigned)B, (unsigned)((M + 15) / 16));      // (see the kernel: heavy_first)
#define MI355Q_MM_LAUNCH(...)                                                                                              \
    hipLaunchKernelGGL((bfp_qmatmul_kernel<__VA_ARGS__, FMT, PLANES>), g2, 256, 0, st, ax, x, yt, out, M, K, Kp, N, mask,   \
                       causal_off, plane_stride, xstats)
    if (Kp == 1)
        MI355Q_MM_LAUNCH(true, 1, false);
    else if (Kp == 2)
        MI355Q_MM_LAUNCH(true, 2, false);
    else if (Kp == 3)
        MI355Q_MM_LAUNCH(true, 3, false);
    else if constexpr (PLANES > 1) {                       // (three fragment sets per step: column chunks of 64)
        if (softmax) return MI355Q_E_UNSUPPORTED;
        MI355Q_MM_LAUNCH(false, 4, false);
    } else if (softmax && N <= 64)
        MI355Q_MM_LAUNCH(false, 4, true);
    else if (softmax && N <= 128)
        MI355Q_MM_LAUNCH(false, 8, true);
    else if (softmax)
        return MI355Q_E_UNSUPPORTED;                       // (one pass over x only: head_dim <= 128)
    else if (N <= 64)
        MI355Q_MM_LAUNCH(false, 4, false);
    else    // (head_dim 128: both halves of the columns in one pass over x)
        MI355Q_MM_LAUNCH(false, 8, false);
#undef MI355Q_MM_LAUNCH
    return (int)hipGetLastError();
}

// fmt: 0 block_fp, 1 block_minifloat (x and y), 2 block_log (x; y raw, as three exact bf16 planes).  `yt`: the workspace of
// mi355q_bfp_matmul_workspace_bytes (block_log: mi355q_block_log_matmul_workspace_bytes: three planes + the statistics slots).
int launch_bfp_qmatmul(const QuantArgs& ax, const QuantArgs& ay, const float* x, const float* y, float* out, void* yt,
                       long long B, long long M, long long K, long long N, hipStream_t st, bool softmax, const float* mask,
                       long long causal_off, int fmt) {
    if (softmax && (K + 63) / 64 <= 3) return MI355Q_E_UNSUPPORTED;     // (short rows: the caller takes softmax + the plain entry)
    if (softmax && fmt == FMT_BL) return MI355Q_E_UNSUPPORTED;          // (block_log zeros are not zeros: no causal short cut)
    dim3 g1((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64), (unsigned)B);
    const long long Kp = (K + 63) / 64;                    // 64-steps: yt is stored in fragment order (kernel 1)
    const long long plane = B * Kp * 64 * N;               // elements of one plane
    uint16_t* ytp = static_cast<uint16_t*>(yt);
    hipError_t e;
    unsigned* stats = fmt == FMT_BL ? reinterpret_cast<unsigned*>(ytp + 3 * plane) : nullptr;      // (block_log: BL_STATS_SLOTS words behind the three planes)
    const bool lane_order = !softmax && tile_route();       // which product kernel follows: its K order inside a 64-step
#define MI355Q_PACK(FMT_)                                                                                                  \
    if (lane_order) hipLaunchKernelGGL((bfp_quant_pack_t_kernel<FMT_, true>), g1, 256, 0, st, ay, y, ytp, K, Kp, N, plane);   \
    else hipLaunchKernelGGL((bfp_quant_pack_t_kernel<FMT_, false>), g1, 256, 0, st, ay, y, ytp, K, Kp, N, plane)
    if (fmt == FMT_BFP) {
        MI355Q_PACK(FMT_BFP);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        return launch_qmatmul_fmt<FMT_BFP, 1>(ax, x, ytp, out, B, M, K, Kp, N, st, softmax, mask, causal_off, plane, nullptr);
    }
    if (fmt == FMT_BM) {
        MI355Q_PACK(FMT_BM);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        return launch_qmatmul_fmt<FMT_BM, 1>(ax, x, ytp, out, B, M, K, Kp, N, st, softmax, mask, causal_off, plane, nullptr);
    }
    if (fmt != FMT_BL) return MI355Q_E_BADARG;
    hipLaunchKernelGGL(bl_block_stats_kernel, dim3(BL_STATS_SLOTS), 256, 0, st, reinterpret_cast<const float4*>(x), B * M * K / 4, stats);
    MI355Q_PACK(FMT_RAW);
#undef MI355Q_PACK
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    return launch_qmatmul_fmt<FMT_BL, 3>(ax, x, ytp, out, B, M, K, Kp, N, st, false, mask, causal_off, plane, stats);
}

}  // namespace mi355q

// diagnostic hook, not part of include/mi355q.h: a device buffer ([workgroups][2][4] 64-bit words) the tile product kernel
// fills with {cycles in its loop, cycles waiting for its loads, cycles at barriers, pieces}; NULL switches it off
extern "C" __attribute__((visibility("default"))) void mi355q_debug_tp_stamps(void* buf) { mi355q::g_tp_stamps = static_cast<unsigned long long*>(buf); }
