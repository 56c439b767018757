// mi355q_matmul.hip -- block_fp quantised batched matmul  out[b] = Qx(x[b]) @ Qy(y[b])  (reference
// quantized_functions/matmul.py:146-196: x quantised along its last dim = the contraction, y along ITS last dim = the
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

// ---- kernel 1: y [B, K, N] fp32 -> fake-quantise along N -> yt [B, N, K] bf16 ------------------------------------
__global__ __launch_bounds__(256) void bfp_quant_pack_t_kernel(const QuantArgs a, const float* __restrict__ y,
                                                               uint16_t* __restrict__ yt, long long K, long long Kp, long long N) {
    __shared__ Lut lut;
    __shared__ uint16_t tile[64][64 + 8];                 // [n][k], row padded against bank conflicts
    load_lut<FMT_BFP>(lut);
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
        if (bmax != 0.f) {
            unsigned code;
            const BlockParam bp = block_param<FMT_BFP>(bmax, a, lut, code);
            int mant;
            q[0] = quant_elem<FMT_BFP>(v.x, bp, a, lut, mant);
            q[1] = quant_elem<FMT_BFP>(v.y, bp, a, lut, mant);
            q[2] = quant_elem<FMT_BFP>(v.z, bp, a, lut, mant);
            q[3] = quant_elem<FMT_BFP>(v.w, bp, a, lut, mant);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[n4 * 4 + j][mm_kperm(kr + 16 * i)] = (uint16_t)(pack_bf16(q[j], 0.f) & 0xFFFFu);
    }
    __syncthreads();
    // 64 rows (n) x 64 k: thread writes 16 bytes (8 k) of one row
    // FRAGMENT ORDER: the 16 bytes lane (c, g) of kernel 2 takes for (16-column tile, 64-step, MFMA t) lie at lane * 16
    // inside a 1-KiB piece, pieces ordered [tile][step][t]: a wave's fragment load is one contiguous KiB (8 cache lines
    // instead of 16 half lines -- the fragment loads are L1-access bound).  Kp = number of 64-steps here.
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = tid + 256 * i, ntl = slot >> 7, t = (slot >> 6) & 1, ln = slot & 63;
        if (n0 + 16 * ntl < N)                            // (whole 64-groups: zeros behind K)
            *reinterpret_cast<uint4*>(yt + ((((b * (N >> 4) + (n0 >> 4) + ntl) * Kp + blockIdx.y) * 2 + t) * 64 + ln) * 8) =
                *reinterpret_cast<const uint4*>(&tile[16 * ntl + (ln & 15)][32 * t + 8 * (ln >> 4)]);
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

// One block_fp element given its block's exponent p, in 8 VALU operations instead of quant_elem's 14, with the same
// result for every input (mi355q_quant_dev.h, block_fp.py:69-94):
//   * sign(x + 1e-9) is only used where |x| > 1e-8 (smaller |x| pass through as x), and there it is the sign of x;
//   * ldexp(v, -p) * 2^mb = ldexp(v, mb - p): where the first product would round (a subnormal intermediate), both are
//     far below 0.5 and round to mantissa 0;  rint of a positive number needs no lower clamp;
//   * sign * 2^p * (m * 2^-mb) = copysign(ldexp(m, p - mb), x): m 2^(p - mb) is a multiple of 2^-149 and below 2^128,
//     so both forms are exact.
__device__ __forceinline__ float quant_elem_fused(float x, int up, int down, float mant_max) {
    const float m = fminf(__builtin_rintf(__builtin_ldexpf(fabsf(x) + EPS9, up)), mant_max);
    const float q = __builtin_copysignf(__builtin_ldexpf(m, down), x);
    return fabsf(x) <= ATOL ? x : q;
}

// quantise the wave's step: lane (r, g) holds float4 g of blocks 0..3 of row r.  The shared exponent of block i is
// worked out by lane group i only (threshold lookup and clamp once per block, not four times) and fetched by the others.
__device__ __forceinline__ void quantise_xblk(const XBlk& s, const QuantArgs& a, const Lut& lut, int mbits, int lane,
                                              bf16x8 (&afr)[2]) {
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
    const int pmine = block_param<FMT_BFP>(mine, a, lut, code).p;         // (an all-zero block: any exponent, see below)
    float q[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = __shfl(pmine, (lane & 15) + 16 * i);
        const float4 v = s.v[i];
        // (elements of an all-zero block are <= 1e-8 and pass through as zeros)
        q[4 * i + 0] = quant_elem_fused(v.x, mbits - p, p - mbits, a.mant_max);
        q[4 * i + 1] = quant_elem_fused(v.y, mbits - p, p - mbits, a.mant_max);
        q[4 * i + 2] = quant_elem_fused(v.z, mbits - p, p - mbits, a.mant_max);
        q[4 * i + 3] = quant_elem_fused(v.w, mbits - p, p - mbits, a.mant_max);
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

constexpr int MM_RESIDENT_STEPS = 3;     // contraction steps (of 64) whose quantised x stays in registers

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

template <bool RESIDENT, int NT, bool SOFTMAX = false>
__global__ __launch_bounds__(256) void bfp_qmatmul_kernel(const QuantArgs a, const float* __restrict__ x,
                                                          const uint16_t* __restrict__ yt, float* __restrict__ out,
                                                          long long M, long long K, long long Kp, long long N,
                                                          const float* __restrict__ mask, long long causal_off) {
    __shared__ Lut lut;
    __shared__ f32x4 red[4][NT][64];                      // [wave][tile][lane]: split-K partial tiles
    __shared__ float stat[4][16];
    load_lut<FMT_BFP>(lut);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const long long b = blockIdx.y, m0 = (long long)blockIdx.x * 16;
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
            quantise_xblk(s, a, lut, mbits, lane, res[st]);
        }
        // Two fragment sets used alternately: the next chunk's yt fragments are requested BEFORE this chunk's stores --
        // vmcnt counts loads and stores in one queue, so fragments requested behind the stores could only be used once
        // those writes were acknowledged.
        BFrag<4> b0[NS], b1[NS];
        const long long step = 4 * MM_NCHUNK, nlast = ((N - 1) / MM_NCHUNK) * MM_NCHUNK;
        long long n0 = (long long)wave * MM_NCHUNK;
#pragma unroll
        for (int st = 0; st < NS; ++st) load_bfrag<4>(b0[st], ytb, min(n0, nlast), st * 64, Kp, N, lane);
#define MI355Q_MM_CHUNK(USE_, FILL_)                                                                         \
        {                                                                                                   \
            f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};                          \
            _Pragma("unroll") for (int st = 0; st < NS; ++st) mma_step<4>(res[st], USE_[st], n0, N, acc);   \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            _Pragma("unroll") for (int st = 0; st < NS; ++st)                                               \
                load_bfrag<4>(FILL_[st], ytb, min(n0 + step, nlast), st * 64, Kp, N, lane);                 \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            _Pragma("unroll") for (int tile = 0; tile < 4; ++tile)                                          \
                if (n0 + 16 * tile < N) store_tile(acc[tile], outb, m0, n0 + 16 * tile, M, N, lane);       \
            n0 += step;                                                                                     \
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
                BFrag<NT> bf;
                load_bfrag<NT>(bf, ytb, n0, st * 64, Kp, N, lane);
                __builtin_amdgcn_sched_barrier(0);
                load_xblk(xb, row, (st + 4) * 64, g, K);   // next step's x in flight under this one's work (behind K: a re-read)
                __builtin_amdgcn_sched_barrier(0);
                if (SOFTMAX) { mask_scores(xa, mrowp, st * 64, g, K, kvis); softmax_xblk(xa, row_max, row_sum, row_inv); }
                if ((st + 1) * 64 > K) mask_xblk(xa, st * 64, K);          // (uniform: the last, partial step only)
                quantise_xblk(xa, a, lut, mbits, lane, afr);
                mma_step<NT>(afr, bf, n0, N, acc);
            }
            if (st + 4 >= nsteps) break;
            {
                BFrag<NT> bf;
                load_bfrag<NT>(bf, ytb, n0, (st + 4) * 64, Kp, N, lane);
                __builtin_amdgcn_sched_barrier(0);
                load_xblk(xa, row, (st + 8) * 64, g, K);
                __builtin_amdgcn_sched_barrier(0);
                if (SOFTMAX) { mask_scores(xb, mrowp, (st + 4) * 64, g, K, kvis); softmax_xblk(xb, row_max, row_sum, row_inv); }
                if ((st + 5) * 64 > K) mask_xblk(xb, (st + 4) * 64, K);
                quantise_xblk(xb, a, lut, mbits, lane, afr);
                mma_step<NT>(afr, bf, n0, N, acc);
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

int launch_bfp_qmatmul(const QuantArgs& ax, const QuantArgs& ay, const float* x, const float* y, float* out, void* yt,
                       long long B, long long M, long long K, long long N, hipStream_t st, bool softmax, const float* mask,
                       long long causal_off) {
    if (softmax && (K + 63) / 64 <= 3) return MI355Q_E_UNSUPPORTED;     // (short rows: the caller takes softmax + the plain entry)
    dim3 g1((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64), (unsigned)B);
    const long long Kp = (K + 63) / 64;                    // 64-steps: yt is stored in fragment order (kernel 1)
    hipLaunchKernelGGL(bfp_quant_pack_t_kernel, g1, 256, 0, st, ay, y, static_cast<uint16_t*>(yt), K, Kp, N);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    dim3 g2((unsigned)((M + 15) / 16), (unsigned)B);
    if (Kp == 1)
        hipLaunchKernelGGL((bfp_qmatmul_kernel<true, 1>), g2, 256, 0, st, ax, x, static_cast<const uint16_t*>(yt), out, M, K, Kp, N, mask, causal_off);
    else if (Kp == 2)
        hipLaunchKernelGGL((bfp_qmatmul_kernel<true, 2>), g2, 256, 0, st, ax, x, static_cast<const uint16_t*>(yt), out, M, K, Kp, N, mask, causal_off);
    else if (Kp == 3)
        hipLaunchKernelGGL((bfp_qmatmul_kernel<true, 3>), g2, 256, 0, st, ax, x, static_cast<const uint16_t*>(yt), out, M, K, Kp, N, mask, causal_off);
    else if (softmax && N <= 64)
        hipLaunchKernelGGL((bfp_qmatmul_kernel<false, 4, true>), g2, 256, 0, st, ax, x, static_cast<const uint16_t*>(yt), out, M, K, Kp, N, mask, causal_off);
    else if (softmax && N <= 128)
        hipLaunchKernelGGL((bfp_qmatmul_kernel<false, 8, true>), g2, 256, 0, st, ax, x, static_cast<const uint16_t*>(yt), out, M, K, Kp, N, mask, causal_off);
    else if (softmax)
        return MI355Q_E_UNSUPPORTED;                       // (one pass over x only: head_dim <= 128)
    else if (N <= 64)
        hipLaunchKernelGGL((bfp_qmatmul_kernel<false, 4>), g2, 256, 0, st, ax, x, static_cast<const uint16_t*>(yt), out, M, K, Kp, N, mask, causal_off);
    else    // (head_dim 128: both halves of the columns in one pass over x)
        hipLaunchKernelGGL((bfp_qmatmul_kernel<false, 8>), g2, 256, 0, st, ax, x, static_cast<const uint16_t*>(yt), out, M, K, Kp, N, mask, causal_off);
    return (int)hipGetLastError();
}

}  // namespace mi355q
