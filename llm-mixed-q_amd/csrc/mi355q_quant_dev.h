// mi355q_quant_dev.h -- device-side arithmetic of the block quantisers, shared by the streaming kernels
// (mi355q_quant.hip) and the fused quantise + matmul kernels (mi355q_matmul.hip): threshold tables for exact
// ceil / floor / rint of log2, the block parameter and the per-element fake-quantisation of each format.
#ifndef MI355Q_QUANT_DEV_H
#define MI355Q_QUANT_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_internal.h"

#define MI355Q_TABLE_QUAL static __device__ const
#include "log2_tables.inc"

namespace mi355q {

constexpr int FMT_BFP = 0, FMT_BM = 1, FMT_BL = 2;
constexpr int LUT_N = MI355Q_LOG2_TABLE_SIZE;
constexpr float EPS9 = 1e-9f;
constexpr float ATOL = 1e-8f;

// workspace words
constexpr int WS_ZERO_FLAG = 0;   // kernel 1 met an all-zero block
constexpr int WS_FILL_GUESS = 4;  // fp32 bits of the last fix-up launch's fill (0: none yet), see zero_fill_guess
constexpr int WS_TICKET = 2;      // fix-up kernel exit ticket (the last workgroup out lowers the flag)
// one word per workgroup of kernel 1: max over its non-zero blocks of ~bits(block max)
constexpr int WS_SLOT0 = 64, WS_SLOTS = 2048;
// exit tickets of the fix-up kernel in two levels: 32 first-level counters a cache line apart (workgroup b takes counter
// b % 32; the last of each takes a ticket of WS_TICKET) -- thousands of returning atomics on ONE word are served one after the
// other at ~60 ns each (r03: 2048 workgroups = 125 us, the whole duration of the launch)
constexpr int WS_TICKET_L1 = WS_SLOT0 + WS_SLOTS, WS_TICKET_L1_N = 32, WS_TICKET_L1_STRIDE = 32;
static_assert((WS_TICKET_L1 + WS_TICKET_L1_N * WS_TICKET_L1_STRIDE) * 4 <= MI355Q_WORKSPACE_BYTES, "workspace too small for the exit tickets");
static_assert((WS_SLOT0 + WS_SLOTS) * 4 <= MI355Q_WORKSPACE_BYTES, "workspace too small for the per-workgroup slots");

struct Lut {
    unsigned a[LUT_N];  // bfp: ceil ; bm: floor ; bl: ceil
    unsigned lo[LUT_N]; // bl: rnd_lo
    unsigned hi[LUT_N]; // bl: rnd_hi
};

template <int FMT, bool PLAIN = false>
__device__ __forceinline__ void load_lut(Lut& lut) {
    if constexpr (PLAIN) {      // (the loop as the compiler likes it: the one-pass attention kernels, whose register budget was tuned around it)
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) {
            lut.a[i] = (FMT == FMT_BM) ? mi355q_log2_floor_thr[i] : mi355q_log2_ceil_thr[i];
            if (FMT == FMT_BL) {
                lut.lo[i] = mi355q_log2_rnd_lo[i];
                lut.hi[i] = mi355q_log2_rnd_hi[i];
            }
        }
        __syncthreads();
        return;
    }
    // (one entry per thread and trip, NOT unrolled: left to itself the compiler runs eight trips' loads together with 64-bit
    //  addresses -- 84 registers for block_log's three tables, which set the streaming quantiser's occupancy at 5 waves a SIMD)
#pragma unroll 1
    for (int i = threadIdx.x; i < LUT_N; i += 2 * blockDim.x) {     // (two entries a trip, their loads together: 277 entries = one trip of 256 threads)
        const int j = i + blockDim.x, jc = j < LUT_N ? j : i;
        const unsigned a0 = (FMT == FMT_BM) ? mi355q_log2_floor_thr[i] : mi355q_log2_ceil_thr[i];
        const unsigned a1 = (FMT == FMT_BM) ? mi355q_log2_floor_thr[jc] : mi355q_log2_ceil_thr[jc];
        unsigned l0 = 0u, h0 = 0u, l1 = 0u, h1 = 0u;
        if (FMT == FMT_BL) {
            l0 = mi355q_log2_rnd_lo[i]; h0 = mi355q_log2_rnd_hi[i];
            l1 = mi355q_log2_rnd_lo[jc]; h1 = mi355q_log2_rnd_hi[jc];
        }
        lut.a[i] = a0;
        lut.a[jc] = a1;
        if (FMT == FMT_BL) {
            lut.lo[i] = l0; lut.hi[i] = h0;
            lut.lo[jc] = l1; lut.hi[jc] = h1;
        }
    }
    __syncthreads();
}

// v > 0 (finite or +inf): v = 2^k (1 + m 2^-23); subnormals normalised, inf -> k = 128, m = 0.
__device__ __forceinline__ void split_pos(float v, int& k, unsigned& m) {
    const unsigned b = __float_as_uint(v) & 0x7FFFFFFFu;
    const unsigned E = b >> 23, M = b & 0x7FFFFFu;
    if (E == 0u) {
        const int p = 31 - __clz((int)(M | 1u));
        k = p - 149;
        m = (M << (23 - p)) & 0x7FFFFFu;
    } else {
        k = (int)E - 127;
        m = (E == 255u) ? 0u : M;
    }
}
__device__ __forceinline__ int lut_index(int k) {
    const int i = k + MI355Q_LOG2_K_OFFSET;
    return i > LUT_N - 1 ? LUT_N - 1 : i;
}
__device__ __forceinline__ int ceil_log2(float v, const Lut& lut) {
    int k; unsigned m;
    split_pos(v, k, m);
    return k + ((m != 0u && m >= lut.a[lut_index(k)]) ? 1 : 0);
}
// (per-ELEMENT uses: the table matters only within a few ulps of the next power of two -- M >= FLOOR_THR_MIN over all
// binades -- so the lookup, an LDS read whose address depends on the exponent, is skipped unless some lane of the wave
// is that close; both paths give the same value for every lane)
__device__ __forceinline__ int floor_log2(float v, const Lut& lut) {
    int k; unsigned m;
    split_pos(v, k, m);
    if (!__any(m >= MI355Q_LOG2_FLOOR_THR_MIN)) return k;
    return k + ((k < 128 && m >= lut.a[lut_index(k)]) ? 1 : 0);
}
// (the tie bands of all binades lie in [RND_LO_MIN, RND_HI_MAX], 125 fraction values around sqrt(2): no lookup unless
// some lane of the wave falls in there)
__device__ __forceinline__ int rint_log2(float v, const Lut& lut) {
    int k; unsigned m;
    split_pos(v, k, m);
    if (!__any(m >= MI355Q_LOG2_RND_LO_MIN && m <= MI355Q_LOG2_RND_HI_MAX))
        return k >= 128 ? 128 : k + (m > MI355Q_LOG2_RND_HI_MAX ? 1 : 0);
    if (k >= 128) return 128;
    const int i = lut_index(k);
    const int even = k + (k & 1);
    return m < lut.lo[i] ? k : (m > lut.hi[i] ? k + 1 : even);
}
__device__ __forceinline__ float sgn(float t) { return t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

struct BlockParam {
    int p;      // bfp: shared exponent e ; bm / bl: shared bias
    float eps;  // bl: 0.1 * 2^-bias
};

// block max (> 0) -> block parameter and its stored byte
template <int FMT>
__device__ __forceinline__ BlockParam block_param(float bmax, const QuantArgs& a, const Lut& lut, unsigned& code) {
    BlockParam bp;
    bp.eps = 0.f;
    if (FMT == FMT_BFP) {           // block_fp.py:72-73
        bp.p = clampi(ceil_log2(bmax, lut), a.e_min, a.e_max);
        code = (unsigned)(bp.p + a.code_bias);
    } else if (FMT == FMT_BM) {     // block_minifloat.py:57-59
        bp.p = clampi(floor_log2(bmax, lut), 0, a.bias_max);
        code = (unsigned)bp.p;
    } else {                        // block_log.py:55-58, log.py:45-48
        bp.p = clampi(a.span - ceil_log2(bmax, lut), 0, a.bias_max);
        bp.eps = __builtin_ldexpf(0.1f, -bp.p);
        code = (unsigned)bp.p;
    }
    return bp;
}

// one element: returns the fake-quantised value, `mant` = signed integer mantissa (bfp only)
template <int FMT>
__device__ __forceinline__ float quant_elem(float x, const BlockParam& bp, const QuantArgs& a, const Lut& lut, int& mant) {
    const float ax = fabsf(x);
    if (FMT == FMT_BFP) {           // block_fp.py:69-82, 93-94
        const float s = sgn(x + EPS9);
        const float v = ax + EPS9;
        const float r = __builtin_ldexpf(v, -bp.p) * a.shift;
        const float m = clampf(__builtin_rintf(r), 0.f, a.mant_max);
        mant = (int)(s * m);
        const float q = __builtin_ldexpf(s, bp.p) * (m * a.inv_shift);
        return ax <= ATOL ? x + 0.0f : q;      // (+ 0: the reference's mask arithmetic turns -0.0 into +0.0)
    } else if (FMT == FMT_BM) {     // minifloat.py:165-194 with exponent_bias = bp.p
        const float s = sgn(x + EPS9);
        const int e_min = -bp.p, e_max = a.span - bp.p;
        const int e = clampi(floor_log2(ax + EPS9, lut), e_min, e_max);
        const float mn = __builtin_ldexpf(ax, -e);
        const bool normal = e != e_min;
        const float sm = normal ? clampf(__builtin_rintf(mn * a.shift - a.shift), 0.f, a.mant_max)
                                : clampf(__builtin_rintf(mn * a.shift * 0.5f), 0.f, a.mant_max);
        const float frac = normal ? (1.0f + sm * a.inv_shift) : (sm * a.inv_shift * 2.0f);
        const float q = __builtin_ldexpf(s, e) * frac;
        mant = 0;
        return ax <= ATOL ? x + 0.0f : q;      // (+ 0: the reference's mask arithmetic turns -0.0 into +0.0)
    } else {                        // log.py:47-56 with exponent_bias = bp.p
        const float s = sgn(x + bp.eps);
        const float v = ax + bp.eps;
        const int e_min = -bp.p, e_max = a.span - bp.p;
        const int r = v > 0.f ? rint_log2(v, lut) : e_min;
        mant = 0;
        return __builtin_ldexpf(s, clampi(r, e_min, e_max));
    }
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

// One block_minifloat element given its block's bias, biased-exponent bounds eminb = 127 - bias, emaxb = 127 + span - bias
// (minifloat.py:165-194 with exponent_bias = the block's), in ~18 VALU operations instead of quant_elem's ~40, same result:
//   * floor(log2(v)) of the normal v = |x| + 1e-9 is its exponent field, except within a few ulps below a power of two where
//     torch's fp32 log2 rounds up (the table's band, M >= FLOOR_THR_MIN): `near` reports those and the caller redoes the step
//     with quant_elem -- about one step in a thousand;
//   * mn * shift = ldexp(|x|, mbits - e) in one exact scaling (overflow gives inf either way, clamped to the top mantissa);
//   * 2^e (1 + sm / shift) = ldexp(shift + sm, e - mbits), 2^e (sm / shift) 2 = ldexp(2 sm, e - mbits): integers below 2^9.
__device__ __forceinline__ float bm_elem_fused(float x, int eminb, int emaxb, int mbits, float shift, float mant_max, bool& near) {
    const float ax = fabsf(x);
    const unsigned vb = __float_as_uint(ax + EPS9);
    near |= (vb | 0xFF800000u) >= (MI355Q_LOG2_FLOOR_THR_MIN | 0xFF800000u);
    const int eb = min(max((int)(vb >> 23), eminb), emaxb);
    const float t = __builtin_ldexpf(ax, mbits + 127 - eb);
    const float un = fminf(fmaxf(__builtin_rintf(t - shift), 0.f), mant_max) + shift;
    const float us = fminf(__builtin_rintf(t * 0.5f), mant_max) * 2.0f;
    const float q = __builtin_copysignf(__builtin_ldexpf(eb != eminb ? un : us, eb - 127 - mbits), x);
    return ax <= ATOL ? x : q;                  // (callers that hand the value on as a RESULT add + 0.0f: -0.0 -> +0.0)
}
// One block_log element (log.py:47-56 with exponent_bias = the block's; eps = 0.1 * 2^-bias): sign(x + eps) * 2^clamp(round(
// log2(|x| + eps))).  round(log2(v)) is the exponent field plus one when the fraction lies above sqrt(2)'s band; inside the
// band ([RND_LO_MIN, RND_HI_MAX], 125 fraction values where torch's fp32 log2 decides) and for subnormal v: `near`, as above.
__device__ __forceinline__ float bl_elem_fused(float x, float eps, int eminb, int emaxb, bool& near) {
    const float sg = x + eps;
    const unsigned vb = __float_as_uint(fabsf(x) + eps);
    near |= ((vb & 0x7FFFFFu) - MI355Q_LOG2_RND_LO_MIN) <= (unsigned)(MI355Q_LOG2_RND_HI_MAX - MI355Q_LOG2_RND_LO_MIN) || vb < 0x00800000u;
    const int rb = min(max((int)((vb + (0x7FFFFFu - MI355Q_LOG2_RND_HI_MAX)) >> 23), eminb), emaxb);
    return __builtin_ldexpf(__builtin_copysignf(sg != 0.f ? 1.0f : 0.0f, sg), rb - 127);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// abs-max over the LPB adjacent lanes that hold one block
template <int LPB>
__device__ __forceinline__ float group_max(float v) {
    if (LPB >= 2) v = fmaxf(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
    if (LPB >= 4) v = fmaxf(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
#pragma unroll
    for (int off = 4; off < LPB; off <<= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// two floats -> one dword of bf16 (v_cvt_pk_bf16_f32, round to nearest even).  Quantised values are exact in bf16; the
// only inexact inputs are elements |x| <= 1e-8, which the reference passes through unquantised (block_fp.py:93-94): they
// enter the product with a relative error of 2^-9 of themselves, at most 2e-11 absolute each.
// The elementwise step in front of a GEMM-operand quantiser (QuantArgs::pre_op), with the arithmetic of the torch kernels
// the reference's MLPs run there: relu = max(x, 0) (NaN kept), silu(x) * u = (x / (1 + exp(-x))) * u, each operation rounded
// to fp32 (no contraction: the product of the reference is a separate kernel).
__device__ __forceinline__ float pre_relu(float x) { return x > 0.f || x != x ? x : 0.f; }
__device__ __forceinline__ float pre_silu_mul(float x, float u) {
    float s = x / (1.0f + expf(-x));
    asm volatile("" : "+v"(s));                  // (keeps the product from being fused into the division's last step)
    return s * u;
}
__device__ __forceinline__ float4 apply_pre(const QuantArgs& a, float4 v, const float4* __restrict__ x2_4, long long idx) {
    if (a.pre_op == MI355Q_PRE_RELU) {
        v = make_float4(pre_relu(v.x), pre_relu(v.y), pre_relu(v.z), pre_relu(v.w));
    } else if (a.pre_op == MI355Q_PRE_SILU_MUL) {
        const float4 u = x2_4[idx];
        v = make_float4(pre_silu_mul(v.x, u.x), pre_silu_mul(v.y, u.y), pre_silu_mul(v.z, u.z), pre_silu_mul(v.w, u.w));
    }
    return v;
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

}  // namespace mi355q
#endif
