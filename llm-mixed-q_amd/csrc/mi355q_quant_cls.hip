// mi355q_quant_cls.hip -- the class-aware activation quantiser of the MIXED contraction (round 6; mi355q_gemm_v9m.hip,
// include/mi355q.h mi355q_block_fp_quantize_classes).
//
// One pass over x [rows, K] fp32 (block_fp, [1,16] blocks along K: quantizers/block_fp.py:21-96 through utils.py:127-144): every
// block is quantised exactly as mi355q_block_fp_quantize_aligned_rows does (same tables, same arithmetic: the two kernels share
// mi355q_quant_dev.h and mi355q_align_row.h), then goes where the caller's COLUMN MAP sends its block column:
//   class 0 -> the row-aligned int8 operand [rows, 16 n0] (tiled mantissas, effective exponents, row flag / scale, bucketed
//              exception list) at block position p: the row's exponent is decided over its class-0 blocks ONLY;
//   class 1 -> the tiled bf16 operand [rows, 16 n1] at block position p, every block with its own exponent (value = mantissa x
//              2^(e - mbits): exact in bf16 for widths <= 8).
// map[kb] = p | (class << 15).  The split is the caller's (quantized_modules/linear.py: F.linear sums over in_features in any
// order); what it buys: activations with outlier channels (README.md:9-11 of the reference) keep every block column WITHOUT such a
// channel on the int8 MFMA.  HBM-bound: 4 B read, 1 + 1/16 B (class 0) or 2 B (class 1) written per element.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q_quant_dev.h"
#include "mi355q_align_row.h"

namespace mi355q {

template <int MAXIT, bool FULL, int WPS>
__global__ __launch_bounds__(256, WPS) void bfp_quant_classes_kernel(const QuantArgs a, const uint16_t* __restrict__ cmap, int n0, int n1,
                                                                int8_t* __restrict__ mt, uint8_t* __restrict__ flag,
                                                                float* __restrict__ rscale, int exp_offset, int* __restrict__ list,
                                                                int* __restrict__ list_to_clear, int bcap, uint16_t* __restrict__ bt) {
    __shared__ Lut lut;
    __shared__ RowAlignSmem rsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long K = a.cols;
    const int nkb = (int)(K >> 4), nit = (nkb + 63) >> 6;
    const long long rows16 = a.rows & ~127ll;
    // (workgroup -> row as in bfp_quant_align_rows_kernel: the 16 rows of a piece row on workgroups that share an XCD)
    auto row_of = [&](long long wi) {
        if (wi >= rows16) return wi;
        const long long grp = wi >> 7, in = wi & 127;
        return (grp << 7) + ((in & 7) << 4) + (in >> 3);
    };
    auto valid = [&](int it) { return FULL || (it < nit && it * 64 + wave * 16 + (lane >> 2) < nkb); };     // (FULL: K = 1024 MAXIT, no guards)
    auto load_raw = [&](float4 (&v)[MAXIT], long long row) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
            v[it] = valid(it) ? reinterpret_cast<const float4*>(a.x)[(unsigned)row * (unsigned)(K >> 2) + (unsigned)(it * 256 + tid)] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 v[MAXIT];
    if ((long long)blockIdx.x < a.rows) load_raw(v, row_of(blockIdx.x));
    // the lane's block columns never change: their classes and positions once, in registers
    int pos[MAXIT];                                            // position | class << 15, decoded where it is used
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) pos[it] = valid(it) ? (int)cmap[it * 64 + wave * 16 + (lane >> 2)] : 0;
#define CLS_C1(it) ((pos[it] >> 15) != 0)
    load_lut<FMT_BFP>(lut);
    if (list_to_clear && blockIdx.x == 0) {
        const int cb = bcap < 0 ? ROW_BCAP : bcap;
        const long long words = row_list_words(a.rows, cb), bw = row_bucket_words(cb);
        if (tid < EXC_HEADER) list_to_clear[tid] = 0;
        for (long long b = EXC_HEADER + (long long)tid * bw; b < words; b += 256ll * bw) {
            list_to_clear[b] = 0;
            list_to_clear[b + 1] = 0;
        }
    }
    const int mbits_int = (int)__builtin_log2f(a.shift);
    const long long kp0 = n0 >> 2, kp1 = n1 >> 1;             // 1-KiB pieces per 16 rows of the two operands
    __syncthreads();
    for (long long wi = blockIdx.x; wi < a.rows; wi += gridDim.x) {
        const long long row = row_of(wi);
        unsigned pk[MAXIT];
        int amax[MAXIT], code[MAXIT], up[MAXIT];
        unsigned bmb[MAXIT];
        bool big = false;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const unsigned b0 = __float_as_uint(v[it].x) & 0x7FFFFFFFu, b1 = __float_as_uint(v[it].y) & 0x7FFFFFFFu;
            const unsigned b2 = __float_as_uint(v[it].z) & 0x7FFFFFFFu, b3 = __float_as_uint(v[it].w) & 0x7FFFFFFFu;
            unsigned m = max(max(b0, b1), max(b2, b3));
            m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
            bmb[it] = m;
            const float bm1 = m != 0u ? __uint_as_float(m) : 1.0f;     // all-zero block: fill 1 (MI355Q_ZERO_BLOCK_FAST)
            const int k = __builtin_amdgcn_frexp_expf(bm1) - 1;
            const unsigned f = __float_as_uint(__builtin_amdgcn_frexp_mantf(bm1)) & 0x7FFFFFu;
            const unsigned thr = lut.a[lut_index(k)];
            const int e = clampi(k + ((f != 0u && f >= thr) ? 1 : 0), a.e_min, a.e_max);
            up[it] = mbits_int - e;
            code[it] = e + a.code_bias;
            big = big || up[it] >= 28;
        }
        if (__any(big)) {
            // some block of the row lies below 2^-23: the exact rule for the whole row (block_fp.py:69)
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                auto mf = [&](float x) {
                    const float t = x + EPS9;
                    const float m = fminf(__builtin_rintf(__builtin_ldexpf(fabsf(x) + EPS9, up[it])), a.mant_max);
                    return t == 0.f ? 0 : (int)__builtin_copysignf(m, t);
                };
                const int q0 = mf(v[it].x), q1 = mf(v[it].y), q2 = mf(v[it].z), q3 = mf(v[it].w);
                const int am = (int)group_max<4>((float)max(max(abs(q0), abs(q1)), max(abs(q2), abs(q3))));
                amax[it] = bmb[it] != 0u ? am : 0;
                const unsigned lo = __builtin_amdgcn_perm((unsigned)q1, (unsigned)q0, 0x0c0c0400u);
                const unsigned hi = __builtin_amdgcn_perm((unsigned)q3, (unsigned)q2, 0x04000c0cu);
                pk[it] = lo | hi;
            }
        } else {
            // (the fast path of bfp_quant_align_rows_kernel: rne(clamp(fma(x, 2^up, copysign(1e-9 2^up, x)))) by the fp32 add of
            //  1.5 * 2^23, the two's-complement mantissa in the low byte of the sum's bit pattern)
            constexpr float MAGIC = 12582912.0f;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const float sc = __builtin_ldexpf(1.0f, up[it]);
                const float es = EPS9 * sc;
                auto mant_bits = [&](float x) {
                    const float r = __builtin_fmaf(x, sc, __builtin_copysignf(es, x));
                    return __float_as_uint(__builtin_amdgcn_fmed3f(r, -a.mant_max, a.mant_max) + MAGIC);
                };
                const unsigned t0 = mant_bits(v[it].x), t1 = mant_bits(v[it].y), t2 = mant_bits(v[it].z), t3 = mant_bits(v[it].w);
                amax[it] = (int)(mant_bits(__uint_as_float(bmb[it])) - 0x4B400000u);
                if (bmb[it] == 0u) amax[it] = 0;
                const unsigned lo = __builtin_amdgcn_perm(t1, t0, 0x0c0c0400u);
                const unsigned hi = __builtin_amdgcn_perm(t3, t2, 0x04000c0cu);
                pk[it] = lo | hi;
            }
        }
        if (wi + gridDim.x < a.rows) load_raw(v, row_of(wi + gridDim.x));          // (the next row's loads behind the mantissas)
        // WIDE (K = 4096, every lane a block in every slab): behind the row decision a 4 x 4 transpose inside the quad leaves lane q
        // with the WHOLE block of slab q (bfp_quant_align_rows_kernel, ST16) -- one 16-byte store for a class-0 block, two for a
        // class-1 block (its sixteen bf16 values) instead of four dword + up to four 8-byte stores a lane
        constexpr bool WIDE = FULL && MAXIT == 4;
        // class 1 first: the block's values as bf16 (mantissa x 2^(e - mbits), exact), 8 bytes a lane; a piece of the bf16
        // operand is 16 rows x 32 values = two blocks, [8-value group 0..3][row][16 bytes]
        if constexpr (!WIDE)
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if (valid(it) && CLS_C1(it)) {
                const int e = code[it] - a.code_bias;
                const float s = __builtin_ldexpf(1.0f, e - mbits_int);
                const float f0 = (float)(int)(signed char)(pk[it]), f1 = (float)(int)(signed char)(pk[it] >> 8);
                const float f2 = (float)(int)(signed char)(pk[it] >> 16), f3 = (float)(int)(signed char)(pk[it] >> 24);
                const unsigned h0 = __float_as_uint(f0 * s) >> 16, h1 = __float_as_uint(f1 * s) >> 16;       // (exact: <= 7 significant bits)
                const unsigned h2 = __float_as_uint(f2 * s) >> 16, h3 = __float_as_uint(f3 * s) >> 16;
                const int p = pos[it] & 0x7FFF, q = lane & 3;
                // (32-bit offsets from the uniform base: rows * K < 2^31, checked by the launcher)
                const unsigned o = (((unsigned)row >> 4) * (unsigned)kp1 + ((unsigned)p >> 1)) * 1024u + (((unsigned)p & 1u) * 2u + ((unsigned)q >> 1)) * 256u +
                                   ((unsigned)row & 15u) * 16u + ((unsigned)q & 1u) * 8u;
                *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(bt) + o) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
            }
        }
        // class 0: the row's exponent over ITS blocks (a class-1 block takes no part: largest mantissa 0 = "no block here")
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) amax[it] = CLS_C1(it) ? 0 : amax[it];
        int E = 0;
        // (an exception entry records pos[it]: for a class-0 block its position in the int8 operand -- the class bit is clear)
        const bool flagged = bcap < 0 ? false : align_row_impl<MAXIT, FULL, true>(pk, amax, code, nit, nkb, row, list, rsm, E, bcap, pos);
        if constexpr (WIDE) {
            const int q = lane & 3;
            unsigned r0 = pk[0], r1 = pk[1], r2 = pk[2], r3 = pk[3];
            {
                const bool odd = q & 1;
                const unsigned s01 = odd ? r0 : r1, s23 = odd ? r2 : r3;
                const unsigned g01 = (unsigned)__builtin_amdgcn_mov_dpp((int)s01, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
                const unsigned g23 = (unsigned)__builtin_amdgcn_mov_dpp((int)s23, 0xB1, 0xF, 0xF, true);
                if (odd) { r0 = g01; r2 = g23; } else { r1 = g01; r3 = g23; }
            }
            {
                const bool hi = q & 2;
                const unsigned s02 = hi ? r0 : r2, s13 = hi ? r1 : r3;
                const unsigned g02 = (unsigned)__builtin_amdgcn_mov_dpp((int)s02, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
                const unsigned g13 = (unsigned)__builtin_amdgcn_mov_dpp((int)s13, 0x4E, 0xF, 0xF, true);
                if (hi) { r0 = g02; r1 = g13; } else { r2 = g02; r3 = g13; }
            }
            // lane q: bytes 0..15 of the block of slab q; its map word and exponent (the same in all four lanes of the quad)
            const int pq = q == 0 ? pos[0] : q == 1 ? pos[1] : q == 2 ? pos[2] : pos[3];
            const int cq = q == 0 ? code[0] : q == 1 ? code[1] : q == 2 ? code[2] : code[3];
            const int p = pq & 0x7FFF;
            if ((pq >> 15) == 0) {
                const unsigned o = (((unsigned)row >> 4) * (unsigned)kp0 + ((unsigned)p >> 2)) * 1024u + ((unsigned)p & 3u) * 256u + ((unsigned)row & 15u) * 16u;
                *reinterpret_cast<uint4*>(mt + o) = make_uint4(r0, r1, r2, r3);
                a.code[(unsigned)row * (unsigned)n0 + (unsigned)p] = (uint8_t)(flagged ? E : cq);
            } else {
                const float sc1 = __builtin_ldexpf(1.0f, cq - a.code_bias - mbits_int);
                auto two = [&](unsigned w, int b) {          // bytes b, b + 1 of w as two bf16 (exact: <= 7 significant bits)
                    const float lo = (float)(int)(signed char)(w >> (8 * b)) * sc1, hi2 = (float)(int)(signed char)(w >> (8 * b + 8)) * sc1;
                    return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi2) & 0xFFFF0000u);
                };
                const unsigned o = (((unsigned)row >> 4) * (unsigned)kp1 + ((unsigned)p >> 1)) * 1024u + ((unsigned)p & 1u) * 512u + ((unsigned)row & 15u) * 16u;
                uint8_t* d = reinterpret_cast<uint8_t*>(bt) + o;
                *reinterpret_cast<uint4*>(d) = make_uint4(two(r0, 0), two(r0, 2), two(r1, 0), two(r1, 2));              // values 0..7: group 2 (p & 1)
                *reinterpret_cast<uint4*>(d + 256) = make_uint4(two(r2, 0), two(r2, 2), two(r3, 0), two(r3, 2));        // values 8..15: the next group
            }
        } else
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if (valid(it) && !CLS_C1(it)) {
                const int p = pos[it];
                const unsigned o = (((unsigned)row >> 4) * (unsigned)kp0 + ((unsigned)p >> 2)) * 1024u + ((unsigned)p & 3u) * 256u + ((unsigned)row & 15u) * 16u +
                                   ((unsigned)lane & 3u) * 4u;
                *reinterpret_cast<unsigned*>(mt + o) = pk[it];
                if ((lane & 3) == 0) a.code[(unsigned)row * (unsigned)n0 + (unsigned)p] = (uint8_t)(flagged ? E : code[it]);
            }
        }
        if (tid == 0) {
            flag[row] = flagged ? 1 : 0;
            rscale[row] = flagged ? __builtin_ldexpf(1.0f, E - exp_offset) : 0.0f;
        }
        if (wi + gridDim.x < a.rows) __syncthreads();       // (the next row reuses the decision words in LDS)
    }
}

int launch_quant_classes(const QuantArgs& a, const uint16_t* cmap, int n0, int n1, int8_t* mt, uint8_t* flag, float* rscale,
                         int exp_offset, int* list, int* list_to_clear, uint16_t* bt, hipStream_t st, int bcap) {
    if (a.rows * a.cols >= (1ll << 30)) return MI355Q_E_UNSUPPORTED;       // (32-bit byte offsets inside the kernel)
    long long grid = a.rows;
    const long long cap = a.cols == 4096 ? 1536 : 1024;          // (K = 4096: 75 registers, six workgroups a compute unit)
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
#define MI355Q_LAUNCH_CLS(MAXIT_, FULL_, WPS_)                                                                        \
    hipLaunchKernelGGL((bfp_quant_classes_kernel<MAXIT_, FULL_, WPS_>), (unsigned)grid, 256, 0, st, a, cmap, n0, n1, mt, flag, rscale, \
                       exp_offset, list, list_to_clear, bcap, bt)
    if (a.cols == 4096) MI355Q_LAUNCH_CLS(4, true, 1);
    else if (a.cols <= 4096) MI355Q_LAUNCH_CLS(4, false, 1);
    else if (a.cols <= 8192) MI355Q_LAUNCH_CLS(8, false, 1);
    else if (a.cols <= 16384) MI355Q_LAUNCH_CLS(16, false, 1);
    else
        return MI355Q_E_UNSUPPORTED;
#undef MI355Q_LAUNCH_CLS
    return (int)hipGetLastError();
}
#undef CLS_C1

}  // namespace mi355q
