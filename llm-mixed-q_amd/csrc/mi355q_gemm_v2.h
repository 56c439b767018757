// mi355q_gemm_v2.h -- device body of GEMM variant 2 (exponent-aligned operands, int32 chains over
// flagged 256-deep K-groups, exact blockwise path for the others).  Included by mi355q_gemm.hip (its own
// kernel) and by mi355q_gemm_v3.hip (the fallback branch of the int32-chain kernel's launch).
#ifndef MI355Q_GEMM_V2_H
#define MI355Q_GEMM_V2_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q_internal.h"

namespace mi355q {

using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

constexpr int ALIGN_G = 16;   // blocks per alignment group (256 values)

// Tiled mantissa layout of an aligned operand: 1-KiB pieces of 16 rows x 64 K-bytes, piece index
// (row/16) * (K/64) + k/64; inside a piece the four 16-byte blocks of a row are 256 bytes apart -- the piece is
// [block 0..3][row 0..15][16 bytes].  This is the LDS image of the GEMM kernels (one global_load_lds copies one piece
// linearly); the ds_read_b128 fragment reads of both the 32x32x32 and the 16x16x64 MFMA (lane = row, block) touch every
// bank once per 16-lane group without any swizzle, and the 16 rows' blocks at one K position -- what the exception
// add-back gathers -- are 256 contiguous bytes.
__device__ __forceinline__ long long tiled_offset(long long row, long long k, long long K) {
    const long long piece = (row >> 4) * (K >> 6) + (k >> 6);
    return piece * 1024 + ((k >> 4) & 3) * 256 + (row & 15) * 16 + (k & 15);
}
// the same inside an LDS image of consecutive pieces of one K-step: row r, block c
__device__ __forceinline__ int piece_lds_off(int r, int c) { return (r >> 4) * 1024 + c * 256 + (r & 15) * 16; }

// ---------------------------------------------------------------------------------------
// Variant 2: 128 x 128 tile, 4 waves x (64 x 64), K-step 64 staged by global_load_lds (16 B/lane)
// into a double buffer, XCD-aware tile order.
// ---------------------------------------------------------------------------------------
constexpr int V2_BM = 128, V2_BN = 128, V2_BK = 64;

struct V2Smem {
    alignas(16) int8_t a[2][V2_BM * V2_BK];
    alignas(16) int8_t b[2][V2_BN * V2_BK];
    alignas(16) float ga[2][V2_BM];      // group scale per row, by group parity
    alignas(16) float gb[2][V2_BN];
    alignas(16) float pa[4][V2_BM];      // per-block scales of one K-step (blockwise path)
    alignas(16) float pb[4][V2_BN];
};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

__device__ __forceinline__ int v2_off(int r, int c) { return piece_lds_off(r, c); }

// XCD-aware tile order: blocks b, b+8, ... share an XCD (L2); each XCD gets a contiguous chunk of the grouped
// (8 tile-rows at a time) tile sequence.
__device__ __forceinline__ void v2_tile_origin(const GemmArgs& a, int tile_id, long long& m0, long long& n0) {
    const int tiles_m = (int)((a.M + V2_BM - 1) / V2_BM), tiles_n = (int)((a.N + V2_BN - 1) / V2_BN);
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int orig = tile_id, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int GM = 8, in_group = GM * tiles_n, group_id = pid / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (pid % in_group) % gsz, tn = (pid % in_group) / gsz;
    m0 = (long long)tm * V2_BM;
    n0 = (long long)tn * V2_BN;
}

// One 128 x 128 tile by 256 threads (`tid` = 0..255 within the team).  A workgroup may run several teams side by
// side on different tiles (each with its own V2Smem): the workgroup barriers inside are reached by every team the same
// number of times (same K), and the all-flagged vote then spans the teams, which only makes it more conservative.
__device__ __forceinline__ void bfp_gemm_v2_tile(const GemmArgs& a, const uint8_t* __restrict__ xf,
                                                 const uint8_t* __restrict__ wf, V2Smem& sm, long long m0, long long n0,
                                                 int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;

    const long long nkb = a.K >> 4;
    const int nsteps = (int)(a.K >> 6), ngroups = (int)((nkb + ALIGN_G - 1) / ALIGN_G);
    const int half_a = a.scale_bias >> 1, half_b = a.scale_bias - half_a;

    float acc[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the two 1-KiB pieces (tiled layout) this wave stages per operand per step
    long long srcA[2], srcB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = wave + 4 * t;
        srcA[t] = ((m0 >> 4) + j) * (a.K >> 6) * 1024 + lane * 16;
        srcB[t] = ((n0 >> 4) + j) * (a.K >> 6) * 1024 + lane * 16;
    }
    auto stage = [&](int step, int buf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = wave + 4 * t;
            __builtin_amdgcn_global_load_lds((gptr_t)(a.xm + srcA[t] + (long long)step * 1024),
                                             (lptr_t)(&sm.a[buf][j * 1024]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(a.wm + srcB[t] + (long long)step * 1024),
                                             (lptr_t)(&sm.b[buf][j * 1024]), 16, 0, 0);
        }
    };

    stage(0, 0);
    int cur = 0, step = 0;
    const bool is_a = tid < 128;
    const int sr = tid & 127;
    const long long srow = is_a ? min(m0 + sr, a.M - 1) : min(n0 + sr, a.N - 1);
    const uint8_t* __restrict__ fl = is_a ? xf : wf;
    const uint8_t *x_e = a.xe, *w_e = a.we;     // (read both, then select: a conditional over the fields is a select of addresses in `a`)
    const uint8_t* __restrict__ ex = is_a ? x_e : w_e;
    const int sh = is_a ? half_a : half_b;

    for (int g = 0; g < ngroups; ++g) {
        const int gs = min(4, nsteps - 4 * g);
        const int f = fl[a.row_mode ? srow : srow * ngroups + g];
        const float gsc = __builtin_ldexpf(1.0f, (int)ex[srow * nkb + (long long)g * ALIGN_G] - sh);
        if (is_a) sm.ga[g & 1][sr] = gsc; else sm.gb[g & 1][sr] = gsc;
        const int fast = __syncthreads_and(f);

        if (fast) {
            i32x16 ci[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ci[i][j][r] = 0;
            for (int s = 0; s < gs; ++s, ++step) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (step + 1 < nsteps) stage(step + 1, cur ^ 1);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    i32x4 fa[2], fb[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int ra = wm * 64 + i * 32 + lr, rb = wn * 64 + i * 32 + lr;
                        fa[i] = *reinterpret_cast<const i32x4*>(&sm.a[cur][v2_off(ra, 2 * p + lh)]);
                        fb[i] = *reinterpret_cast<const i32x4*>(&sm.b[cur][v2_off(rb, 2 * p + lh)]);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
                }
                cur ^= 1;
            }
            // one rescale per group
            float sw[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) sw[j] = sm.gb[g & 1][wn * 64 + j * 32 + lr];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(&sm.ga[g & 1][wm * 64 + i * 32 + 8 * q + 4 * lh]);
                    const float sx[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[i][j][4 * q + r] += (float)ci[i][j][4 * q + r] * sx[r] * sw[j];
                }
            }
        } else {
            for (int s = 0; s < gs; ++s, ++step) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (step + 1 < nsteps) stage(step + 1, cur ^ 1);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const float sc = __builtin_ldexpf(1.0f, (int)ex[srow * nkb + (long long)step * 4 + kb] - sh);
                    if (is_a) sm.pa[kb][sr] = sc; else sm.pb[kb][sr] = sc;
                }
                __syncthreads();
#pragma unroll 1
                for (int kb = 0; kb < 4; ++kb) {
                    long fa[2], fb[2];
                    float sw[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int ra = wm * 64 + i * 32 + lr, rb = wn * 64 + i * 32 + lr;
                        fa[i] = *reinterpret_cast<const long*>(&sm.a[cur][v2_off(ra, kb) + lh * 8]);
                        fb[i] = *reinterpret_cast<const long*>(&sm.b[cur][v2_off(rb, kb) + lh * 8]);
                        sw[i] = sm.pb[kb][rb];
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float sx[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = *reinterpret_cast<const float4*>(&sm.pa[kb][wm * 64 + i * 32 + 8 * q + 4 * lh]);
                            sx[4 * q + 0] = v.x; sx[4 * q + 1] = v.y; sx[4 * q + 2] = v.z; sx[4 * q + 3] = v.w;
                        }
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            i32x16 z = {0};
                            const i32x16 d = __builtin_amdgcn_mfma_i32_32x32x16_i8(fa[i], fb[j], z, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] += (float)d[r] * sx[r] * sw[j];
                        }
                    }
                }
                cur ^= 1;
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long col = n0 + wn * 64 + j * 32 + lr;
            if (col >= a.N) continue;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                // (a.resid: the caller's residual -- the mixed contraction's class-1 product, mi355q_gemm_v9.hip -- in the store)
                if (row < a.M) a.y[row * a.ldy + col] = acc[i][j][r] + bv + (a.resid ? a.resid[row * a.ldr + col] : 0.f);
            }
        }
}

__device__ __forceinline__ void bfp_gemm_v2_body(const GemmArgs& a, const uint8_t* __restrict__ xf,
                                                 const uint8_t* __restrict__ wf, V2Smem& sm, int tile_id) {
    long long m0, n0;
    v2_tile_origin(a, tile_id, m0, n0);
    bfp_gemm_v2_tile(a, xf, wf, sm, m0, n0, threadIdx.x);
}

}  // namespace mi355q
#endif
