// mi355q_gemm.hip -- block-floating-point GEMM on int8 MFMA for gfx950 (MI355X).
//
//   y[m,n] = sum_kb 2^(xe[m,kb] + we[n,kb] - scale_bias) * ( sum_{j<16} xm[m,16kb+j] * wm[n,16kb+j] ) + bias[n]
//
// replaces F.linear(x_q, W_q, b_q) of the reference's PTQ LinearBlockFP
// (quantized_modules/linear.py:59-76) for [1,16] blocks along in_features.
// The inner 16-term dot is an exact int8 x int8 -> int32 MFMA product; the sum over K-blocks is
// carried in fp32 (as the reference's SGEMM does, in a different order).
//
// Variant 1 ("blockwise"): one v_mfma_i32_32x32x16_i8 per 16-wide K-block, each int32 tile
// rescaled by its own 2^(xe+we) into fp32 accumulators.  Exact for any exponent pattern; VALU
// bound (48 VALU per MFMA).  It is the correctness anchor the faster variants are tested against.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_internal.h"

namespace mi355q {

using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;

// ---------------------------------------------------------------------------------------
// Variant 1
//   workgroup tile 128 x 128, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32
//   K-step 64 (four 16-blocks), operands staged global -> VGPR -> LDS, one barrier pair per step
//   LDS per step: A 128x64 B + B 128x64 B + scales 2 x 4 x 128 floats = 20 KiB
// ---------------------------------------------------------------------------------------
constexpr int V1_BM = 128, V1_BN = 128, V1_BK = 64, V1_KB = V1_BK / 16;

struct V1Smem {
    alignas(16) int8_t a[V1_BM * V1_BK];
    alignas(16) int8_t b[V1_BN * V1_BK];
    alignas(16) float sa[V1_KB][V1_BM];   // 2^(xe - half of scale_bias)
    alignas(16) float sb[V1_KB][V1_BN];
};

// 16-byte chunk c (0..3) of row r lives at chunk slot c ^ ((r >> 2) & 3): spreads the 64-byte rows
// over the banks for the 8-byte fragment reads
__device__ __forceinline__ int v1_off(int r, int c) { return r * V1_BK + ((c ^ ((r >> 2) & 3)) << 4); }

__global__ __launch_bounds__(256) void bfp_gemm_v1(const GemmArgs a) {
    __shared__ V1Smem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long long m0 = (long long)blockIdx.y * V1_BM, n0 = (long long)blockIdx.x * V1_BN;
    const long long nkb = a.K >> 4;
    const int half_a = a.scale_bias >> 1, half_b = a.scale_bias - half_a;

    float acc[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;

    for (long long k0 = 0; k0 < a.K; k0 += V1_BK) {
        // ---- stage: 128 rows x 64 B per operand = 512 chunks of 16 B -> 2 per thread per operand
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int idx = tid + t * 256;
            const int r = idx >> 2, c = idx & 3;
            int4 va = make_int4(0, 0, 0, 0), vb = make_int4(0, 0, 0, 0);
            if (m0 + r < a.M && k0 + c * 16 < a.K)
                va = *reinterpret_cast<const int4*>(a.xm + (m0 + r) * a.K + k0 + c * 16);
            if (n0 + r < a.N && k0 + c * 16 < a.K)
                vb = *reinterpret_cast<const int4*>(a.wm + (n0 + r) * a.K + k0 + c * 16);
            *reinterpret_cast<int4*>(&sm.a[v1_off(r, c)]) = va;
            *reinterpret_cast<int4*>(&sm.b[v1_off(r, c)]) = vb;
        }
        // scales: 4 x 128 per operand -> 2 per thread per operand
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int idx = tid + t * 256;
            const int r = idx & 127, kb = idx >> 7;
            const long long gkb = (k0 >> 4) + kb;
            float fa = 0.f, fb = 0.f;
            if (m0 + r < a.M && gkb < nkb) fa = __builtin_ldexpf(1.0f, (int)a.xe[(m0 + r) * nkb + gkb] - half_a);
            if (n0 + r < a.N && gkb < nkb) fb = __builtin_ldexpf(1.0f, (int)a.we[(n0 + r) * nkb + gkb] - half_b);
            sm.sa[kb][r] = fa;
            sm.sb[kb][r] = fb;
        }
        __syncthreads();

#pragma unroll 1
        for (int kb = 0; kb < V1_KB; ++kb) {
            long fa[2], fb[2];
            float sw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + lr, rb = wn * 64 + i * 32 + lr;
                fa[i] = *reinterpret_cast<const long*>(&sm.a[v1_off(ra, kb) + lh * 8]);
                fb[i] = *reinterpret_cast<const long*>(&sm.b[v1_off(rb, kb) + lh * 8]);
                sw[i] = sm.sb[kb][rb];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // rows of this lane's 16 results: 8*(r>>2) + 4*lh + (r&3)
                float sx[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(&sm.sa[kb][wm * 64 + i * 32 + 8 * g + 4 * lh]);
                    sx[4 * g + 0] = v.x; sx[4 * g + 1] = v.y; sx[4 * g + 2] = v.z; sx[4 * g + 3] = v.w;
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
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane & 31, row = 8*(r>>2) + 4*(lane>>5) + (r&3)
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
                if (row < a.M) a.y[row * a.ldy + col] = acc[i][j][r] + bv;
            }
        }
}


// =======================================================================================
// Exponent-aligned operands ("aligned format")
//   K is cut into groups of ALIGN_G = 16 blocks (256 values).  For a row and a group, if every
//   block's mantissas can be shifted left onto the group's smallest exponent without leaving int8
//   (5-bit W6 mantissas have 2 spare bits, W4 has 4), the row-group is stored shifted with ONE
//   effective exponent and flagged 1; otherwise it is stored unchanged and flagged 0.
//   (mant', eff_exp) denote exactly the same values as (mant, exp) -- variant 1 accepts them too.
//   Variant 2 runs an int32 MFMA chain over a whole group (8 x v_mfma_i32_32x32x32_i8 per 32x32
//   tile) when all 128 + 128 rows of the workgroup tile are flagged, and rescales ONCE per group;
//   any other group takes the exact blockwise path.  Which path runs never changes the value.
// =======================================================================================
constexpr int ALIGN_G = 16;

// Tiled mantissa layout of an aligned operand: 1-KiB pieces of 16 rows x 64 K-bytes, piece index
// (row/16) * (K/64) + k/64; inside a piece row r's 16-byte chunk c sits in slot c ^ ((row >> 2) & 3)
// -- the LDS image of the GEMM kernels, so that one global_load_lds copies one piece linearly.
__device__ __forceinline__ long long tiled_offset(long long row, long long k, long long K) {
    const long long piece = (row >> 4) * (K >> 6) + (k >> 6);
    const int chunk = (int)((k >> 4) & 3), slot = chunk ^ (int)((row >> 2) & 3);
    return piece * 1024 + (row & 15) * 64 + slot * 16 + (k & 15);
}

__global__ __launch_bounds__(256) void bfp_align_kernel(const int8_t* __restrict__ mi, const uint8_t* __restrict__ ei,
                                                        int8_t* __restrict__ mo, uint8_t* __restrict__ eo,
                                                        uint8_t* __restrict__ flag, float* __restrict__ gscale,
                                                        long long rows_pad, int exp_offset, int* __restrict__ list,
                                                        int list_cap, int8_t* __restrict__ mt, long long rows,
                                                        long long K) {
    const long long nkb = K >> 4, ngroups = (nkb + ALIGN_G - 1) / ALIGN_G;
    const int lane = threadIdx.x & 63;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long pair = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; pair < rows * ngroups; pair += nwaves) {
        const long long row = pair / ngroups, g = pair - row * ngroups;
        const long long kb = g * ALIGN_G + (lane >> 2);
        const bool valid = kb < nkb;
        const long long moff = row * K + kb * 16 + (lane & 3) * 4;
        const unsigned v = valid ? *reinterpret_cast<const unsigned*>(mi + moff) : 0u;
        const int b0 = (int)(int8_t)(v & 0xFF), b1 = (int)(int8_t)((v >> 8) & 0xFF);
        const int b2 = (int)(int8_t)((v >> 16) & 0xFF), b3 = (int)(int8_t)(v >> 24);
        int amax = max(max(abs(b0), abs(b1)), max(abs(b2), abs(b3)));
        amax = max(amax, __shfl_xor(amax, 1));
        amax = max(amax, __shfl_xor(amax, 2));
        const int e = valid ? (int)ei[row * nkb + kb] : 0;
        const bool nz = amax > 0;
        int emin = nz ? e : (1 << 20);
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) emin = min(emin, __shfl_xor(emin, off));
        const int s = nz ? e - emin : 0;
        const bool ok = !nz || (s <= 7 && (amax << s) <= 127);
        const bool all_ok = __all(ok);
        unsigned out = v;
        int eout = e;
        if (all_ok) {
            out = (unsigned)((b0 << s) & 0xFF) | ((unsigned)((b1 << s) & 0xFF) << 8) |
                  ((unsigned)((b2 << s) & 0xFF) << 16) | ((unsigned)((b3 << s) & 0xFF) << 24);
            eout = emin == (1 << 20) ? e : emin;
        }
        if (valid) {
            if (mo) *reinterpret_cast<unsigned*>(mo + moff) = out;
            if ((lane & 3) == 0) eo[row * nkb + kb] = (uint8_t)eout;
            if (mt) *reinterpret_cast<unsigned*>(mt + tiled_offset(row, kb * 16 + (lane & 3) * 4, K)) = out;
        }
        if (lane == 0) {
            flag[row * ngroups + g] = all_ok ? 1 : 0;
            // fast-GEMM view: one fp32 scale per (group, row); 0 neutralises a row-group that could not
            // be aligned (its exact contribution comes from the sparse correction kernel)
            if (gscale) gscale[g * rows_pad + row] = all_ok ? __builtin_ldexpf(1.0f, eout - exp_offset) : 0.0f;
            if (list && !all_ok) {
                const int at = atomicAdd(&list[0], 1);
                if (at < list_cap) { list[2 + 2 * at] = (int)row; list[3 + 2 * at] = (int)g; }
            }
        }
    }
}

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

__device__ __forceinline__ int v2_off(int r, int c) { return r * V2_BK + ((c ^ ((r >> 2) & 3)) << 4); }

__global__ __launch_bounds__(256, 2) void bfp_gemm_v2(const GemmArgs a, const uint8_t* __restrict__ xf,
                                                      const uint8_t* __restrict__ wf, const int* __restrict__ xlist,
                                                      const int* __restrict__ wlist, int list_cap) {
    __shared__ V2Smem sm;
    // as the fallback of the int32-chain kernel: run only when the unaligned lists overflowed
    if (xlist && xlist[0] <= list_cap && wlist[0] <= list_cap) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;

    // ---- XCD-aware tile order: blocks b, b+8, ... share an XCD (L2); give each XCD a contiguous
    //      chunk of the grouped (8 tile-rows at a time) tile sequence
    const int tiles_m = (int)((a.M + V2_BM - 1) / V2_BM), tiles_n = (int)((a.N + V2_BN - 1) / V2_BN);
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int GM = 8, in_group = GM * tiles_n, group_id = pid / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (pid % in_group) % gsz, tn = (pid % in_group) / gsz;
    const long long m0 = (long long)tm * V2_BM, n0 = (long long)tn * V2_BN;

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
    const uint8_t* __restrict__ ex = is_a ? a.xe : a.we;
    const int sh = is_a ? half_a : half_b;

    for (int g = 0; g < ngroups; ++g) {
        const int gs = min(4, nsteps - 4 * g);
        const int f = fl[srow * ngroups + g];
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
                if (row < a.M) a.y[row * a.ldy + col] = acc[i][j][r] + bv;
            }
        }
}

int launch_bfp_align(const int8_t* mi, const uint8_t* ei, int8_t* mo, uint8_t* eo, uint8_t* flag, float* gscale,
                     long long rows_pad, int exp_offset, int* list, int list_cap, int8_t* mt, long long rows,
                     long long K, hipStream_t st) {
    const long long ngroups = ((K >> 4) + ALIGN_G - 1) / ALIGN_G;
    long long grid = (rows * ngroups + 3) / 4;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    if (list) {
        const hipError_t e = hipMemsetAsync(list, 0, 16, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(bfp_align_kernel, (unsigned)grid, 256, 0, st, mi, ei, mo, eo, flag, gscale, rows_pad, exp_offset,
                       list, list_cap, mt, rows, K);
    return (int)hipGetLastError();
}

int launch_bfp_gemm_aligned(const GemmArgs& a, const uint8_t* xf, const uint8_t* wf, const int* xlist,
                            const int* wlist, int list_cap, hipStream_t st) {
    const unsigned tiles = (unsigned)(((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN));
    hipLaunchKernelGGL(bfp_gemm_v2, tiles, 256, 0, st, a, xf, wf, xlist, wlist, list_cap);
    return (int)hipGetLastError();
}

int launch_bfp_gemm(const GemmArgs& a, int variant, hipStream_t st) {
    (void)variant;
    dim3 grid((unsigned)((a.N + V1_BN - 1) / V1_BN), (unsigned)((a.M + V1_BM - 1) / V1_BM));
    hipLaunchKernelGGL(bfp_gemm_v1, grid, 256, 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace mi355q
