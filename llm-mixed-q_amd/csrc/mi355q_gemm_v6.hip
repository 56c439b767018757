// mi355q_gemm_v6.hip -- block-floating-point GEMM over GROUP-aligned operands (gfx950): int32 MFMA chains over
// exponent-aligned K-groups of 256 values.
//
//   y[m,n] = sum_g  gx[g][m] * gw[g][n] * ( sum_{k in group g} xm'[m,k] * wm'[n,k] )   (+ bias[n])
//
// xm'/wm' are the exponent-aligned int8 mantissas written by the align step (mi355q_align.h), gx/gw the per
// (group,row) scales 2^(effective exponent - bias) as fp32, group-major ([K/256][rows padded to 256]).  Blocks
// the align step took out (exceptions) are zero here; the tail launch adds them back exactly.  No data-dependent
// branch in the main kernel.
//
// Workgroup = 256 x 128 outputs, 8 waves x (64 x 64), v_mfma_i32_16x16x64_i8.  K-step 128 per barrier (two
// 64-wide sub-steps per stage, three 48-KiB stages filled by global_load_lds two steps ahead, counted
// s_waitcnt vmcnt, one raw s_barrier per step).  The int32 tile of a group is folded into the fp32 accumulators
// once per group with the magic-constant trick: the MFMA chain starts from C = 0x4B400000 in every lane, so the
// int32 result D read AS FLOAT is 12582912 + D exactly (|D| < 2^22 over a group), and
//     u   = pk_fma(f, gx, -12582912 * gx)      ( = D * gx, exact: gx is a power of two )
//     acc = pk_fma(u, gw, acc)
// Group scales arrive by LDS-DMA too (no VGPR loads in the loop, so the compiler never drains the pipeline).
// The row-aligned flavour of the same GEMM (no rescale in the loop, 256 x 256 tile) is mi355q_gemm_v8.hip.
// Roofline: int8 MFMA (2 * M * N * K ops); HBM traffic is the operands once per tile pass through L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_gemm_v2.h"
#include "mi355q_fix.h"

namespace mi355q {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int V5_BM = 256, V5_BN = 128;
constexpr int V5_A_BYTES = V5_BM * 64, V5_B_BYTES = V5_BN * 64, V5_STAGE = V5_A_BYTES + V5_B_BYTES;
constexpr int V5_MAGIC_I = 0x4B400000;
constexpr float V5_MAGIC_F = 12582912.0f;
constexpr int V6_S = 3, V6_SUB = 2;
constexpr int V6_STAGE = V6_SUB * V5_STAGE;          // [sub][A 16 KiB | B 8 KiB]
constexpr int V6_GA = V6_S * V6_STAGE, V6_GB = V6_GA + 2 * 256 * 4, V6_LDS = V6_GB + 2 * 256 * 4;
static_assert(V6_LDS <= 160 * 1024, "LDS budget");

// 16-byte chunk c (0..3) of the 64-byte row r sits in slot c ^ h((r >> 2) & 3), h = [0,2,3,1]
__device__ __forceinline__ int v6_off(int r, int c) { return piece_lds_off(r, c); }

#define V6_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

__global__ __launch_bounds__(512, 2) void bfp_gemm_v6(const GemmArgs a, const float* __restrict__ gx,
                                                      const float* __restrict__ gw, long long mpad, long long npad,
                                                      const int* __restrict__ xlist, const int* __restrict__ wlist,
                                                      int list_cap) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[V6_LDS];
    if (xlist && (xlist[0] > list_cap || wlist[0] > list_cap)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, lq = lane >> 4;

    const int tiles_m = (int)((a.M + V5_BM - 1) / V5_BM), tiles_n = (int)((a.N + V5_BN - 1) / V5_BN);
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int GM = 4, in_group = GM * tiles_n, group_id = pid / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (pid % in_group) % gsz, tn = (pid % in_group) / gsz;
    const long long m0 = (long long)tm * V5_BM, n0 = (long long)tn * V5_BN;
    const int nsteps = (int)(a.K >> 7), ngroups = nsteps >> 1;      // steps of 128, groups of 256

    f32x2 acc[4][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j][0] = f32x2{0.f, 0.f}; acc[i][j][1] = f32x2{0.f, 0.f}; }
    const i32x4 magicv = {V5_MAGIC_I, V5_MAGIC_I, V5_MAGIC_I, V5_MAGIC_I};

    const long long kp = a.K >> 6;
    const long long pa_max = ((a.M + 127) / 128) * 8 - 1, pb_max = ((a.N + 127) / 128) * 8 - 1;
    const int8_t* srcA0 = a.xm + min((m0 >> 4) + wave, pa_max) * kp * 1024 + lane * 16;
    const int8_t* srcA1 = a.xm + min((m0 >> 4) + wave + 8, pa_max) * kp * 1024 + lane * 16;
    const int8_t* srcB0 = a.wm + min((n0 >> 4) + wave, pb_max) * kp * 1024 + lane * 16;
    auto stage = [&](int step, int slot) {
#pragma unroll
        for (int u = 0; u < V6_SUB; ++u) {
            unsigned char* base = smem + slot * V6_STAGE + u * V5_STAGE;
            const long long ko = (long long)(step * V6_SUB + u) * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA0 + ko), (lptr_t)(base + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA1 + ko), (lptr_t)(base + (wave + 8) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(srcB0 + ko), (lptr_t)(base + V5_A_BYTES + wave * 1024), 16, 0, 0);
        }
    };
    const float* gxs = gx + m0 + lane * 4;
    const float* gws = gw + n0 + lane * 4;
    auto stage_scales = [&](int g) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gxs + (long long)g * mpad), (lptr_t)(smem + V6_GA + (g & 1) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gws + (long long)g * npad), (lptr_t)(smem + V6_GB + (g & 1) * 1024), 16, 0, 0);
    };
    int aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aoff[i] = v6_off(wm * 64 + i * 16 + l16, lq);
        boff[i] = V5_A_BYTES + v6_off(wn * 64 + i * 16 + l16, lq);
    }

    stage(0, 0);
    if (nsteps > 1) stage(1, 1);
    int islot = 2 % V6_S, rslot = 0;

    for (int g = 0; g < ngroups; ++g) {
        i32x4 ci[4][4];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int t = 2 * g + st;
            // stage t must have landed; only the next stage's 6 loads may stay in flight (the group's scale
            // loads are issued BEFORE them, so they have landed by the second step, ahead of the fold)
            if (t + 1 < nsteps) V6_WAIT(6); else V6_WAIT(0);
            __builtin_amdgcn_s_barrier();
            if (st == 0) stage_scales(g);
            if (t + 2 < nsteps) {
                stage(t + 2, islot);
                islot = islot + 1 == V6_S ? 0 : islot + 1;
            }
#pragma unroll
            for (int u = 0; u < V6_SUB; ++u) {
                const unsigned char* sbase = smem + rslot * V6_STAGE + u * V5_STAGE;
                i32x4 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = *reinterpret_cast<const i32x4*>(sbase + aoff[i]);
                    fb[i] = *reinterpret_cast<const i32x4*>(sbase + boff[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (st == 0 && u == 0) ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], magicv, 0, 0, 0);
                        else ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
                    }
            }
            rslot = rslot + 1 == V6_S ? 0 : rslot + 1;
        }
        const float* ga = reinterpret_cast<const float*>(smem + V6_GA) + (g & 1) * 256;
        const float* gb = reinterpret_cast<const float*>(smem + V6_GB) + (g & 1) * 256;
        f32x2 sw2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float sw = gb[wn * 64 + j * 16 + l16]; sw2[j] = f32x2{sw, sw}; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 sx = *reinterpret_cast<const f32x4*>(&ga[wm * 64 + i * 16 + lq * 4]);
            const f32x2 sx01 = {sx[0], sx[1]}, sx23 = {sx[2], sx[3]};
            const f32x2 nm01 = sx01 * (-V5_MAGIC_F), nm23 = sx23 * (-V5_MAGIC_F);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 f = __builtin_bit_cast(f32x4, ci[i][j]);
                const f32x2 u01 = __builtin_elementwise_fma(f32x2{f[0], f[1]}, sx01, nm01);
                const f32x2 u23 = __builtin_elementwise_fma(f32x2{f[2], f[3]}, sx23, nm23);
                acc[i][j][0] = __builtin_elementwise_fma(u01, sw2[j], acc[i][j][0]);
                acc[i][j][1] = __builtin_elementwise_fma(u23, sw2[j], acc[i][j][1]);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long col = n0 + wn * 64 + j * 16 + l16;
            if (col >= a.N) continue;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = m0 + wm * 64 + i * 16 + lq * 4 + r;
                if (row < a.M) a.y[row * a.ldy + col] = acc[i][j][r >> 1][r & 1] + bv;
            }
        }
}

int launch_bfp_gemm_v6(const GemmArgs& a, const float* gx, const float* gw, long long mpad, long long npad,
                       const int* xlist, const int* wlist, int list_cap, hipStream_t st) {
    const unsigned tiles = (unsigned)(((a.M + V5_BM - 1) / V5_BM) * ((a.N + V5_BN - 1) / V5_BN));
    hipLaunchKernelGGL(bfp_gemm_v6, tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// Tail launch of the default path.  Normal case: add the exception blocks of both operands back (mi355q_fix.h).
// If an exception list overflowed, the int32-chain kernel returned at once and this launch forms the whole
// product with the blockwise-exact body instead, correcting each tile right after its stores.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void bfp_gemm_tail(const GemmArgs a, const uint8_t* __restrict__ xf,
                                                        const uint8_t* __restrict__ wf, const int* __restrict__ xlist,
                                                        const int* __restrict__ wlist, int list_cap,
                                                        const float* __restrict__ xscale, const float* __restrict__ wscale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(V2Smem)];
    // row mode (behind the row-scale GEMM, which forms its own correction vectors): this launch only acts when an
    // exception bucket overflowed -- it then forms the whole product blockwise -- and leaves at once otherwise
    const bool overflow = a.row_mode ? (xlist[0] != 0 || wlist[0] != 0) : (xlist[0] > list_cap || wlist[0] > list_cap);
    if (overflow) {
        const int ntiles = (int)(((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN));
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            bfp_gemm_v2_body(a, xf, wf, *reinterpret_cast<V2Smem*>(smem), tile);
            long long m0, n0;
            v2_tile_origin(a, tile, m0, n0);
            __threadfence();
            __syncthreads();
            if (a.row_mode) tile_fix_body(a, row_bucket(xlist, m0, a.x_bcap), row_bucket(wlist, n0, a.w_bcap), a.x_bcap, a.w_bcap, m0, n0);
            else tile_fix_body(a, xlist, wlist, list_cap, list_cap, m0, n0);
            __syncthreads();
        }
        return;
    }
    if (!a.row_mode) block_fix_body(a, xlist, wlist, list_cap, blockIdx.x, gridDim.x);
}

int launch_bfp_gemm_tail(const GemmArgs& a, const uint8_t* xf, const uint8_t* wf, const int* xlist, const int* wlist,
                         int list_cap, hipStream_t st, const float* xscale, const float* wscale) {
    // two workgroups per CU: enough for the fallback GEMM (it walks the tiles) and cheap to dispatch when the
    // launch only has the sparse correction to do
    unsigned tiles = (unsigned)(((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN));
    if (tiles > 512) tiles = 512;
    hipLaunchKernelGGL(bfp_gemm_tail, tiles, 256, 0, st, a, xf, wf, xlist, wlist, list_cap, xscale, wscale);
    return (int)hipGetLastError();
}

}  // namespace mi355q
