// mi355q_gemm_v3.hip -- the fast block-floating-point GEMM for gfx950: int32 MFMA chains over
// exponent-aligned K-groups of 256 values.
//
//   y[m,n] = sum_g  gx[g][m] * gw[g][n] * ( sum_{k in group g} xm'[m,k] * wm'[n,k] )   (+ bias[n])
//
// xm'/wm' are the exponent-aligned int8 mantissas written by the align kernel, gx/gw the per
// (group,row) scales 2^(effective exponent - bias) as fp32, group-major ([K/256][rows padded to 256]).
// A (row, group) the align step could not shift onto one exponent has scale 0 here: its exact
// contribution is added by the sparse correction kernel.  No data-dependent branch in this kernel.
//
// Structure: workgroup = 128 x 128 outputs, 4 waves as 2 x 2 (one per SIMD), wave tile 64 x 64 =
// 2 x 2 MFMA tiles of 32 x 32; TWO workgroups per CU, so that while one folds a group's int32 tile
// into fp32 (VALU) the other keeps the matrix pipe busy.
//   * K-step 64: A 128x64 B + B 128x64 B = 16 KiB per stage, 4 stages in LDS, filled by
//     global_load_lds (16 B / lane, 4 per wave per stage) three steps ahead; counted s_waitcnt vmcnt,
//     one raw s_barrier per step; 64-byte LDS rows XOR-swizzled through the per-lane SOURCE address so
//     the ds_read_b128 fragment reads are conflict-free;
//   * per step and wave: 2 K-slices of 32, each 4 fragment reads and 4 x v_mfma_i32_32x32x32_i8;
//   * once per group (4 steps): acc_f32 += float(acc_i32) * gx[m] * gw[n]; the group scales arrive by
//     global_load_lds too (2 per wave per group) -- no VGPR loads in the loop, so the compiler never
//     drains the pipeline.
// Roofline: int8 MFMA (2 * M * N * K ops); HBM traffic is the operands once per tile pass through L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_gemm_v2.h"
#include "mi355q_fix.h"

namespace mi355q {

constexpr int V3_BM = 128, V3_BN = 128, V3_BK = 64, V3_S = 4;
constexpr int V3_A_BYTES = V3_BM * V3_BK, V3_B_BYTES = V3_BN * V3_BK, V3_STAGE = V3_A_BYTES + V3_B_BYTES;
constexpr int V3_GA = V3_S * V3_STAGE;          // float ga[2][256] (first 128 of each used)
constexpr int V3_GB = V3_GA + 2 * 256 * 4;      // float gb[2][256]
constexpr int V3_LDS = V3_GB + 2 * 256 * 4;
static_assert(2 * V3_LDS <= 160 * 1024, "two workgroups per CU");

// 16-byte chunk c (0..3) of the 64-byte row r sits in slot c ^ h((r >> 2) & 3), h = [0,2,3,1]
__device__ __forceinline__ int v3_off(int r, int c) { return r * V3_BK + ((c ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3)) << 4); }

#define V3_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

template <int DBG>
__global__ __launch_bounds__(256, 2) void bfp_gemm_v3(const GemmArgs a, const float* __restrict__ gx,
                                                      const float* __restrict__ gw, long long mpad, long long npad,
                                                      const int* __restrict__ xlist, const int* __restrict__ wlist,
                                                      int list_cap, const uint8_t* __restrict__ xf,
                                                      const uint8_t* __restrict__ wf) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[V3_LDS];
    static_assert(sizeof(V2Smem) <= V3_LDS, "the fallback body reuses this kernel's LDS");
    // too many unaligned row-groups for the sparse correction (decided on the device, uniform over the
    // grid): this launch runs the blockwise-fallback body instead; same 128 x 128 tiling and grid
    if (xlist && (xlist[0] > list_cap || wlist[0] > list_cap)) {
        bfp_gemm_v2_body(a, xf, wf, *reinterpret_cast<V2Smem*>(smem), blockIdx.x);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;

    const int tiles_m = (int)((a.M + V3_BM - 1) / V3_BM), tiles_n = (int)((a.N + V3_BN - 1) / V3_BN);
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int GM = 8, in_group = GM * tiles_n, group_id = pid / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (pid % in_group) % gsz, tn = (pid % in_group) / gsz;
    const long long m0 = (long long)tm * V3_BM, n0 = (long long)tn * V3_BN;
    const int nsteps = (int)(a.K >> 6), ngroups = nsteps >> 2;

    float acc[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging: the operands are stored tiled (mi355q_gemm.hip: tiled_offset) in 1-KiB pieces of
    // 16 rows x 64 B that are already the swizzled LDS image: one global_load_lds copies one piece.
    // A and B have 8 pieces per stage each, 2 + 2 per wave.
    const int8_t* srcA[2];
    const int8_t* srcB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        srcA[t] = a.xm + ((m0 >> 4) + wave + 4 * t) * (a.K >> 6) * 1024 + lane * 16;
        srcB[t] = a.wm + ((n0 >> 4) + wave + 4 * t) * (a.K >> 6) * 1024 + lane * 16;
    }
    auto stage = [&](int step) {
        unsigned char* base = smem + (step & (V3_S - 1)) * V3_STAGE;
        const long long ko = (long long)step * 1024;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[t] + ko), (lptr_t)(base + (wave + 4 * t) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(srcB[t] + ko), (lptr_t)(base + V3_A_BYTES + (wave + 4 * t) * 1024),
                                             16, 0, 0);
        }
    };
    // group scales: one 1-KiB piece (256 floats, the tile's 128 + the next tile's) per operand and group
    const float* gxs = gx + m0 + lane * 4;
    const float* gws = gw + n0 + lane * 4;
    auto stage_scales = [&](int g) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gxs + (long long)g * mpad), (lptr_t)(smem + V3_GA + (g & 1) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gws + (long long)g * npad), (lptr_t)(smem + V3_GB + (g & 1) * 1024), 16, 0, 0);
    };

    const int arow = wm * 64 + lr, brow = wn * 64 + lr;
    const int sx_base = wm * 64 + 4 * lh, sw_base = wn * 64 + lr;

    stage(0);
    if (nsteps > 1) stage(1);
    if (nsteps > 2) stage(2);

    for (int g = 0; g < ngroups; ++g) {
        i32x16 ci[2][2];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int t = 4 * g + st;
            // stage t must have landed; the 8 (10 right after a scale issue) younger loads may stay in flight
            if (t + 2 < nsteps) {
                if (st == 1 || st == 2) V3_WAIT(10); else V3_WAIT(8);
            } else {
                V3_WAIT(0);
            }
            if (DBG != 4 && DBG != 6) __builtin_amdgcn_s_barrier();
            if (st == 0) stage_scales(g);
            if (t + 3 < nsteps && DBG != 1 && DBG != 6) stage(t + 3);
            const unsigned char* sa = smem + (t & (V3_S - 1)) * V3_STAGE;
            const unsigned char* sb = sa + V3_A_BYTES;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                i32x4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (DBG == 5 || DBG == 6) {
                        fa[i] = i32x4{lane, p, i, t};
                        fb[i] = i32x4{lane, i, p, t};
                        asm volatile("" : "+v"(fa[i]), "+v"(fb[i]));
                    } else {
                        fa[i] = *reinterpret_cast<const i32x4*>(sa + v3_off(arow + i * 32, 2 * p + lh));
                        fb[i] = *reinterpret_cast<const i32x4*>(sb + v3_off(brow + i * 32, 2 * p + lh));
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (DBG == 2) {
                            ci[i][j][0] = fa[i][0] + fb[j][1];
                        } else if (st == 0 && p == 0) {
                            const i32x16 z = {0};
                            ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], z, 0, 0, 0);
                        } else {
                            ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
                        }
                    }
            }
        }
        // ---- fold the group: scales of group g landed before the wait of its 4th step
        if ((DBG == 3 || DBG == 6) && g + 1 < ngroups) continue;
        const float* ga = reinterpret_cast<const float*>(smem + V3_GA) + (g & 1) * 256;
        const float* gb = reinterpret_cast<const float*>(smem + V3_GB) + (g & 1) * 256;
        float sw[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) sw[j] = gb[sw_base + j * 32];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&ga[sx_base + i * 32 + 8 * q]);
                const float sx[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] += (float)ci[i][j][4 * q + r] * sx[r] * sw[j];
            }
    }

    // ---- store: C/D layout col = lane & 31, row = 8*(reg>>2) + 4*(lane>>5) + (reg&3)
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
// Variant 4: 256 x 128 workgroup tile (CU ingest per MAC 25 % below the 128 x 128 tile: the L2 -> LDS
// path, ~30 B/clk/CU, is what bounds this kernel), 8 waves as 4 x 2 with 64 x 64 wave tiles, ONE
// workgroup per CU.  Waves 4-7 (the second wave of each SIMD) run two K-steps behind waves 0-3, so a
// wave-set folds its int32 group tile into fp32 (VALU) while the other set keeps the matrix pipe busy.
// Six 24-KiB stages: a stage is written 3 intervals ahead, read by waves 0-3 in interval t and by
// waves 4-7 in interval t+2.
// =======================================================================================
constexpr int V4_BM = 256, V4_BN = 128, V4_S = 6, V4_LAG = 2;
constexpr int V4_A_BYTES = V4_BM * 64, V4_B_BYTES = V4_BN * 64, V4_STAGE = V4_A_BYTES + V4_B_BYTES;
constexpr int V4_GA = V4_S * V4_STAGE;          // float ga[2][256]
constexpr int V4_GB = V4_GA + 2 * 256 * 4;      // float gb[2][256] (first 128 used)
constexpr int V4_LDS = V4_GB + 2 * 256 * 4;
static_assert(V4_LDS <= 160 * 1024, "LDS budget");

template <bool FIRST>
__device__ __forceinline__ void v4_cluster(i32x16 (&ci)[2][2], const unsigned char* sa, const unsigned char* sb,
                                           int arow, int brow, int lh) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        i32x4 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i] = *reinterpret_cast<const i32x4*>(sa + v3_off(arow + i * 32, 2 * p + lh));
            fb[i] = *reinterpret_cast<const i32x4*>(sb + v3_off(brow + i * 32, 2 * p + lh));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (FIRST && p == 0) {
                    const i32x16 z = {0};
                    ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], z, 0, 0, 0);
                } else {
                    ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
                }
            }
    }
}

__global__ __launch_bounds__(512, 2) void bfp_gemm_v4(const GemmArgs a, const float* __restrict__ gx,
                                                      const float* __restrict__ gw, long long mpad, long long npad) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[V4_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wave & 3;                       // position inside the wave-set
    const int lag = (wave >> 2) * V4_LAG;          // waves 4..7 run two steps behind
    const int wm = (wave >> 2) * 2 + (ws >> 1), wn = ws & 1;    // set 0: rows 0-127, set 1: rows 128-255
    const int lr = lane & 31, lh = lane >> 5;

    const int tiles_m = (int)((a.M + V4_BM - 1) / V4_BM), tiles_n = (int)((a.N + V4_BN - 1) / V4_BN);
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int GM = 4, in_group = GM * tiles_n, group_id = pid / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (pid % in_group) % gsz, tn = (pid % in_group) / gsz;
    const long long m0 = (long long)tm * V4_BM, n0 = (long long)tn * V4_BN;
    const int nsteps = (int)(a.K >> 6), ngroups = nsteps >> 2;

    float acc[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    i32x16 ci[2][2];

    // staging: 16 A pieces + 8 B pieces of 1 KiB per stage; wave w copies A pieces {w, w+8} and B piece {w}
    const int8_t* srcA0 = a.xm + ((m0 >> 4) + wave) * (a.K >> 6) * 1024 + lane * 16;
    const int8_t* srcA1 = a.xm + ((m0 >> 4) + wave + 8) * (a.K >> 6) * 1024 + lane * 16;
    const int8_t* srcB0 = a.wm + ((n0 >> 4) + wave) * (a.K >> 6) * 1024 + lane * 16;
    auto stage = [&](int step, int buf) {
        unsigned char* base = smem + buf * V4_STAGE;
        const long long ko = (long long)step * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA0 + ko), (lptr_t)(base + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA1 + ko), (lptr_t)(base + (wave + 8) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(srcB0 + ko), (lptr_t)(base + V4_A_BYTES + wave * 1024), 16, 0, 0);
    };
    const float* gxs = gx + m0 + lane * 4;
    const float* gws = gw + n0 + lane * 4;
    auto stage_scales = [&](int g) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gxs + (long long)g * mpad), (lptr_t)(smem + V4_GA + (g & 1) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gws + (long long)g * npad), (lptr_t)(smem + V4_GB + (g & 1) * 1024), 16, 0, 0);
    };

    const int arow = wm * 64 + lr, brow = wn * 64 + lr;
    const int sx_base = wm * 64 + 4 * lh, sw_base = wn * 64 + lr;

    stage(0, 0);
    if (nsteps > 1) stage(1, 1);
    if (nsteps > 2) stage(2, 2);
    int ibuf = 3;                    // ring slot of the stage issued next
    int rbuf = (V4_S - lag) % V4_S;  // ring slot this wave reads in the current interval (slot of step tau - lag)

    const int nint = nsteps + V4_LAG;
    for (int tau = 0; tau < nint; ++tau) {
        // stage tau must have landed (own loads); younger loads may stay in flight
        if (tau + 2 < nsteps) {
            const int ph = tau & 3;
            if (ph == 1 || ph == 2) V3_WAIT(8); else V3_WAIT(6);
        } else {
            V3_WAIT(0);
        }
        __builtin_amdgcn_s_barrier();
        if ((tau & 3) == 0 && (tau >> 2) < ngroups) stage_scales(tau >> 2);
        if (tau + 3 < nsteps) {
            stage(tau + 3, ibuf);
            ibuf = ibuf + 1 == V4_S ? 0 : ibuf + 1;
        }
        const int t = tau - lag;
        if (t >= 0 && t < nsteps) {
            const unsigned char* sa = smem + rbuf * V4_STAGE;
            const unsigned char* sb = sa + V4_A_BYTES;
            const int st = t & 3;
            if (st == 0) v4_cluster<true>(ci, sa, sb, arow, brow, lh);
            else v4_cluster<false>(ci, sa, sb, arow, brow, lh);
            if (st == 3) {
                const int g = t >> 2;
                const float* ga = reinterpret_cast<const float*>(smem + V4_GA) + (g & 1) * 256;
                const float* gb = reinterpret_cast<const float*>(smem + V4_GB) + (g & 1) * 256;
                float sw[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) sw[j] = gb[sw_base + j * 32];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(&ga[sx_base + i * 32 + 8 * q]);
                        const float sx[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] += (float)ci[i][j][4 * q + r] * sx[r] * sw[j];
                    }
            }
        }
        rbuf = rbuf + 1 == V4_S ? 0 : rbuf + 1;
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

int launch_bfp_gemm_v4(const GemmArgs& a, const float* gx, const float* gw, long long mpad, long long npad,
                       hipStream_t st) {
    const unsigned tiles = (unsigned)(((a.M + V4_BM - 1) / V4_BM) * ((a.N + V4_BN - 1) / V4_BN));
    hipLaunchKernelGGL(bfp_gemm_v4, tiles, 512, 0, st, a, gx, gw, mpad, npad);
    return (int)hipGetLastError();
}


// =======================================================================================
// Variant 5: 256 x 128 workgroup tile, 8 waves as 4 x 2 (two per SIMD), wave tile 64 x 64 as 4 x 4 tiles of
// v_mfma_i32_16x16x64_i8 (one instruction = one K-step of 64 for one 16x16 tile; the four lane
// quarters hold the four 16-blocks), ONE workgroup per CU, four 24-KiB LDS stages.
// The fold of a group's int32 tile into fp32 costs 2 packed VALU per 2 outputs instead of 6 scalar:
// the chain starts from C = 0x4B400000 (the bit pattern of 1.5 * 2^23), so the accumulator read back as
// fp32 is exactly 12582912 + D (|D| < 2^22), and
//     u   = pk_fma(f, gx, -12582912 * gx)      ( = D * gx, exact: gx is a power of two )
//     acc = pk_fma(u, gw, acc)
// =======================================================================================
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int V5_BM = 256, V5_BN = 128, V5_S = 6, V5_P = V5_S - 1;   // stages, prefetch distance
constexpr int V5_A_BYTES = V5_BM * 64, V5_B_BYTES = V5_BN * 64, V5_STAGE = V5_A_BYTES + V5_B_BYTES;
constexpr int V5_GA = V5_S * V5_STAGE;          // float ga[2][256]
constexpr int V5_GB = V5_GA + 2 * 256 * 4;      // float gb[2][256] (first 128 used)
constexpr int V5_LDS = V5_GB + 2 * 256 * 4;
constexpr int V5_MAGIC_I = 0x4B400000;
constexpr float V5_MAGIC_F = 12582912.0f;
static_assert(V5_LDS <= 160 * 1024, "LDS budget");

template <int DBG, int LAG>
__global__ __launch_bounds__(512, 2) void bfp_gemm_v5(const GemmArgs a, const float* __restrict__ gx,
                                                      const float* __restrict__ gw, long long mpad, long long npad,
                                                      const int* __restrict__ xlist, const int* __restrict__ wlist,
                                                      int list_cap) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[V5_LDS];
    if (xlist && (xlist[0] > list_cap || wlist[0] > list_cap)) return;   // the fallback kernel takes this call
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, lq = lane >> 4;
    constexpr int P = V5_S - 1 - LAG;              // prefetch distance
    const int lag = (wave >> 2) * LAG;             // waves 4..7 run LAG steps behind waves 0..3

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
    const int nsteps = (int)(a.K >> 6), ngroups = nsteps >> 2;

    f32x2 acc[4][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j][0] = f32x2{0.f, 0.f}; acc[i][j][1] = f32x2{0.f, 0.f}; }
    const i32x4 magicv = {V5_MAGIC_I, V5_MAGIC_I, V5_MAGIC_I, V5_MAGIC_I};

    // staging: 16 A pieces + 8 B pieces of 1 KiB per stage (tiled operands); wave w copies A {w, w+8}, B {w}.
    // Rows past M / N: the tiled operand is allocated in whole 128-row tiles; an M tile of 256 may reach one
    // 128-row tile further, so those piece indices are clamped to the last allocated piece row.
    const long long kp = a.K >> 6;
    const long long pa_max = ((a.M + 127) / 128) * 8 - 1, pb_max = ((a.N + 127) / 128) * 8 - 1;
    const int8_t* srcA0 = a.xm + min((m0 >> 4) + wave, pa_max) * kp * 1024 + lane * 16;
    const int8_t* srcA1 = a.xm + min((m0 >> 4) + wave + 8, pa_max) * kp * 1024 + lane * 16;
    const int8_t* srcB0 = a.wm + min((n0 >> 4) + wave, pb_max) * kp * 1024 + lane * 16;
    auto stage = [&](int step, int slot) {
        unsigned char* base = smem + slot * V5_STAGE;
        const long long ko = (long long)step * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA0 + ko), (lptr_t)(base + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA1 + ko), (lptr_t)(base + (wave + 8) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(srcB0 + ko), (lptr_t)(base + V5_A_BYTES + wave * 1024), 16, 0, 0);
    };
    const float* gxs = gx + m0 + lane * 4;
    const float* gws = gw + n0 + lane * 4;
    auto stage_scales = [&](int g) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gxs + (long long)g * mpad), (lptr_t)(smem + V5_GA + (g & 1) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gws + (long long)g * npad), (lptr_t)(smem + V5_GB + (g & 1) * 1024), 16, 0, 0);
    };

    // fragment addresses inside a stage: row r of the tile sits in piece r / 16 at (r & 15) * 64, chunk lq swizzled
    int aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aoff[i] = v3_off(wm * 64 + i * 16 + l16, lq);
        boff[i] = V5_A_BYTES + v3_off(wn * 64 + i * 16 + l16, lq);
    }

#pragma unroll
    for (int p = 0; p < P; ++p)
        if (p < nsteps) stage(p, p);
    int islot = P % V5_S;                   // ring slot of the stage issued next
    int rslot = (V5_S - lag) % V5_S;        // ring slot this wave reads in the current interval

    // interval tau: every wave waits for / issues the loads of the shared ring; waves 0..3 compute step tau,
    // waves 4..7 step tau - LAG.  Intervals are walked in groups of four so that the step phase is static.
    i32x4 ci[4][4];
    const int nint = nsteps + LAG;
    for (int tg = 0; tg * 4 < nint; ++tg) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int tau = 4 * tg + ph;
            if (tau < nint) {
                // stage tau must have landed: the P-1 younger stages (3 loads each) and at most one pair of scale
                // loads may stay in flight; before a fold (own phase 3) the group's scales must have landed too
                if (tau + P - 1 < nsteps) {
                    if (P == 5) { if (ph == 3) V3_WAIT(9); else V3_WAIT(14); }
                    else { if (ph == 1 || ph == 2) V3_WAIT(8); else V3_WAIT(6); }
                } else {
                    V3_WAIT(0);
                }
                if (DBG != 4) __builtin_amdgcn_s_barrier();
                if (ph == 0 && tg < ngroups) stage_scales(tg);
                if (tau + P < nsteps && DBG != 1) {
                    stage(tau + P, islot);
                    islot = islot + 1 == V5_S ? 0 : islot + 1;
                }
                const int t = tau - lag;                 // this wave's step; its phase is (ph - LAG) & 3
                if (t >= 0 && t < nsteps) {
                    const unsigned char* sbase = smem + rslot * V5_STAGE;
                    i32x4 fa[4], fb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        fa[i] = *reinterpret_cast<const i32x4*>(sbase + aoff[i]);
                        fb[i] = *reinterpret_cast<const i32x4*>(sbase + boff[i]);
                    }
                    const bool first = (t & 3) == 0;
                    if (first) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], magicv, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (DBG == 2) ci[i][j][0] = fa[i][0] + fb[j][1];
                                else ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
                            }
                    }
                    if ((t & 3) == 3 && DBG != 3) {
                        // ---- fold the group (its scales landed before this interval's wait)
                        const int g = t >> 2;
                        const float* ga = reinterpret_cast<const float*>(smem + V5_GA) + (g & 1) * 256;
                        const float* gb = reinterpret_cast<const float*>(smem + V5_GB) + (g & 1) * 256;
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
                }
                rslot = rslot + 1 == V5_S ? 0 : rslot + 1;
            }
        }
    }

    // ---- store: 16x16 C/D layout col = lane & 15, row = 4 * (lane >> 4) + reg
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
                if (row < a.M && (DBG != 5 || acc[i][j][r >> 1][r & 1] == 1.2345e-30f)) a.y[row * a.ldy + col] = acc[i][j][r >> 1][r & 1] + bv;
            }
        }
}

int launch_bfp_gemm_v5(const GemmArgs& a, const float* gx, const float* gw, long long mpad, long long npad,
                       const int* xlist, const int* wlist, int list_cap, hipStream_t st) {
    const unsigned tiles = (unsigned)(((a.M + V5_BM - 1) / V5_BM) * ((a.N + V5_BN - 1) / V5_BN));
    const char* dbg = getenv("MI355Q_V3_DBG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 7) hipLaunchKernelGGL((bfp_gemm_v5<0, 2>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else if (d == 1) hipLaunchKernelGGL((bfp_gemm_v5<1, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else if (d == 2) hipLaunchKernelGGL((bfp_gemm_v5<2, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else if (d == 3) hipLaunchKernelGGL((bfp_gemm_v5<3, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else if (d == 4) hipLaunchKernelGGL((bfp_gemm_v5<4, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else if (d == 5) hipLaunchKernelGGL((bfp_gemm_v5<5, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else hipLaunchKernelGGL((bfp_gemm_v5<0, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    return (int)hipGetLastError();
}


// =======================================================================================
// Variant 6: variant 5 with a K-step of 128 per barrier (two 64-wide sub-steps per stage, three 48-KiB
// stages, prefetch distance 2): half as many barriers / waits per MFMA.
// =======================================================================================
constexpr int V6_S = 3, V6_SUB = 2;
constexpr int V6_STAGE = V6_SUB * V5_STAGE;          // [sub][A 16 KiB | B 8 KiB]
constexpr int V6_GA = V6_S * V6_STAGE, V6_GB = V6_GA + 2 * 256 * 4, V6_LDS = V6_GB + 2 * 256 * 4;
static_assert(V6_LDS <= 160 * 1024, "LDS budget");

template <int GMV, int PRIO>
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
    const int GM = GMV, in_group = GM * tiles_n, group_id = pid / in_group, first_m = group_id * GM;
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
        aoff[i] = v3_off(wm * 64 + i * 16 + l16, lq);
        boff[i] = V5_A_BYTES + v3_off(wn * 64 + i * 16 + l16, lq);
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
            if (t + 1 < nsteps) V3_WAIT(6); else V3_WAIT(0);
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
                if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (st == 0 && u == 0) ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], magicv, 0, 0, 0);
                        else ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
                    }
                if (PRIO) __builtin_amdgcn_s_setprio(0);
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
    const char* dbg = getenv("MI355Q_V6_CFG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 1) hipLaunchKernelGGL((bfp_gemm_v6<4, 1>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else if (d == 2) hipLaunchKernelGGL((bfp_gemm_v6<2, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else if (d == 3) hipLaunchKernelGGL((bfp_gemm_v6<8, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else if (d == 4) hipLaunchKernelGGL((bfp_gemm_v6<16, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    else hipLaunchKernelGGL((bfp_gemm_v6<4, 0>), tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    return (int)hipGetLastError();
}


// =======================================================================================
// Variant 7: variant 5's tile (256 x 128, 8 waves x 64 x 64, v_mfma_i32_16x16x64_i8, six 24-KiB stages) on
// a PING-PONG schedule.  Waves 0-3 (set A) and 4-7 (set B) are the two waves of each SIMD.  Every K-step
// of 64 has two phases separated by workgroup barriers:
//     phase 1:  A issues its 16 MFMAs of step t          |  B reads its fragments of step t from LDS,
//                                                         |    issues its LDS-DMA loads, folds a finished group
//     phase 2:  A reads fragments of step t+1, issues     |  B issues its 16 MFMAs of step t
//               its LDS-DMA loads, folds a finished group |
// so the matrix pipe always has exactly one wave feeding it while the other wave does everything else.
// =======================================================================================
template <bool FIRST>
__device__ __forceinline__ void v7_mfma(i32x4 (&ci)[4][4], const i32x4 (&fa)[4], const i32x4 (&fb)[4], const i32x4& magicv) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (FIRST) ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], magicv, 0, 0, 0);
            else ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
        }
}

__device__ __forceinline__ void v7_fold(f32x2 (&acc)[4][4][2], const i32x4 (&ci)[4][4], const float* ga, const float* gb,
                                        int wm, int wn, int l16, int lq) {
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

__global__ __launch_bounds__(512, 2) void bfp_gemm_v7(const GemmArgs a, const float* __restrict__ gx,
                                                      const float* __restrict__ gw, long long mpad, long long npad,
                                                      const int* __restrict__ xlist, const int* __restrict__ wlist,
                                                      int list_cap) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[V5_LDS];
    if (xlist && (xlist[0] > list_cap || wlist[0] > list_cap)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wave >> 2;                                  // 0 = A, 1 = B
    const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, lq = lane >> 4;
    constexpr int P = V5_S - 1;                                 // prefetch distance (5)

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
    const int nsteps = (int)(a.K >> 6), ngroups = nsteps >> 2;

    f32x2 acc[4][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j][0] = f32x2{0.f, 0.f}; acc[i][j][1] = f32x2{0.f, 0.f}; }
    const i32x4 magicv = {V5_MAGIC_I, V5_MAGIC_I, V5_MAGIC_I, V5_MAGIC_I};
    i32x4 ci[4][4];
    i32x4 fa[4], fb[4];

    const long long kp = a.K >> 6;
    const long long pa_max = ((a.M + 127) / 128) * 8 - 1, pb_max = ((a.N + 127) / 128) * 8 - 1;
    const int8_t* srcA0 = a.xm + min((m0 >> 4) + wave, pa_max) * kp * 1024 + lane * 16;
    const int8_t* srcA1 = a.xm + min((m0 >> 4) + wave + 8, pa_max) * kp * 1024 + lane * 16;
    const int8_t* srcB0 = a.wm + min((n0 >> 4) + wave, pb_max) * kp * 1024 + lane * 16;
    auto stage = [&](int step, int slot) {
        unsigned char* base = smem + slot * V5_STAGE;
        const long long ko = (long long)step * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA0 + ko), (lptr_t)(base + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA1 + ko), (lptr_t)(base + (wave + 8) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(srcB0 + ko), (lptr_t)(base + V5_A_BYTES + wave * 1024), 16, 0, 0);
    };
    const float* gxs = gx + m0 + lane * 4;
    const float* gws = gw + n0 + lane * 4;
    auto stage_scales = [&](int g) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gxs + (long long)g * mpad), (lptr_t)(smem + V5_GA + (g & 1) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gws + (long long)g * npad), (lptr_t)(smem + V5_GB + (g & 1) * 1024), 16, 0, 0);
    };
    int aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aoff[i] = v3_off(wm * 64 + i * 16 + l16, lq);
        boff[i] = V5_A_BYTES + v3_off(wn * 64 + i * 16 + l16, lq);
    }
    auto read_frags = [&](int slot) {
        const unsigned char* sbase = smem + slot * V5_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = *reinterpret_cast<const i32x4*>(sbase + aoff[i]);
            fb[i] = *reinterpret_cast<const i32x4*>(sbase + boff[i]);
        }
    };
    auto scales_of = [&](int g, const float*& ga, const float*& gb) {
        ga = reinterpret_cast<const float*>(smem + V5_GA) + (g & 1) * 256;
        gb = reinterpret_cast<const float*>(smem + V5_GB) + (g & 1) * 256;
    };

    // ---- prologue: every wave issues stages 0..P-1 and reads its fragments of step 0
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (p < nsteps) stage(p, p);
    int islot = P % V5_S;          // ring slot the next issued stage goes to
    int rslot = 1;                 // ring slot of the step whose fragments this wave reads next
    if (nsteps > P) { V3_WAIT(12); } else { V3_WAIT(0); }        // stage 0 landed (4 younger stages may fly)
    __builtin_amdgcn_s_barrier();                                  // P0
    read_frags(0);

    // Global barrier sequence G0, G1, ...: set A computes step t between G(2t) and G(2t+1) and does its
    // LDS / load / fold work between G(2t+1) and G(2t+2); set B runs the same body one barrier later.  Stage
    // t+1 is first read (by A) after G(2t+1), so every wave retires its own loads of that stage before
    // arriving there: A right after its MFMAs of step t, B at the end of its work phase of step t-1.
    // Per-wave issue order inside an iteration: [scales of the group (first step)] [stage t+P].
    if (set == 0) {
        for (int tg = 0; tg < ngroups; ++tg) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int t = 4 * tg + st;
                __builtin_amdgcn_s_barrier();                           // G(2t)
                if (st == 0) v7_mfma<true>(ci, fa, fb, magicv); else v7_mfma<false>(ci, fa, fb, magicv);
                // retire stage t+1 (younger: the stages issued in iterations t-3..t-1 and their scale loads;
                // before a fold, st == 3, also the group's scale loads)
                if (t + P <= nsteps) { if (st == 0 || st == 3) V3_WAIT(9); else V3_WAIT(11); } else V3_WAIT(0);
                __builtin_amdgcn_s_barrier();                           // G(2t+1)
                if (st == 3) {
                    const float *ga, *gb;
                    scales_of(tg, ga, gb);
                    v7_fold(acc, ci, ga, gb, wm, wn, l16, lq);
                }
                if (st == 0) stage_scales(tg);
                if (t + P < nsteps) stage(t + P, islot);
                islot = islot + 1 == V5_S ? 0 : islot + 1;
                if (t + 1 < nsteps) read_frags(rslot);
                rslot = rslot + 1 == V5_S ? 0 : rslot + 1;
            }
        }
        __builtin_amdgcn_s_barrier();                                   // G(2 nsteps): B's last barrier
    } else {
        // B's first wait: stage 1 before G1
        if (nsteps > P) { V3_WAIT(9); } else { V3_WAIT(0); }
        __builtin_amdgcn_s_barrier();                                   // G0
        for (int tg = 0; tg < ngroups; ++tg) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int t = 4 * tg + st;
                __builtin_amdgcn_s_barrier();                           // G(2t+1)
                if (st == 0) v7_mfma<true>(ci, fa, fb, magicv); else v7_mfma<false>(ci, fa, fb, magicv);
                __builtin_amdgcn_s_barrier();                           // G(2t+2)
                if (st == 3) {
                    const float *ga, *gb;
                    scales_of(tg, ga, gb);
                    v7_fold(acc, ci, ga, gb, wm, wn, l16, lq);
                }
                if (st == 0) stage_scales(tg);
                if (t + P < nsteps) stage(t + P, islot);
                islot = islot + 1 == V5_S ? 0 : islot + 1;
                if (t + 1 < nsteps) read_frags(rslot);
                rslot = rslot + 1 == V5_S ? 0 : rslot + 1;
                // retire stage t+2 before G(2t+3) (younger: iterations t-2..t; before the fold of iteration
                // 4g+3 the scales issued in iteration 4g must be in too: stricter count at st == 2)
                if (t + 1 + P <= nsteps) { if (st == 2 || st == 3) V3_WAIT(9); else V3_WAIT(11); } else V3_WAIT(0);
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

int launch_bfp_gemm_v7(const GemmArgs& a, const float* gx, const float* gw, long long mpad, long long npad,
                       const int* xlist, const int* wlist, int list_cap, hipStream_t st) {
    const unsigned tiles = (unsigned)(((a.M + V5_BM - 1) / V5_BM) * ((a.N + V5_BN - 1) / V5_BN));
    hipLaunchKernelGGL(bfp_gemm_v7, tiles, 512, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Tail launch of the default path.  Normal case: add the exception blocks of both operands back (mi355q_fix.h).
// If an exception list overflowed, the int32-chain kernel returned at once and this launch forms the whole
// product with the blockwise-exact body instead, correcting each tile right after its stores.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void bfp_gemm_tail(const GemmArgs a, const uint8_t* __restrict__ xf,
                                                        const uint8_t* __restrict__ wf, const int* __restrict__ xlist,
                                                        const int* __restrict__ wlist, int list_cap) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(V2Smem)];
    const bool overflow = a.row_mode ? (xlist[0] != 0 || wlist[0] != 0) : (xlist[0] > list_cap || wlist[0] > list_cap);
    if (overflow) {
        const int ntiles = (int)(((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN));
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            bfp_gemm_v2_body(a, xf, wf, *reinterpret_cast<V2Smem*>(smem), tile);
            long long m0, n0;
            v2_tile_origin(a, tile, m0, n0);
            __threadfence();
            __syncthreads();
            if (a.row_mode) tile_fix_body(a, row_bucket(xlist, m0), row_bucket(wlist, n0), ROW_BCAP, m0, n0);
            else tile_fix_body(a, xlist, wlist, list_cap, m0, n0);
            __syncthreads();
        }
        return;
    }
    if (!a.row_mode) block_fix_body(a, xlist, wlist, list_cap, blockIdx.x, gridDim.x);   // (row mode: done by the GEMM)
}

int launch_bfp_gemm_tail(const GemmArgs& a, const uint8_t* xf, const uint8_t* wf, const int* xlist, const int* wlist,
                         int list_cap, hipStream_t st) {
    // two workgroups per CU: enough for the fallback GEMM (it walks the tiles) and cheap to dispatch when the
    // launch only has the sparse correction to do
    unsigned tiles = (unsigned)(((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN));
    if (tiles > 512) tiles = 512;
    if (a.row_mode && tiles > 64) tiles = 64;        // only ever the fallback: keep the (usually empty) launch small
    hipLaunchKernelGGL(bfp_gemm_tail, tiles, 256, 0, st, a, xf, wf, xlist, wlist, list_cap);
    return (int)hipGetLastError();
}

int launch_bfp_gemm_v3(const GemmArgs& a, const float* gx, const float* gw, long long mpad, long long npad,
                       const int* xlist, const int* wlist, int list_cap, const uint8_t* xf, const uint8_t* wf,
                       hipStream_t st) {
    const unsigned tiles = (unsigned)(((a.M + V3_BM - 1) / V3_BM) * ((a.N + V3_BN - 1) / V3_BN));
    const char* dbg = getenv("MI355Q_V3_DBG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 1) hipLaunchKernelGGL(bfp_gemm_v3<1>, tiles, 256, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap, xf, wf);
    else if (d == 2) hipLaunchKernelGGL(bfp_gemm_v3<2>, tiles, 256, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap, xf, wf);
    else if (d == 3) hipLaunchKernelGGL(bfp_gemm_v3<3>, tiles, 256, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap, xf, wf);
    else if (d == 4) hipLaunchKernelGGL(bfp_gemm_v3<4>, tiles, 256, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap, xf, wf);
    else if (d == 5) hipLaunchKernelGGL(bfp_gemm_v3<5>, tiles, 256, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap, xf, wf);
    else if (d == 6) hipLaunchKernelGGL(bfp_gemm_v3<6>, tiles, 256, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap, xf, wf);
    else hipLaunchKernelGGL(bfp_gemm_v3<0>, tiles, 256, 0, st, a, gx, gw, mpad, npad, xlist, wlist, list_cap, xf, wf);
    return (int)hipGetLastError();
}

}  // namespace mi355q
