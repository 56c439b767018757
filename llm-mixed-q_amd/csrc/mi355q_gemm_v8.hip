// mi355q_gemm_v8.hip -- block-floating-point GEMM over ROW-aligned operands (gfx950).
//
// When the align step can put every block of a row onto ONE exponent (whole-K window, exceptions kept
// aside -- mi355q_align.h), the contraction is a plain int8 x int8 -> int32 GEMM with one scale per row of x
// and one per row of w:
//     y[m,n] = sx[m] * sw[n] * ( sum_k xm'[m,k] * wm'[n,k] )  (+ bias[n]),   K <= 131072 (int32 cannot overflow)
// No rescale in the K loop, so the int32 accumulators are the only live tile: they sit in AccVGPRs and the wave
// tile can be 128 x 128 (config <2>: 4 waves, one per SIMD) or 128 x 64 (config <4>: 8 waves, two per SIMD).
//
// Workgroup tile 256 x 256, K-step 64: one stage = A 16 KiB + B 16 KiB, 4 stages in LDS filled by
// global_load_lds (1-KiB pre-swizzled pieces, mi355q_gemm_v2.h) two steps ahead; counted s_waitcnt vmcnt and ONE
// s_barrier per step; v_mfma_i32_16x16x64_i8; fragments of step t+1 are read while the MFMAs of step t run.
// Roofline: int8 MFMA, 2*M*N*K ops; LDS traffic 64 KiB (config <2>) per 256x256x64 step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_gemm_v2.h"
#include "mi355q_fix.h"

namespace mi355q {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int V8_BM = 256, V8_BN = 256, V8_S = 4;
constexpr int V8_HALF = 256 * 64, V8_STAGE = 2 * V8_HALF, V8_LDS = V8_S * V8_STAGE;
static_assert(V8_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int v8_off(int r, int c) { return r * 64 + ((c ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3)) << 4); }

#define V8_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <int WAVES_N, int PREFETCH>
__global__ __launch_bounds__(WAVES_N * 128, 1) void bfp_gemm_v8(const GemmArgs a, const float* __restrict__ sx,
                                                                const float* __restrict__ sw,
                                                                const int* __restrict__ xlist,
                                                                const int* __restrict__ wlist, int list_cap) {
    constexpr int NW = 2 * WAVES_N, TI = 8, TJ = 16 / WAVES_N, LPW = 32 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[V8_LDS];
    if (xlist && (xlist[0] != 0 || wlist[0] != 0)) return;        // a bucket overflowed: the fallback launch runs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N, l16 = lane & 15, lq = lane >> 4;

    const int tiles_m = (int)((a.M + V8_BM - 1) / V8_BM), tiles_n = (int)((a.N + V8_BN - 1) / V8_BN);
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int GM = 4, in_group = GM * tiles_n, group_id = pid / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (pid % in_group) % gsz, tn = (pid % in_group) / gsz;
    const long long m0 = (long long)tm * V8_BM, n0 = (long long)tn * V8_BN;
    const int nsteps = (int)(a.K >> 6);

    const long long kp = a.K >> 6;
    const long long pa_max = ((a.M + 127) / 128) * 8 - 1, pb_max = ((a.N + 127) / 128) * 8 - 1;
    // piece p of a stage: p < 16 -> 16 rows of A, else 16 rows of B; this wave stages pieces wave + NW * q
    const int8_t* src[LPW];
    int dst[LPW];
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
        const int p = wave + NW * q;
        src[q] = p < 16 ? a.xm + min((m0 >> 4) + p, pa_max) * kp * 1024 + lane * 16
                        : a.wm + min((n0 >> 4) + (p - 16), pb_max) * kp * 1024 + lane * 16;
        dst[q] = p * 1024;
    }
    auto stage = [&](int step, int slot) {
#pragma unroll
        for (int q = 0; q < LPW; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[q] + (long long)step * 1024),
                                             (lptr_t)(smem + slot * V8_STAGE + dst[q]), 16, 0, 0);
    };
    int aoff[TI], boff[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) aoff[i] = v8_off(wm * 128 + i * 16 + l16, lq);
#pragma unroll
    for (int j = 0; j < TJ; ++j) boff[j] = V8_HALF + v8_off(wn * (TJ * 16) + j * 16 + l16, lq);

    i32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = i32x4{0, 0, 0, 0};

    auto read_frags = [&](int slot, i32x4 (&fa)[TI], i32x4 (&fb)[TJ]) {
        const unsigned char* sbase = smem + slot * V8_STAGE;
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[i] = *reinterpret_cast<const i32x4*>(sbase + aoff[i]);
#pragma unroll
        for (int j = 0; j < TJ; ++j) fb[j] = *reinterpret_cast<const i32x4*>(sbase + boff[j]);
    };
    auto mfmas = [&](const i32x4 (&fa)[TI], const i32x4 (&fb)[TJ]) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };

    stage(0, 0);
    if (nsteps > 1) stage(1, 1);
    if (nsteps > 2) stage(2, 2);

    if (PREFETCH == 2) {
        // as PREFETCH == 1, with the next step's fragment reads and the LDS-DMA loads issued BETWEEN the MFMAs of
        // the current step (one row of MFMA tiles, then two reads / one load), so the matrix pipe never waits
        // for their issue slots
        i32x4 fa0[TI], fb0[TJ], fa1[TI], fb1[TJ];
        if (nsteps > 2) V8_WAIT(2 * LPW); else V8_WAIT(LPW);
        __builtin_amdgcn_s_barrier();
        read_frags(0, fa0, fb0);
        auto body = [&](const i32x4 (&fa)[TI], const i32x4 (&fb)[TJ], i32x4 (&na)[TI], i32x4 (&nb)[TJ], int nslot,
                        bool do_stage, int sstep, int sslot) {
            const unsigned char* sbase = smem + nslot * V8_STAGE;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
                na[i] = *reinterpret_cast<const i32x4*>(sbase + aoff[i]);
                if (i < TJ) nb[i] = *reinterpret_cast<const i32x4*>(sbase + boff[i]);
                if (do_stage && i >= TI - LPW) {
                    const int q = i - (TI - LPW);
                    __builtin_amdgcn_global_load_lds((gptr_t)(src[q] + (long long)sstep * 1024),
                                                     (lptr_t)(smem + sslot * V8_STAGE + dst[q]), 16, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TJ, 0);     // TJ MFMA
                if (i < TJ) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS reads
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i >= TI - LPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM
            }
        };
        for (int t = 0; t < nsteps; t += 2) {
            if (t + 2 < nsteps) V8_WAIT(LPW); else V8_WAIT(0);
            __builtin_amdgcn_s_barrier();
            body(fa0, fb0, fa1, fb1, (t + 1) & 3, t + 3 < nsteps, t + 3, (t + 3) & 3);
            if (t + 3 < nsteps) V8_WAIT(LPW); else V8_WAIT(0);
            __builtin_amdgcn_s_barrier();
            body(fa1, fb1, fa0, fb0, (t + 2) & 3, t + 4 < nsteps, t + 4, t & 3);
        }
    } else if (PREFETCH) {
        // fragments of step t+1 are requested before the MFMAs of step t are issued; two register sets, the
        // loop is unrolled by two (nsteps is even: K % 128 == 0)
        i32x4 fa0[TI], fb0[TJ], fa1[TI], fb1[TJ];
        if (nsteps > 2) V8_WAIT(2 * LPW); else V8_WAIT(LPW);
        __builtin_amdgcn_s_barrier();
        read_frags(0, fa0, fb0);
        for (int t = 0; t < nsteps; t += 2) {
            // stage t+1 landed (for every wave, after the barrier); stage t+2 may be in flight
            if (t + 2 < nsteps) V8_WAIT(LPW); else V8_WAIT(0);
            __builtin_amdgcn_s_barrier();
            if (t + 3 < nsteps) stage(t + 3, (t + 3) & 3);
            read_frags((t + 1) & 3, fa1, fb1);
            mfmas(fa0, fb0);
            if (t + 3 < nsteps) V8_WAIT(LPW); else V8_WAIT(0);
            __builtin_amdgcn_s_barrier();
            if (t + 4 < nsteps) stage(t + 4, t & 3);
            read_frags((t + 2) & 3, fa0, fb0);          // past the end: stale data, never used
            mfmas(fa1, fb1);
        }
    } else {
        for (int t = 0; t < nsteps; ++t) {
            i32x4 fa[TI], fb[TJ];
            if (t + 2 < nsteps) V8_WAIT(2 * LPW); else if (t + 1 < nsteps) V8_WAIT(LPW); else V8_WAIT(0);
            __builtin_amdgcn_s_barrier();
            if (t + 3 < nsteps) stage(t + 3, (t + 3) & 3);
            read_frags(t & 3, fa, fb);
            mfmas(fa, fb);
        }
    }

    // ---- epilogue: y = float(acc) * sx[m] * sw[n] + bias[n]
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const long long col = n0 + wn * (TJ * 16) + j * 16 + l16;
        const bool cok = col < a.N;
        const float swv = cok ? sw[col] : 0.f;
        const float bv = (cok && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = m0 + wm * 128 + i * 16 + lq * 4 + r;
                if (cok && row < a.M) a.y[row * a.ldy + col] = (float)acc[i][j][r] * sx[row] * swv + bv;
            }
        }
    }
    // ---- exception blocks of this tile's rows / columns (usually a few dozen): added to the tile just stored
    if (xlist) {
        V8_WAIT(0);
        __syncthreads();
        tile_fix_pairs<V8_BM, V8_BN>(a, row_bucket(xlist, m0), row_bucket(wlist, n0), ROW_BCAP, m0, n0);
    }
}

int launch_bfp_gemm_v8(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist,
                       int list_cap, hipStream_t st) {
    const unsigned tiles = (unsigned)(((a.M + V8_BM - 1) / V8_BM) * ((a.N + V8_BN - 1) / V8_BN));
    const char* dbg = getenv("MI355Q_V8_CFG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 1) hipLaunchKernelGGL((bfp_gemm_v8<4, 0>), tiles, 512, 0, st, a, sx, sw, xlist, wlist, list_cap);
    else if (d == 2) hipLaunchKernelGGL((bfp_gemm_v8<2, 0>), tiles, 256, 0, st, a, sx, sw, xlist, wlist, list_cap);
    else if (d == 3) hipLaunchKernelGGL((bfp_gemm_v8<4, 1>), tiles, 512, 0, st, a, sx, sw, xlist, wlist, list_cap);
    else if (d == 4) hipLaunchKernelGGL((bfp_gemm_v8<2, 2>), tiles, 256, 0, st, a, sx, sw, xlist, wlist, list_cap);
    else if (d == 5) hipLaunchKernelGGL((bfp_gemm_v8<2, 1>), tiles, 256, 0, st, a, sx, sw, xlist, wlist, list_cap);
    else hipLaunchKernelGGL((bfp_gemm_v8<4, 2>), tiles, 512, 0, st, a, sx, sw, xlist, wlist, list_cap);
    return (int)hipGetLastError();
}

}  // namespace mi355q
