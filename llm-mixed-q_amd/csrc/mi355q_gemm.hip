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
#include "mi355q_gemm_v2.h"
#include "mi355q_align.h"
#include "mi355q_align_row.h"
#include "mi355q_fix.h"

namespace mi355q {


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



// Row alignment of a packed operand (mi355q_align_row.h): one workgroup per row.  K % 64 == 0, K <= 1024 * MAXIT.
template <int MAXIT>
__global__ __launch_bounds__(256) void bfp_align_rows_kernel(const int8_t* __restrict__ mi, const uint8_t* __restrict__ ei,
                                                             int8_t* __restrict__ mt, uint8_t* __restrict__ eo,
                                                             uint8_t* __restrict__ flag, float* __restrict__ rscale,
                                                             int exp_offset, int* __restrict__ list, long long rows,
                                                             long long K, int bcap) {
    __shared__ RowAlignSmem rsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nkb = (int)(K >> 4), nit = (nkb + 63) >> 6;
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        unsigned pk[MAXIT];
        int amax[MAXIT], code[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int kb = it * 64 + wave * 16 + (lane >> 2);
            const bool valid = it < nit && kb < nkb;
            pk[it] = valid ? *reinterpret_cast<const unsigned*>(mi + row * K + (long long)kb * 16 + (lane & 3) * 4) : 0u;
            code[it] = valid ? (int)ei[row * nkb + kb] : 0;
            const unsigned v = pk[it];
            int am = max(max(abs((int)(int8_t)(v & 0xFF)), abs((int)(int8_t)((v >> 8) & 0xFF))),
                         max(abs((int)(int8_t)((v >> 16) & 0xFF)), abs((int)(int8_t)(v >> 24))));
            am = max(am, __shfl_xor(am, 1));
            am = max(am, __shfl_xor(am, 2));
            amax[it] = am;
        }
        int E = 0;
        const bool flagged = align_row<MAXIT>(pk, amax, code, nit, nkb, row, list, rsm, E, bcap);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int kb = it * 64 + wave * 16 + (lane >> 2);
            if (it < nit && kb < nkb) {
                *reinterpret_cast<unsigned*>(mt + tiled_offset(row, (long long)kb * 16 + (lane & 3) * 4, K)) = pk[it];
                if ((lane & 3) == 0) eo[row * nkb + kb] = (uint8_t)(flagged ? E : code[it]);
            }
        }
        if (tid == 0) {
            flag[row] = flagged ? 1 : 0;
            rscale[row] = flagged ? __builtin_ldexpf(1.0f, E - exp_offset) : 0.0f;
        }
        if (row + gridDim.x < rows) __syncthreads();      // (the next row reuses the decision words in LDS)
    }
}

int launch_bfp_align_rows(const int8_t* mi, const uint8_t* ei, int8_t* mt, uint8_t* eo, uint8_t* flag, float* rscale,
                          int exp_offset, int* list, long long rows, long long K, hipStream_t st, int bcap) {
    long long grid = rows;
    if (grid > 65536) grid = 65536;
    if (grid < 1) grid = 1;
    if (list) {
        const hipError_t e = hipMemsetAsync(list, 0, (size_t)row_list_words(rows, bcap) * 4, st);
        if (e != hipSuccess) return (int)e;
    }
    if (K <= 4096)
        hipLaunchKernelGGL((bfp_align_rows_kernel<4>), (unsigned)grid, 256, 0, st, mi, ei, mt, eo, flag, rscale, exp_offset, list, rows, K, bcap);
    else if (K <= 8192)
        hipLaunchKernelGGL((bfp_align_rows_kernel<8>), (unsigned)grid, 256, 0, st, mi, ei, mt, eo, flag, rscale, exp_offset, list, rows, K, bcap);
    else if (K <= 16384)
        hipLaunchKernelGGL((bfp_align_rows_kernel<16>), (unsigned)grid, 256, 0, st, mi, ei, mt, eo, flag, rscale, exp_offset, list, rows, K, bcap);
    else
        return MI355Q_E_UNSUPPORTED;
    return (int)hipGetLastError();
}

// Blockwise-exact GEMM over aligned operands.  guard != 0: act only as the fallback of the int32-chain kernel
// (when an exception list overflowed).  Exception blocks of either operand are added back per tile.
__global__ __launch_bounds__(256, 2) void bfp_gemm_v2(const GemmArgs a, const uint8_t* __restrict__ xf,
                                                      const uint8_t* __restrict__ wf, const int* __restrict__ xlist,
                                                      const int* __restrict__ wlist, int list_cap, int guard) {
    __shared__ V2Smem sm;
    if (guard && xlist[0] <= list_cap && wlist[0] <= list_cap) return;
    bfp_gemm_v2_body(a, xf, wf, sm, blockIdx.x);
    if (xlist || wlist) {
        long long m0, n0;
        v2_tile_origin(a, blockIdx.x, m0, n0);
        __threadfence();
        __syncthreads();
        if (a.row_mode) tile_fix_body(a, row_bucket(xlist, m0, a.x_bcap), row_bucket(wlist, n0, a.w_bcap), a.x_bcap, a.w_bcap, m0, n0);
        else tile_fix_body(a, xlist, wlist, list_cap, list_cap, m0, n0);
    }
}


int launch_bfp_gemm_aligned(const GemmArgs& a, const uint8_t* xf, const uint8_t* wf, const int* xlist,
                            const int* wlist, int list_cap, int guard, hipStream_t st) {
    const unsigned tiles = (unsigned)(((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN));
    hipLaunchKernelGGL(bfp_gemm_v2, tiles, 256, 0, st, a, xf, wf, xlist, wlist, list_cap, guard);
    return (int)hipGetLastError();
}

int launch_bfp_gemm(const GemmArgs& a, int variant, hipStream_t st) {
    (void)variant;
    dim3 grid((unsigned)((a.N + V1_BN - 1) / V1_BN), (unsigned)((a.M + V1_BM - 1) / V1_BM));
    hipLaunchKernelGGL(bfp_gemm_v1, grid, 256, 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace mi355q
