// mi355q_gemm_tile.h -- what the tile-GEMM translation units share (mi355q_gemm_v8.hip: 128-row tiles and the
// staggered schedules; mi355q_gemm_v9.hip: the 256 x 256 tile of the benchmark path): the LDS constants of the v8
// kernel, the exception-entry accessors, the atomics last resort, the in-launch blockwise fallback and the split-K
// workspace.
#ifndef MI355Q_GEMM_TILE_H
#define MI355Q_GEMM_TILE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_gemm_v2.h"
#include "mi355q_fix.h"

namespace mi355q {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int V8_BM = 256, V8_BN = 256, V8_S = 3, V8_NW = 8, V8_NT = V8_NW * 64;
constexpr int V8_HALF = 256 * 64, V8_STAGE = 2 * V8_HALF;
constexpr int V8_BUCKET = 4096;                     // LDS copy of one exception bucket (ROW_BUCKET_WORDS * 4 <= 4096)
constexpr int V8_XB = V8_S * V8_STAGE, V8_WB = V8_XB + V8_BUCKET, V8_MAP = V8_WB + V8_BUCKET;
constexpr int V8_SXT = V8_MAP + 2 * 256 * 4, V8_SWT = V8_SXT + 1024, V8_BIAS = V8_SWT + 1024;
constexpr int V8_FLAGS = V8_BIAS + 1024;             // a few flag words
constexpr int V8_OVF = V8_FLAGS + 256;                // LDS copies of the two lists' header words (word 0 = overflow)
constexpr int V8_CORR = V8_OVF + 512;
constexpr int V8_LDS = 159 * 1024;            // (the blockwise fallback body keeps a few words of its own)
constexpr int V8_FAST_MAX = (V8_LDS - V8_CORR) / 1024;      // entries (x + w) whose vectors fit beside the stages
constexpr int V8_SLOW_MAX = V8_S * V8_STAGE / 1024;         // ... that fit the stage area after the K loop
static_assert(ROW_BUCKET_WORDS * 4 <= V8_BUCKET, "bucket copy");
static_assert(V8_FAST_MAX >= 40, "spare LDS for correction vectors");

__device__ __forceinline__ int v8_off(int r, int c) { return piece_lds_off(r, c); }

#define V8_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// entry i of the combined list (x entries first, then w entries) in the LDS bucket copies
__device__ __forceinline__ int* v8_entry(int* xb, int* wb, int cx, int i) {
    return (i < cx ? xb + EXC_ENTRY * i : wb + EXC_ENTRY * (i - cx)) + EXC_HEADER;
}

// Last resort (more entries than LDS holds): add the exception products to the tile after its stores.
__device__ __forceinline__ void v8_fix_atomic(const GemmArgs& a, int* xb, int* wb, int cx, int cw, const float* sxt,
                                              const float* swt, long long m0, long long n0, int bm) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nitems = (cx + cw) * 4;
    for (int j = wave; j < nitems; j += V8_NW) {
        const bool is_x = (j >> 2) < cx;
        const int* e = v8_entry(xb, wb, cx, j >> 2);
        const int rl = (j & 3) * 64 + lane;
        const long long q = (is_x ? n0 : m0) + rl, r = e[0];
        if (e[3] == -2 || q >= (is_x ? +a.N : +a.M) || (!is_x && rl >= bm)) continue;
        const int4 qv = *reinterpret_cast<const int4*>((is_x ? +a.wm : +a.xm) + tiled_offset(q, (long long)e[1] * 16, a.K));
        const int d = dot16(*reinterpret_cast<const int4*>(e + 4), qv);
        if (d != 0)
            atomicAdd(&a.y[(is_x ? r : q) * a.ldy + (is_x ? q : r)],
                      __builtin_ldexpf((float)d, e[2] - (is_x ? +a.x_off : +a.w_off)) * (is_x ? swt : sxt)[rl]);
    }
    for (int idx = tid; idx < cx * cw; idx += V8_NT) {
        const int* e = xb + EXC_HEADER + EXC_ENTRY * (idx / cw);
        const int* f = wb + EXC_HEADER + EXC_ENTRY * (idx % cw);
        if (e[3] == -2 || f[3] == -2 || e[1] != f[1]) continue;
        const int d = dot16(*reinterpret_cast<const int4*>(e + 4), *reinterpret_cast<const int4*>(f + 4));
        if (d != 0) atomicAdd(&a.y[(long long)e[0] * a.ldy + f[0]], __builtin_ldexpf((float)d, e[2] + f[2] - a.scale_bias));
    }
}

// The FALLBACK of the launch: when an exception bucket overflowed somewhere (a row that could not store its exception
// blocks keeps its own exponents, rowflag 0) the row-scale product does not apply; the launch's workgroups then share the
// whole product with the blockwise-exact body (128 x 128 tiles, one 256-thread team per workgroup) and add each tile's
// exception blocks right after its stores.  One launch either way, and no extra workgroups in the common case (256 of
// them used to ride behind the tiles and leave at once: their dispatch alone cost the launch 1-2 us).
__device__ __forceinline__ void v8_fallback(const GemmArgs& a, const uint8_t* __restrict__ xf, const uint8_t* __restrict__ wf,
                                         const int* __restrict__ xlist, const int* __restrict__ wlist, unsigned char* smem,
                                         int wg, int nwg) {
    if (xlist[0] == 0 && wlist[0] == 0) return;
    if (threadIdx.x >= 256) return;                  // (terminated waves do not take part in the barriers below)
    const int ntiles = (int)(((a.M + V2_BM - 1) / V2_BM) * ((a.N + V2_BN - 1) / V2_BN));
    for (int tile = wg; tile < ntiles; tile += nwg) {
        bfp_gemm_v2_body(a, xf, wf, *reinterpret_cast<V2Smem*>(smem), tile);
        long long m0, n0;
        v2_tile_origin(a, tile, m0, n0);
        __threadfence();
        __syncthreads();
        tile_fix_body(a, row_bucket(xlist, m0, a.x_bcap), row_bucket(wlist, n0, a.w_bcap), a.x_bcap, a.w_bcap, m0, n0,
                      (int)threadIdx.x, 256);
        __syncthreads();
    }
}

// ---- split-K workspace: raw accumulator slabs + one ticket per tile, owned by the library, one per (device, stream),
//      grow-only; tickets are zero whenever no launch is in flight (the reducer of a tile clears its ticket).
struct SplitWorkspace {
    void* slabs = nullptr;
    int* tickets = nullptr;
    size_t slab_bytes = 0;
    int ntickets = 0;
};
SplitWorkspace* split_workspace(hipStream_t st, size_t slab_bytes, int ntickets);
int choose_splits(long long tiles, int nsteps_all, bool need_even, int min_steps = 8);
// the 256 x 256 tile (mi355q_gemm_v9.hip): K % 128 == 0, a.splits / a.slabs / a.tickets set by the caller
int launch_bfp_gemm_v9(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist, hipStream_t st,
                       const uint8_t* xf, const uint8_t* wf, bool bf16);

// the small tiles (mi355q_gemm_v10.hip; geometry 1: 128 x 256, 2: 256 x 128, 3: 128 x 128): K % 64 == 0, a.splits / a.slabs /
// a.tickets set by the caller (slabs of one tile's fp32 / int32 accumulators)
int launch_bfp_gemm_v10(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist, hipStream_t st,
                        const uint8_t* xf, const uint8_t* wf, bool bf16, int geom);
void v10_tile_shape(int geom, int& bm, int& bn);

}  // namespace mi355q
#endif
