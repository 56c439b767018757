// mi355q_gemm_post.hip -- ROW POST-PASS of the row-scale GEMM: adds the exception blocks of the ACTIVATION operand to y
// after mi355q_gemm_v8.hip stored it.
//
// An exception block (row r, block kb) of x contributes to the whole row r of y:
//     y[r, n] += 2^(code - x_off) * sw[n] * dot16(entry, wm'[n, kb])                      (w as stored)
//              + 2^(code + wcode - x_off - w_off) * dot16(entry, w-entry(n, kb))          (w's own exception there)
// Row r belongs to exactly one bucket and, inside a work item (bucket, 64 columns, quarter of the rows), to exactly one
// wave, which therefore owns the y elements it updates: plain read-add-write, no atomics.  The bucket's entries are
// ordered by (owner, row, block) first (counting sort in LDS); every y element receives ONE add of the sum of its
// row's terms taken in block order -- results do not depend on the order in which the align step reserved the entries.  Nothing limits the entries of a row or a tile here (the bucket holds
// up to 1016), which is what post-activation inputs (half zeros, wide spread of block exponents) need; the in-LDS
// vectors of the GEMM remain for the weights' few exceptions, which hit COLUMNS of y.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_fix.h"

namespace mi355q {

constexpr int RP_U = 8;                  // entries whose operand blocks / y elements are requested together

struct RowPostSmem {
    int cnt[256];                        // entries per row of the bucket
    int start[256];                      // first sorted position of the row
    int fill[256];
    unsigned kbits[MI355Q_ROW_ALIGN_MAX_K / 16 / 32];   // blocks in which some w row of this chunk has an exception
    int wsum[4];
    int wb[ROW_BUCKET_WORDS];            // w bucket of this column chunk
    unsigned short tmp[ROW_BCAP_MAX];
    unsigned short order[ROW_BCAP_MAX];
    int ent[EXC_ENTRY * ROW_BCAP_MAX];   // the x bucket's entries
};

__global__ __launch_bounds__(256) void bfp_gemm_rowpost(const GemmArgs a, const int* __restrict__ xlist,
                                                        const int* __restrict__ wlist, const float* __restrict__ wscale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rp_smem[];
    RowPostSmem& sm = *reinterpret_cast<RowPostSmem*>(rp_smem);
    if (xlist[0] != 0 || (wlist && wlist[0] != 0)) return;      // a bucket overflowed: the blockwise launch formed y
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // work item = (bucket, 64 columns, quarter of the bucket's rows); wave w of it owns the rows with
    // row % 16 == 4 * quarter + w.  Sorting by (row % 16, row / 16, block) makes every wave's entries one range.
    const int nbx = (int)((a.M + ROW_BUCKET_ROWS - 1) / ROW_BUCKET_ROWS), nch = (int)((a.N + 63) >> 6);
    for (int wi = blockIdx.x; wi < nbx * nch * 4; wi += gridDim.x) {
        const int bb = wi / (nch * 4), rem = wi - bb * (nch * 4), chunk = rem >> 2, quarter = rem & 3;
        const int* bk = row_bucket(xlist, (long long)bb * ROW_BUCKET_ROWS, a.x_bcap);
        const int cnt = __builtin_amdgcn_readfirstlane(min(bk[0], a.x_bcap));
        if (cnt == 0) continue;
        __syncthreads();
        for (int i = tid; i < cnt * 2; i += 256)
            reinterpret_cast<int4*>(sm.ent)[i] = reinterpret_cast<const int4*>(bk + EXC_HEADER)[i];
        sm.cnt[tid] = 0;
        sm.fill[tid] = 0;
        if (tid < (int)(sizeof(sm.kbits) / 4)) sm.kbits[tid] = 0u;
        int cw = 0;
        if (wlist) {
            const int* wbk = row_bucket(wlist, (long long)(chunk >> 2) * ROW_BUCKET_ROWS, a.w_bcap);
            cw = __builtin_amdgcn_readfirstlane(min(wbk[0], a.w_bcap));
            for (int i = tid; i < cw * 2; i += 256)
                reinterpret_cast<int4*>(sm.wb + EXC_HEADER)[i] = reinterpret_cast<const int4*>(wbk + EXC_HEADER)[i];
        }
        __syncthreads();
        auto bin_of = [](int r) { return ((r & 15) << 4) | ((r >> 4) & 15); };
        for (int i = tid; i < cnt; i += 256) {
            const int r = sm.ent[EXC_ENTRY * i];
            if (r >= 0) atomicAdd(&sm.cnt[bin_of(r)], 1);
        }
        for (int i = tid; i < cw; i += 256) {
            const int* f = sm.wb + EXC_HEADER + EXC_ENTRY * i;
            if (f[0] >= 0) atomicOr(&sm.kbits[f[1] >> 5], 1u << (f[1] & 31));
        }
        __syncthreads();
        {   // exclusive scan of cnt[256] -> start[256]
            int c = sm.cnt[tid], s = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(s, o);
                if (lane >= o) s += t;
            }
            if (lane == 63) sm.wsum[wave] = s;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += sm.wsum[w];
            sm.start[tid] = base + s - c;
        }
        __syncthreads();
        for (int i = tid; i < cnt; i += 256) {
            const int r = sm.ent[EXC_ENTRY * i];
            if (r >= 0) sm.tmp[sm.start[bin_of(r)] + atomicAdd(&sm.fill[bin_of(r)], 1)] = (unsigned short)i;
        }
        __syncthreads();
        const int total = sm.start[255] + sm.cnt[255];               // (void entries of overflowed rows never get here)
        for (int p = tid; p < total; p += 256) {                      // rank inside the row by block index
            const int i = sm.tmp[p];
            const int bin = bin_of(sm.ent[EXC_ENTRY * i]), kb = sm.ent[EXC_ENTRY * i + 1];
            const int s0 = sm.start[bin], s1 = s0 + sm.cnt[bin];
            int rank = 0;
            for (int q = s0; q < s1; ++q) rank += sm.ent[EXC_ENTRY * sm.tmp[q] + 1] < kb ? 1 : 0;
            sm.order[s0 + rank] = (unsigned short)i;
        }
        __syncthreads();

        const int own = (quarter * 4 + wave) << 4;                    // this wave's 16 bins
        const int p0 = sm.start[own], p1 = sm.start[own + 15] + sm.cnt[own + 15];
        const long long n = (long long)chunk * 64 + lane;
        const bool nok = n < a.N;
        const long long nc = nok ? n : a.N - 1;
        const float sc = nok ? wscale[n] : 0.f;
        float acc = 0.f;
        for (int e0 = p0; e0 < p1; e0 += RP_U) {                     // (wave-uniform bounds)
            int4 qv[RP_U];
            float yv[RP_U];
#pragma unroll
            for (int u = 0; u < RP_U; ++u) {
                const int* e = sm.ent + EXC_ENTRY * sm.order[min(e0 + u, p1 - 1)];
                // (unconditional, columns behind N read column N - 1: a load inside a branch makes the counted waits drain
                // everything in flight)
                qv[u] = *reinterpret_cast<const int4*>(a.wm + tiled_offset(nc, (long long)e[1] * 16, a.K));
                yv[u] = a.y[(long long)e[0] * a.ldy + nc];
            }
#pragma unroll
            for (int u = 0; u < RP_U; ++u) {
                if (e0 + u >= p1) break;
                const int* e = sm.ent + EXC_ENTRY * sm.order[e0 + u];
                const int row = e[0], kb = e[1], code = e[2];
                const int4 pv = *reinterpret_cast<const int4*>(e + 4);
                acc += __builtin_ldexpf((float)dot16(pv, qv[u]), code - a.x_off) * sc;
                if (sm.kbits[kb >> 5] & (1u << (kb & 31))) {          // uniform: some w row has its own exception at kb
                    for (int t = 0; t < cw; ++t) {
                        const int* f = sm.wb + EXC_HEADER + EXC_ENTRY * t;
                        if (f[1] == kb && f[0] == (int)n)
                            acc += __builtin_ldexpf((float)dot16(pv, *reinterpret_cast<const int4*>(f + 4)),
                                                    code + f[2] - a.scale_bias);
                    }
                }
                const bool last = e0 + u == p1 - 1 || sm.ent[EXC_ENTRY * sm.order[e0 + u + 1]] != row;
                if (last) {
                    if (nok) a.y[(long long)row * a.ldy + n] = yv[u] + acc;
                    acc = 0.f;
                }
            }
        }
    }
}

int launch_bfp_gemm_rowpost(const GemmArgs& a, const int* xlist, const int* wlist, const float* xscale, const float* wscale,
                            hipStream_t st) {
    (void)xscale;
    const long long nbx = (a.M + ROW_BUCKET_ROWS - 1) / ROW_BUCKET_ROWS, items = nbx * ((a.N + 63) / 64) * 4;
    const unsigned grid = (unsigned)(items > 8192 ? 8192 : items);
    hipLaunchKernelGGL(bfp_gemm_rowpost, grid, 256, sizeof(RowPostSmem), st, a, xlist, wlist, wscale);
    return (int)hipGetLastError();
}

}  // namespace mi355q
