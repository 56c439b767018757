// mi355q_fix.h -- exact add-back of the exception blocks the align step took out of an aligned operand
// (mi355q_align.h).  An exception block (row r, block kb) of x contributes
//     y[r, n] += 2^(code + we[n, kb] - bias) * dot16(mant, wm[n, kb])        for every n,
// where (wm, we) is w as stored (its own exception blocks are zero there), plus the exception x exception
// products of the same kb, taken from the two lists.  A w exception adds the mirror image over m, against x
// as stored.  Every term is an exact integer dot times a power of two; fp32 atomics add them to y.
#ifndef MI355Q_FIX_H
#define MI355Q_FIX_H
#include "mi355q_align.h"
#include "mi355q_align_row.h"
#include "mi355q_gemm_v2.h"

namespace mi355q {

__device__ __forceinline__ int dot16(const int4& p, const int4& q) {
    int d = __builtin_amdgcn_sdot4(q.x, p.x, 0, false);
    d = __builtin_amdgcn_sdot4(q.y, p.y, d, false);
    d = __builtin_amdgcn_sdot4(q.z, p.z, d, false);
    return __builtin_amdgcn_sdot4(q.w, p.w, d, false);
}

__device__ __forceinline__ int list_count(const int* __restrict__ list, int cap) {
    return list ? min(list[0], cap) : 0;
}

// exception bucket of a row-aligned operand that covers row r0 (mi355q_align_row.h); same layout as a list
__device__ __forceinline__ const int* row_bucket(const int* __restrict__ list, long long r0, int bcap = ROW_BCAP) {
    return list ? list + EXC_HEADER + (r0 / ROW_BUCKET_ROWS) * row_bucket_words(bcap) : nullptr;
}

// One exception entry against rows [q0, q1) of the other operand; `lane_id`/`nlanes` = the threads that share it.
__device__ __forceinline__ void fix_entry_sweep(const GemmArgs& a, bool is_x, int row, int kb, int code, const int4& pv,
                                                long long q0, long long q1, int lane_id, int nlanes) {
    const long long nkb = a.K >> 4;
    // (both fields read first: a conditional over `a.we` / `a.xe` themselves becomes a select of addresses inside `a`,
    //  which keeps a caller's by-value copy of the argument block in scratch memory)
    const int8_t *w_m = a.wm, *x_m = a.xm;
    const uint8_t *w_e = a.we, *x_e = a.xe;
    const int8_t* qm = is_x ? w_m : x_m;
    const uint8_t* qe = is_x ? w_e : x_e;
    for (long long q = q0 + lane_id; q < q1; q += nlanes) {
        const int4 qv = *reinterpret_cast<const int4*>(qm + tiled_offset(q, (long long)kb * 16, a.K));
        const int ecode = (int)qe[q * nkb + kb];
        const int d = dot16(pv, qv);
        if (d != 0) {
            const long long m = is_x ? row : q, n = is_x ? q : row;
            atomicAdd(&a.y[m * a.ldy + n], __builtin_ldexpf((float)d, code + ecode - a.scale_bias));
        }
    }
}

// exception(x) x exception(w) products of one x entry, w rows restricted to [n0, n1)
__device__ __forceinline__ void fix_entry_cross(const GemmArgs& a, int row, int kb, int code, const int4& pv,
                                                const int* __restrict__ wlist, int cw, long long n0, long long n1,
                                                int lane_id, int nlanes) {
    for (int t = lane_id; t < cw; t += nlanes) {
        const int* f = wlist + EXC_HEADER + EXC_ENTRY * t;
        const int n = f[0];
        if (n < n0 || n >= n1 || f[1] != kb) continue;
        const int4 wv = *reinterpret_cast<const int4*>(f + 4);
        const int d = dot16(pv, wv);
        if (d != 0) atomicAdd(&a.y[(long long)row * a.ldy + n], __builtin_ldexpf((float)d, code + f[2] - a.scale_bias));
    }
}

// Whole-matrix correction spread over `nwg` workgroups: work item = (entry, 256 rows of the other operand).
__device__ __forceinline__ void block_fix_body(const GemmArgs& a, const int* __restrict__ xlist,
                                               const int* __restrict__ wlist, int cap, int wg, int nwg) {
    const int cx = list_count(xlist, cap), cw = list_count(wlist, cap);
    const int xch = (int)((a.N + 255) >> 8), wch = (int)((a.M + 255) >> 8);
    const int xitems = cx * xch, total = xitems + cw * wch;
    for (int it = wg; it < total; it += nwg) {                       // uniform over the workgroup
        const bool is_x = it < xitems;
        const int j = is_x ? it : it - xitems, nch = is_x ? xch : wch;
        const int ent = j / nch, chunk = j - ent * nch;
        const int* e = (is_x ? xlist : wlist) + EXC_HEADER + EXC_ENTRY * ent;
        const int row = e[0], kb = e[1], code = e[2];
        if (row < 0) continue;
        const int4 pv = *reinterpret_cast<const int4*>(e + 4);
        const long long qrows = is_x ? a.N : a.M, q0 = (long long)chunk * 256;
        fix_entry_sweep(a, is_x, row, kb, code, pv, q0, min(q0 + 256, qrows), threadIdx.x, 256);
        if (is_x && chunk == 0 && cw > 0) fix_entry_cross(a, row, kb, code, pv, wlist, cw, 0, a.N, threadIdx.x, 256);
    }
}

// Correction of ONE output tile [m0, m0+BM) x [n0, n0+BN) by the workgroup that owns it (blockwise-fallback
// path: runs after the tile's own stores, behind a workgroup barrier).
template <int BM = V2_BM, int BN = V2_BN>
__device__ __forceinline__ void tile_fix_body(const GemmArgs& a, const int* __restrict__ xlist,
                                              const int* __restrict__ wlist, int capx, int capw, long long m0,
                                              long long n0, int tid = -1, int nthreads = 0) {
    const int cx = list_count(xlist, capx), cw = list_count(wlist, capw);
    if (cx == 0 && cw == 0) return;
    if (tid < 0) { tid = threadIdx.x; nthreads = blockDim.x; }       // default: the whole workgroup is the team
    const int lane = tid & 63;
    const long long m1 = min(m0 + BM, a.M), n1 = min(n0 + BN, a.N);
    for (int pass = 0; pass < 2; ++pass) {
        const bool is_x = pass == 0;
        const int* list = is_x ? xlist : wlist;
        const int cnt = is_x ? cx : cw;
        const long long r0 = is_x ? m0 : n0, r1 = is_x ? m1 : n1, q0 = is_x ? n0 : m0, q1 = is_x ? n1 : m1;
        for (int base = 0; base < cnt; base += nthreads) {            // uniform
            const int t = base + tid;
            int row = -1;
            if (t < cnt) row = list[EXC_HEADER + EXC_ENTRY * t];
            unsigned long long hit = __ballot(row >= r0 && row < r1);
            while (hit) {                                              // the wave works through its own hits
                const int src = __builtin_ctzll(hit);
                hit &= hit - 1;
                const int* e = list + EXC_HEADER + EXC_ENTRY * (base + (tid & ~63) + src);
                const int er = e[0], kb = e[1], code = e[2];
                const int4 pv = *reinterpret_cast<const int4*>(e + 4);
                fix_entry_sweep(a, is_x, er, kb, code, pv, q0, q1, lane, 64);
                if (is_x && cw > 0) fix_entry_cross(a, er, kb, code, pv, wlist, cw, n0, n1, lane, 64);
            }
        }
    }
}

}  // namespace mi355q
#endif
