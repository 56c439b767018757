// mi355q_corr.h -- exception corrections of the row-scale product formed WHERE THE OPERANDS ARE PRODUCED (round 4).
//
// The row-scale int8 GEMM (mi355q_gemm_v9.hip) multiplies row-aligned operands whose exception blocks are zero; what the
// exception blocks contribute (quantized_modules/linear.py:59-76: F.linear on the fake-quantised operands, every block
// with its own exponent) is a sparse rank-one-like update:
//     y[m, n] += xvec[m][n]   for the rows m of x that carry exception blocks      (x entries x aligned W)
//     y[m, n] += wvec[n][m]   for the rows n of W that carry exception blocks      (W entries x TRUE x, cross terms included)
// Seven schedules of forming those vectors inside the product launch cost ~10 us of a 66-us kernel (round 3,
// profiles/r03_v9_exception_designs.txt).  Here the producers form them:
//   * W's entries are static: mi355q_bfp_corr_plan sorts each 256-row bucket's entries by (row, block) once, gives every
//     W row with entries a COLUMN SLOT (< CORR_WV per bucket) and writes the tile's column map;
//   * the activation quantiser (bfp_quant_align_rows_kernel<..., CORR>) holds a whole row of x: for every column slot of W
//     it writes  wvec[m][bucket][slot] = sum over the slot's entries of 2^(..) dot16(entry, x's TRUE block)  (16 dot
//     products per W entry and row), and, for the few rows that have exception blocks of their own, takes a ROW SLOT
//     (< CORR_XV per 256-row bucket, reserved in word 1 of the bucket header) and writes the row's vector against the
//     aligned W operand  xvec[bucket][slot][n], n = 0 .. N-1;
//   * the product launch reads its tile's two maps, <= 16 row vectors of 1 KiB and one 16-KiB block of column values into
//     spare LDS behind its first K-steps and adds them in its store epilogue.  No gathers, no bookkeeping, no chains.
// More rows / columns with exceptions than slots: a word of x's list header (word 1) / of the plan (word 0) says so and the
// product launch forms its add-back itself as before (decided on the device, uniform over the grid).
#ifndef MI355Q_CORR_H
#define MI355Q_CORR_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q_align_row.h"

namespace mi355q {

constexpr int CORR_XV = 32;      // rows with exception blocks per 256-row bucket of x that get a vector
constexpr int CORR_WV = 16;      // rows with exception blocks per 256-row bucket of W that get a column slot
constexpr int CORR_MAX_W = 3;    // weight operands one activation operand is prepared for (q / k / v, gate / up)

// ---- W's correction plan, int32 words:
//   [0] 1 = not usable (a bucket overflowed, or holds more than CORR_WV distinct rows)   [1] column slots in use, all buckets
//   [2] buckets   [3] N   [4..15] spare
//   colmap [npad]: column slot of W row n inside its bucket, -1 none
//   ncols  [nb16]: column slots in use per bucket
//   slots  [nb][CORR_WV][4]: {row n (-1 unused), first entry (index into the bucket's sorted entries), entries, 0}
//   entries[nb][ROW_BCAP][8]: the bucket's entries sorted by (row, block), entry layout of mi355q_align.h
//   dense  [max(nb * CORR_WV, 256)][8]: one record per column slot IN USE, in (bucket, slot) order -- what a row workgroup of
//          the activation quantiser walks: {bucket * CORR_WV + slot, block, exponent, further entries of the row << 16 | index
//          of the first entry in the bucket's sorted entries, 16 mantissa bytes of the first entry}; records past [1] unused
constexpr int PLAN_HDR = 16;
__host__ __device__ inline long long corr_pad256(long long n) { return (n + 255) / 256 * 256; }
__host__ __device__ inline long long plan_nb(long long N) { return (N + ROW_BUCKET_ROWS - 1) / ROW_BUCKET_ROWS; }
__host__ __device__ inline long long plan_colmap_off() { return PLAN_HDR; }
__host__ __device__ inline long long plan_ncols_off(long long N) { return PLAN_HDR + corr_pad256(N); }
__host__ __device__ inline long long plan_slots_off(long long N) { return plan_ncols_off(N) + (plan_nb(N) + 15) / 16 * 16; }
__host__ __device__ inline long long plan_entries_off(long long N) { return plan_slots_off(N) + plan_nb(N) * CORR_WV * 4; }
__host__ __device__ inline long long plan_dense_off(long long N) { return plan_entries_off(N) + plan_nb(N) * ROW_BCAP * EXC_ENTRY; }
__host__ __device__ inline long long plan_dense_records(long long N) { return plan_nb(N) * CORR_WV > 256 ? plan_nb(N) * CORR_WV : 256; }
__host__ __device__ inline long long plan_words(long long N) { return plan_dense_off(N) + plan_dense_records(N) * 8; }

// ---- vectors of one (x, W) pair, fp32:
//   xvec [mb][CORR_XV][npad]   (mb = 256-row buckets of x, npad = N padded to 256)
//   wvec [mpad][nb][CORR_WV]   (nb = 256-row buckets of W, mpad = M padded to 256): a row's values are contiguous -- one
//                              workgroup of the quantiser owns every line it writes
__host__ __device__ inline long long corr_xvec_floats(long long M, long long N) { return plan_nb(M) * CORR_XV * corr_pad256(N); }
__host__ __device__ inline long long corr_wvec_floats(long long M, long long N) { return plan_nb(N) * corr_pad256(M) * CORR_WV; }

// what the activation quantiser needs to form the vectors and the product launch to read them: ONE struct in device
// memory per (activation buffers, weight set), written once (mi355q_bfp_corr_bind) -- as a kernel argument its thirty
// pointers cost the quantiser's hot path scalar registers (spilled to vector lanes) and a wave of occupancy
struct CorrArgs {
    int count;                       // weight operands (0: no corrections formed)
    int x_off, w_off;                // exponent_bias + mbits of x / of the weights (2^(code - off) = a block's scale)
    int* rowmap;                     // [mpad] out: the row's slot in its bucket, -1 none
    long long mpad;
    long long N[CORR_MAX_W];
    const int8_t* wm[CORR_MAX_W];    // tiled aligned mantissas of W
    const float* sw[CORR_MAX_W];     // W's row scales
    const int* plan[CORR_MAX_W];
    float* xvec[CORR_MAX_W];
    float* wvec[CORR_MAX_W];
};

// what the product launch reads (one set per weight of a grouped launch)
struct CorrRead {
    const int* rowmap;
    const int* plan[CORR_MAX_W];
    const float* xvec[CORR_MAX_W];
    const float* wvec[CORR_MAX_W];
};

// a launch of the activation quantiser with a binding: the device-resident struct plus what the kernel wants as arguments
struct CorrLaunch {
    const CorrArgs* dev;
    const int* plan0;            // the first weight's plan
    const char* dense0;          // ... its dense records
    int ntab;                    // dense records to fetch into LDS (multiple of 32, <= 256)
    int dbg;                     // diagnostic (MI355Q_CORR_DBG): 1 no prefetch, 2 no staging, 4 no corr_row, 8 no row vectors, 16 no column values
};

int launch_corr_plan(const int* wlist, long long N, int* plan, hipStream_t st);

#ifdef __HIPCC__
__device__ __forceinline__ int corr_dot16(const int4& p, const int4& q) {
    int d = __builtin_amdgcn_sdot4(q.x, p.x, 0, false);
    d = __builtin_amdgcn_sdot4(q.y, p.y, d, false);
    d = __builtin_amdgcn_sdot4(q.z, p.z, d, false);
    return __builtin_amdgcn_sdot4(q.w, p.w, d, false);
}
// byte offset of the 16-byte block kb of `row` in a tiled operand of K bytes per row (tiled_offset of mi355q_gemm_v2.h)
__device__ __forceinline__ long long corr_block_off(long long row, int kb, long long K) {
    return ((row >> 4) * (K >> 6) + (kb >> 2)) * 1024 + (kb & 3) * 256 + (row & 15) * 16;
}

#ifndef CORR_XB
#define CORR_XB 8     // columns per thread whose gathers fly together in a row's vector
#endif
constexpr int CORR_TAB = 256;      // dense records of the first weight's plan a row workgroup keeps in LDS

// LDS of the activation quantiser's correction part (bfp_quant_align_rows_kernel<..., CORR>)
template <int MAXIT>
struct CorrSmem {
    alignas(16) unsigned blk[MAXIT * 256];       // the row: block kb (16 mantissa bytes) at blk[4 kb .. 4 kb + 3]
    alignas(16) int tab[CORR_TAB * 8];           // dense records 0 .. CORR_TAB-1 of plan[0] (LDS-DMA at kernel start)
    alignas(16) int args[64];                    // the binding (CorrArgs, 256 bytes; LDS-DMA at kernel start)
    alignas(16) int phdr[64];                    // header words of plan[0] (LDS-DMA at kernel start)
    unsigned long long exc[MAXIT * 4];           // the row's exception blocks, one ballot per (slab, wave)
    unsigned char eff[MAXIT * 64];               // effective exponent of block kb
    int slot;                                    // the row's vector slot (-1 none)
};
static_assert(sizeof(CorrArgs) <= 256, "the binding is staged in 256 bytes of LDS");

// kernel start: the binding, the first weight's plan header and its first dense records on their way into LDS (no
// registers held, no dependent load in front: every address is a kernel argument; the round trips hide behind the row's own)
template <int MAXIT>
__device__ __forceinline__ void corr_prefetch(const CorrLaunch& cl, CorrSmem<MAXIT>& sm) {
    using gptr_t = const __attribute__((address_space(1))) void*;
    using lptr_t = __attribute__((address_space(3))) void*;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid * 2 < cl.ntab * 4)       // (16 bytes a lane: records 0 .. 127 by the first instruction of the four waves)
        __builtin_amdgcn_global_load_lds((gptr_t)(cl.dense0 + tid * 16), (lptr_t)(reinterpret_cast<char*>(sm.tab) + wave * 1024), 16, 0, 0);
    if (cl.ntab > 128)
        __builtin_amdgcn_global_load_lds((gptr_t)(cl.dense0 + 4096 + tid * 16), (lptr_t)(reinterpret_cast<char*>(sm.tab) + 4096 + wave * 1024), 16, 0, 0);
    if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const int*>(cl.dev) + lane), (lptr_t)sm.args, 4, 0, 0);
    if (wave == 1) __builtin_amdgcn_global_load_lds((gptr_t)(cl.plan0 + lane), (lptr_t)sm.phdr, 4, 0, 0);
}

// one column slot's value for this row: the slot's entries (ascending block) against the row's blocks as staged
__device__ __forceinline__ float corr_slot_value(const int4& head, const int4& mant, const int* __restrict__ entries,
                                                 const unsigned* blk, const unsigned char* eff, int off) {
    const int kb = head.y;
    float acc = __builtin_ldexpf((float)corr_dot16(mant, *reinterpret_cast<const int4*>(&blk[kb * 4])), (int)eff[kb] + head.z - off);
    const int more = head.w >> 16;
    if (more) {                                                    // (a W row with several exception blocks: rare)
        const int* e = entries + ((long long)(head.x / CORR_WV) * ROW_BCAP + (head.w & 0xffff) + 1) * EXC_ENTRY;
#pragma unroll 1
        for (int j = 0; j < more; ++j, e += EXC_ENTRY) {
            const int k2 = e[1];
            acc += __builtin_ldexpf((float)corr_dot16(*reinterpret_cast<const int4*>(e + 4), *reinterpret_cast<const int4*>(&blk[k2 * 4])),
                                    (int)eff[k2] + e[2] - off);
        }
    }
    return acc;
}

// Behind the row's stores, all 256 threads.  The caller has staged the row in sm.blk / sm.eff (blocks shifted onto the row's
// exponent, or as quantised with their own exponent where they became exceptions).  Thread t (wave w, lane l) holds blocks
// kb = 64 it + 16 w + l / 4; excb bit `it`: that block is an exception.  Writes the row's value for every column slot of the
// weights' plans, takes the row's vector slot (sm.slot, rowmap) if the row has exception blocks.  Returns that slot (-1
// none): the caller forms the vector at the very end of the kernel (corr_xvec), where no register of the row's arithmetic
// is live any more.
template <int MAXIT>
__device__ __forceinline__ int corr_row(CorrSmem<MAXIT>& sm, unsigned excb, bool flagged, long long row, int* __restrict__ list, int bcap, int dbg = 0) {
    const CorrArgs& c = *reinterpret_cast<const CorrArgs*>(sm.args);     // (the LDS copy)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool any = false;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const unsigned long long bal = __ballot(((excb >> it) & 1u) && (lane & 3) == 0);
        if (lane == 0) sm.exc[it * 4 + wave] = bal;
        any = any || bal != 0ull;
    }
    const int has_exc = __syncthreads_or(any ? 1 : 0);            // (also: the row's LDS image is complete)
    const int count = c.count;
    if (!flagged) {                                                // the row kept its own exponents: the launch that multiplies
        if (tid == 0) { c.rowmap[row] = -1; sm.slot = -1; }        // it takes its blockwise path, nothing here is read
        __syncthreads();
        return -1;
    }
    // every row: its products with the weights' exception blocks, one value per column slot of W in use
    //     wvec[bucket of n][row][slot] = sum over the slot's entries (ascending block) of 2^(eff + code_w - x_off - w_off)
    //                                    dot16(entry, x's block as staged)
    const int off = c.x_off + c.w_off;
    {
        const int nused = (dbg & 16) ? 0 : sm.phdr[1];
        if (nused) {
            const int* __restrict__ entries = c.plan[0] + plan_entries_off(c.N[0]);
            float* __restrict__ out = c.wvec[0];
            const long long nbw = plan_nb(c.N[0]) * CORR_WV;
            if (tid < nused && tid < CORR_TAB) {
                const int4 head = *reinterpret_cast<const int4*>(&sm.tab[tid * 8]), mant = *reinterpret_cast<const int4*>(&sm.tab[tid * 8 + 4]);
                out[row * nbw + head.x] = corr_slot_value(head, mant, entries, sm.blk, sm.eff, off);
            }
            if (nused > CORR_TAB) {
                const int4* __restrict__ dense = reinterpret_cast<const int4*>(c.plan[0] + plan_dense_off(c.N[0]));
#pragma unroll 1
                for (int idx = tid + CORR_TAB; idx < nused; idx += 256) {
                    const int4 head = dense[idx * 2], mant = dense[idx * 2 + 1];
                    out[row * nbw + head.x] = corr_slot_value(head, mant, entries, sm.blk, sm.eff, off);
                }
            }
        }
    }
#pragma unroll 1
    for (int g = 1; g < count; ++g) {                              // (further weights of a group: records straight from memory)
        const int* __restrict__ plan = c.plan[g];
        const int nused = plan[1];
        const long long N = c.N[g];
        const int4* __restrict__ dense = reinterpret_cast<const int4*>(plan + plan_dense_off(N));
        const int* __restrict__ entries = plan + plan_entries_off(N);
        float* __restrict__ out = c.wvec[g];
        const long long nbw = plan_nb(N) * CORR_WV;
#pragma unroll 1
        for (int idx = tid; idx < nused; idx += 256) {
            const int4 head = dense[idx * 2], mant = dense[idx * 2 + 1];
            out[row * nbw + head.x] = corr_slot_value(head, mant, entries, sm.blk, sm.eff, off);
        }
    }
    if (tid == 0) {
        int s = -1;
        if (has_exc) {
            int* bucket = list + EXC_HEADER + (row / ROW_BUCKET_ROWS) * row_bucket_words(bcap);
            s = atomicAdd(&bucket[1], 1);
            if (s >= CORR_XV) { atomicAdd(&list[1], 1); s = -1; }
        }
        sm.slot = s;
        c.rowmap[row] = s;
    }
    __syncthreads();
    return sm.slot;
}

// A row with exception blocks of its own: its vector against every aligned weight operand of the binding,
//     xvec[n] = sw[n] * sum over the row's exception blocks (ascending block) of 2^(code - x_off) dot16(block, w'[n, kb]),
// CORR_XB columns per thread at a time (their gathers of W's blocks fly together: one round trip per 256 CORR_XB columns).
template <int MAXIT>
__device__ __forceinline__ void corr_xvec(CorrSmem<MAXIT>& sm, long long K, long long row, int slot) {
    const CorrArgs& c = *reinterpret_cast<const CorrArgs*>(sm.args);
    const int tid = threadIdx.x;
    const int count = c.count, x_off = c.x_off;
#pragma unroll 1
    for (int g = 0; g < count; ++g) {
        const int N = (int)c.N[g], npad = (int)corr_pad256(N);
        const int8_t* __restrict__ wm = c.wm[g];
        const float* __restrict__ sw = c.sw[g];
        float* __restrict__ out = c.xvec[g] + ((row / ROW_BUCKET_ROWS) * CORR_XV + slot) * (long long)npad;
#pragma unroll 1
        for (int nb0 = 0; nb0 < npad; nb0 += 256 * CORR_XB) {
            float acc[CORR_XB];
#pragma unroll
            for (int j = 0; j < CORR_XB; ++j) acc[j] = 0.f;
#pragma unroll 1
            for (int q = 0; q < MAXIT * 4; ++q) {           // (uniform: the ballots are in LDS)
                const unsigned long long bq = sm.exc[q];
                unsigned long long bal = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bq >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)bq);
                while (bal) {
                    const int bit = __builtin_ctzll(bal);
                    bal &= bal - 1ull;
                    const int kb = (q >> 2) * 64 + (q & 3) * 16 + (bit >> 2);
                    const int4 pv = *reinterpret_cast<const int4*>(&sm.blk[kb * 4]);
                    const int sh = (int)sm.eff[kb] - x_off;
                    int4 wv[CORR_XB];
#pragma unroll
                    for (int j = 0; j < CORR_XB; ++j) {
                        const int n = nb0 + tid + 256 * j;
                        wv[j] = n < N ? *reinterpret_cast<const int4*>(wm + corr_block_off(n, kb, K)) : int4{0, 0, 0, 0};
                    }
#pragma unroll
                    for (int j = 0; j < CORR_XB; ++j) acc[j] += __builtin_ldexpf((float)corr_dot16(pv, wv[j]), sh);
                }
            }
#pragma unroll
            for (int j = 0; j < CORR_XB; ++j) {
                const int n = nb0 + tid + 256 * j;
                if (n < npad) out[n] = n < N ? acc[j] * sw[n] : 0.f;
            }
        }
    }
}
#endif

}  // namespace mi355q
#endif
