// mi355q_align_row.h -- ROW alignment: one exponent for a whole row of an operand (all K/16 blocks), decided by
// one 256-thread workgroup per row.  Shared by the fused activation kernel (mi355q_quant.hip) and the
// packed-operand kernel (mi355q_gemm.hip); both must produce identical results.
//
// Thread t of the workgroup (wave w = t / 64, lane l = t % 64) holds, for it = 0 .. nit-1, the 4 mantissas
// (packed int8 x 4) at k = 16 * kb + 4 * (l & 3) of block kb = 64 * it + 16 * w + l / 4, plus that block's largest
// |mantissa| and biased exponent.  Rules (the same as for 256-value groups, mi355q_align.h):
//   * E = smallest exponent of a non-zero block if every block can be shifted left onto it inside int8;
//   * otherwise E = the exponent whose window [E, E + head-room(block)] holds the most blocks (smallest such E);
//     the other non-zero blocks are EXCEPTIONS: zeroed in the operand, listed exactly in the row's bucket;
//   * a row whose exceptions do not fit its bucket stays as it was (rowflag 0, scale 0, own exponents) and bumps
//     the overflow word list[0]: the GEMM then takes its blockwise-exact kernel.
// Exception list of a row-aligned operand (int32 words):
//   [0] rows that could not store their exceptions (0 = the fast GEMM applies), [1..7] spare,
//   then one bucket per 256 rows (bucket b covers rows 256 b .. 256 b + 255), 8 + 8 * bcap words each
//   (bcap = the operand's entries per bucket, 120 unless stated):
//   [0] entries reserved, [1..7] spare, then bcap entries of 8 words
//   {row (-1 = void), block, exponent, 0, 16 mantissa bytes}  -- the entry layout of mi355q_align.h.
#ifndef MI355Q_ALIGN_ROW_H
#define MI355Q_ALIGN_ROW_H
#include <hip/hip_runtime.h>
#include "mi355q_align.h"

namespace mi355q {

constexpr int ROW_BUCKET_ROWS = 256, ROW_BCAP = 120;
constexpr int ROW_BUCKET_WORDS = EXC_HEADER + EXC_ENTRY * ROW_BCAP;

// ROW_BCAP is the bucket size the row-scale GEMM can take into LDS (weights; activations of the in-LDS add-back).
// An operand whose exceptions are added by the row post-pass instead (mi355q_gemm_post.hip) may use larger buckets:
// every function below takes the operand's entries-per-bucket `bcap`.
constexpr int ROW_BCAP_MAX = 1016;
__host__ __device__ inline long long row_bucket_words(int bcap) { return EXC_HEADER + (long long)EXC_ENTRY * bcap; }
__host__ __device__ inline long long row_list_words(long long rows, int bcap = ROW_BCAP) {
    return EXC_HEADER + ((rows + ROW_BUCKET_ROWS - 1) / ROW_BUCKET_ROWS) * row_bucket_words(bcap);
}

struct RowAlignSmem {
    int emin[4];
    int wkey[4];
    int code0;
    int nexc;
    int base;
    int cnt[256];
};

// per-byte left shift of 4 packed int8 that are known not to overflow
__device__ __forceinline__ unsigned shl_packed(unsigned pk, int s) {
    return (pk << s) & (0x01010101u * (0xFFu & ~((1u << s) - 1u)));
}

// Returns true when the row carries one exponent (E); pk[] is rewritten in place (shifted / zeroed).
// When it returns false nothing was changed.  All 256 threads must call it (workgroup barriers inside).
// MAPPED (round 6, the class-aware quantiser of the mixed contraction): the block index an exception entry records is kbd[it]
// -- the block's position in the operand it is written to -- instead of its position in the row it was read from
template <int MAXIT, bool FULL, bool MAPPED>
__device__ __forceinline__ bool align_row_impl(unsigned (&pk)[MAXIT], const int (&amax)[MAXIT], const int (&code)[MAXIT], int nit,
                                               int nkb, long long row, int* __restrict__ list, RowAlignSmem& sm, int& E,
                                               int bcap, const int (&kbd)[MAXIT]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int BIG = 1 << 20;
    bool has[MAXIT];
    int head[MAXIT];
    int em = BIG, lo = -BIG;            // smallest exponent; largest "lowest exponent a block can be shifted onto"
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int kb = it * 64 + wave * 16 + (lane >> 2);
        has[it] = (FULL || (it < nit && kb < nkb)) && amax[it] > 0;
        head[it] = has[it] ? __clz(amax[it]) - 25 : 0;
        em = min(em, has[it] ? code[it] : BIG);
        lo = max(lo, has[it] ? code[it] - head[it] : -BIG);
    }
    em = min(em, __builtin_amdgcn_mov_dpp(em, 0x121, 0xF, 0xF, true));   // row_ror:1
    em = min(em, __builtin_amdgcn_mov_dpp(em, 0x122, 0xF, 0xF, true));   // row_ror:2
    em = min(em, __builtin_amdgcn_mov_dpp(em, 0x124, 0xF, 0xF, true));   // row_ror:4
    em = min(em, __builtin_amdgcn_mov_dpp(em, 0x128, 0xF, 0xF, true));   // row_ror:8
    em = min(min(__builtin_amdgcn_readlane(em, 0), __builtin_amdgcn_readlane(em, 16)),
             min(__builtin_amdgcn_readlane(em, 32), __builtin_amdgcn_readlane(em, 48)));
    lo = max(lo, __builtin_amdgcn_mov_dpp(lo, 0x121, 0xF, 0xF, true));
    lo = max(lo, __builtin_amdgcn_mov_dpp(lo, 0x122, 0xF, 0xF, true));
    lo = max(lo, __builtin_amdgcn_mov_dpp(lo, 0x124, 0xF, 0xF, true));
    lo = max(lo, __builtin_amdgcn_mov_dpp(lo, 0x128, 0xF, 0xF, true));
    lo = max(max(__builtin_amdgcn_readlane(lo, 0), __builtin_amdgcn_readlane(lo, 16)),
             max(__builtin_amdgcn_readlane(lo, 32), __builtin_amdgcn_readlane(lo, 48)));
    if (lane == 0) { sm.emin[wave] = em; sm.wkey[wave] = lo; }
    if (tid == 0) { sm.code0 = code[0]; sm.nexc = 0; }
    sm.cnt[tid] = 0;
    __syncthreads();
    const int emin = min(min(sm.emin[0], sm.emin[1]), min(sm.emin[2], sm.emin[3]));
    const int lomax = max(max(sm.wkey[0], sm.wkey[1]), max(sm.wkey[2], sm.wkey[3]));
    const int code0 = sm.code0;
    // every block can be shifted onto emin  <=>  emin >= code - head-room for every block  (one barrier decides)
    if (emin >= lomax) {
        E = emin == BIG ? code0 : emin;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
            if (has[it]) pk[it] = shl_packed(pk[it], code[it] - E);
        return true;
    }
    // ---- rare path: the exponent that keeps the most blocks (each block votes for every E it can join)
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
        if (has[it] && (lane & 3) == 0)
            for (int e = max(code[it] - head[it], 0); e <= code[it]; ++e) atomicAdd(&sm.cnt[e & 255], 1);
    __syncthreads();
    int key = (sm.cnt[tid] << 8) | (255 - tid);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) key = max(key, __shfl_xor(key, o));
    if (lane == 0) sm.wkey[wave] = key;
    __syncthreads();
    key = max(max(sm.wkey[0], sm.wkey[1]), max(sm.wkey[2], sm.wkey[3]));
    const int best = 255 - (key & 255);
    bool exc[MAXIT];
    int slot[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        exc[it] = has[it] && !(code[it] >= best && code[it] - best <= head[it]);
        slot[it] = 0;
        if (exc[it] && (lane & 3) == 0) slot[it] = atomicAdd(&sm.nexc, 1);
        slot[it] = __shfl(slot[it], lane & ~3);
    }
    __syncthreads();
    const int k = sm.nexc;
    int* bucket = list ? list + EXC_HEADER + (row / ROW_BUCKET_ROWS) * row_bucket_words(bcap) : nullptr;
    if (tid == 0) sm.base = bucket ? atomicAdd(&bucket[0], k) : bcap;
    __syncthreads();
    const int base = sm.base;
    if (base + k > bcap) {
        if (tid == 0 && list) atomicAdd(&list[0], 1);
        if (bucket) {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it)
                if (exc[it] && (lane & 3) == 0 && base + slot[it] < bcap)
                    bucket[EXC_HEADER + EXC_ENTRY * (base + slot[it])] = -1;
        }
        return false;
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if (exc[it]) {
            int* e = bucket + EXC_HEADER + EXC_ENTRY * (base + slot[it]);
            if ((lane & 3) == 0) { e[0] = (int)row; e[1] = MAPPED ? kbd[it] : it * 64 + wave * 16 + (lane >> 2); e[2] = code[it]; e[3] = 0; }
            e[4 + (lane & 3)] = (int)pk[it];
            pk[it] = 0u;
        } else if (has[it]) {
            pk[it] = shl_packed(pk[it], code[it] - best);
        }
    }
    E = best;
    return true;
}

template <int MAXIT, bool FULL = false>
__device__ __forceinline__ bool align_row(unsigned (&pk)[MAXIT], const int (&amax)[MAXIT], const int (&code)[MAXIT], int nit,
                                          int nkb, long long row, int* __restrict__ list, RowAlignSmem& sm, int& E,
                                          int bcap = ROW_BCAP) {
    return align_row_impl<MAXIT, FULL, false>(pk, amax, code, nit, nkb, row, list, sm, E, bcap, code);     // (kbd unused)
}

}  // namespace mi355q
#endif
