// mi355q_align.h -- exponent alignment of one (row, 256-value group) by one wave; shared by the fused
// activation kernel (mi355q_quant.hip) and the packed-operand kernel (mi355q_gemm.hip).
//
// Lane l holds the 4 integer mantissas q[0..3] of block l >> 2 (16 blocks, 4 lanes each), that block's largest
// |mantissa| `amax` and its biased exponent `code`.  The group gets ONE effective exponent E:
//   * common case: E = smallest exponent of a non-zero block and every block can be shifted left onto it;
//   * otherwise E is the candidate exponent whose window [E, E + head-room] holds the most blocks; blocks
//     outside the window are EXCEPTIONS: their mantissas are zeroed in the operand and the block is appended
//     to the operand's exception list (row, block index, exponent, 16 mantissa bytes) -- the sparse correction
//     kernel adds them back exactly;
//   * if the list has no room the row-group is left as it was (rowflag 0, own exponents): the list count still
//     grows past the capacity, which makes the GEMM take its blockwise-fallback kernel.
// Exception list layout (int32): [0] count (reservations, may exceed the capacity), [1] spare, [8 + 8 i ...]
// entry i = {row (-1: void), block, exponent code, 0, 4 dwords of mantissas}.
#ifndef MI355Q_ALIGN_H
#define MI355Q_ALIGN_H
#include <hip/hip_runtime.h>

namespace mi355q {

constexpr int EXC_HEADER = 8, EXC_ENTRY = 8;   // int32 words

struct AlignResult {
    int eout;        // effective exponent code to store for this lane's block
    bool flagged;    // the row-group carries one exponent (rowflag 1)
};

__device__ __forceinline__ AlignResult align_group(int (&q)[4], int amax, int code, bool valid, long long row, int kb,
                                                  int* __restrict__ list, int list_cap) {
    const int lane = threadIdx.x & 63;
    const bool has = valid && amax > 0;
    constexpr int BIG = 1 << 20;
    int emin = has ? code : BIG;
    emin = min(emin, __builtin_amdgcn_mov_dpp(emin, 0x124, 0xF, 0xF, true));   // row_ror:4
    emin = min(emin, __builtin_amdgcn_mov_dpp(emin, 0x128, 0xF, 0xF, true));   // row_ror:8
    emin = min(emin, __shfl_xor(emin, 16));
    emin = min(emin, __shfl_xor(emin, 32));
    const int s0 = has ? code - emin : 0;
    const bool ok = !has || (s0 <= 7 && (amax << s0) <= 127);
    AlignResult r;
    if (__all(ok)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] <<= s0;
        r.eout = emin == BIG ? code : emin;
        r.flagged = true;
        return r;
    }
    // ---- rare path (uniform over the wave): pick the exponent window that keeps the most blocks
    const int head = has ? __clz(amax) - 25 : 0;            // largest left shift that keeps |mantissa| <= 127
    int best_e = 0, best_n = -1;
    for (int c = 0; c < 16; ++c) {
        const int ec = __builtin_amdgcn_readlane(code, 4 * c);
        const int hc = __builtin_amdgcn_readlane((int)has, 4 * c);
        if (!hc) continue;
        const bool in = has && code >= ec && code - ec <= head;
        const int n = __builtin_popcountll(__ballot(in));
        if (n > best_n) { best_n = n; best_e = ec; }
    }
    const bool inw = has && code >= best_e && code - best_e <= head;
    const bool exc = has && !inw;
    const unsigned long long em = __ballot(exc);
    const int k = __builtin_popcountll(em) >> 2;                               // exception blocks in this group
    int base = 0;
    if (lane == 0 && list) base = atomicAdd(&list[0], k);
    base = __builtin_amdgcn_readfirstlane(base);
    const bool stored = list != nullptr && base + k <= list_cap;
    const int rank = __builtin_popcountll(em & ((1ull << (lane & ~3)) - 1ull)) >> 2;
    if (!stored) {
        // void the part of the reservation that lies inside the list, leave the row-group unaligned
        if (list && exc && (lane & 3) == 0 && base + rank < list_cap) list[EXC_HEADER + EXC_ENTRY * (base + rank)] = -1;
        r.eout = code;
        r.flagged = false;
        return r;
    }
    if (exc) {
        int* e = list + EXC_HEADER + EXC_ENTRY * (base + rank);
        if ((lane & 3) == 0) { e[0] = (int)row; e[1] = kb; e[2] = code; e[3] = 0; }
        e[4 + (lane & 3)] = (int)((unsigned)(q[0] & 0xFF) | ((unsigned)(q[1] & 0xFF) << 8) |
                                  ((unsigned)(q[2] & 0xFF) << 16) | ((unsigned)(q[3] & 0xFF) << 24));
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = 0;
    } else if (inw) {
        const int s = code - best_e;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] <<= s;
    }
    r.eout = best_e;
    r.flagged = true;
    return r;
}

}  // namespace mi355q
#endif
