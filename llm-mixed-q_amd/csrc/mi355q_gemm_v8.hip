// mi355q_gemm_v8.hip -- block-floating-point GEMM over ROW-aligned operands (gfx950).
//
// When the align step can put every block of a row onto ONE exponent (whole-K window, exceptions kept aside --
// mi355q_align_row.h), the contraction is a plain int8 x int8 -> int32 GEMM with one scale per row of x and one
// per row of w:
//     y[m,n] = sx[m] * sw[n] * ( sum_k xm'[m,k] * wm'[n,k] )  (+ bias[n])  (+ exception blocks),  K <= 16384.
// No rescale in the K loop: the int32 accumulators are the only live tile, so the wave tile is 128 x 64.
//
// Workgroup = 256 x 256 outputs, 8 waves as 2 x 4 (two per SIMD), v_mfma_i32_16x16x64_i8.  K-step 64: one stage =
// A 16 KiB + B 16 KiB of 1-KiB pieces (block-major inside, mi355q_gemm_v2.h), three stages filled by global_load_lds,
// counted s_waitcnt vmcnt, raw s_barriers.  Default schedule (SCHED 2): ONE barrier per K-step, both waves of a SIMD run
// the same stream -- 32 MFMAs a step from registers while fragment i + 2 is read (inline-asm ds_read_b128, hand-counted
// lgkmcnt) and the step's four LDS-DMA pieces for step t + 2 go out one per MFMA group.  (SCHED 0: the two waves of a
// SIMD one barrier apart, two phases per K-step; SCHED 1: the one-phase schedule of the 128 x 256 tile.)  The scale / bias
// slices of the tile, its two exception buckets and the lists' overflow words ride in front of the operand stream.
//
// Exceptions (blocks outside their row's exponent window; a few dozen per tile at most in the usual case) are added back
// without floating-point atomics and without a launch of their own: the tile reads its two buckets, gathers -- ONE round
// trip for all entries -- the other operand's blocks at each entry's K position (256 rows x 16 bytes = 4 KiB contiguous
// in the block-major pieces), multiplies, and keeps one vector of 256 products per entry in spare LDS; links the entries
// of each tile row / column into chains while the gathers fly; after the K loop adds the exception x exception terms (same
// K position in both lists), folds chains into their heads, and the store epilogue adds one vector per affected row /
// column.  Tiles with more entries than the spare area holds form the vectors in the stage area after the K loop; beyond
// that the products are added with atomics after the stores.  If a bucket overflowed anywhere (uniform over the grid) the
// launch's workgroups share the blockwise-exact product instead (v8_fallback).  Under-filled grids split K over several
// workgroups per tile (slabs + tickets, choose_splits).
// Roofline: int8 MFMA, 2*M*N*K ops; y leaves as full fp32 (64 MiB at 4096^2: ~10 us of HBM write time).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "mi355q_gemm_tile.h"

namespace mi355q {

// the two arithmetics of the tile kernel: int8 mantissas -> int32 (row-scale block-fp GEMM), or bf16 values -> fp32
// (operands that keep every block's own exponent: a block_fp value of width <= 9 is exact in bf16 and a product of two
// of them is exact in fp32).  Same fragment geometry: one 16-byte read per lane = 16 int8 or 8 bf16 of one row.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x4 v8_mma(const i32x4& fa, const i32x4& fb, const i32x4& c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 v8_mma(const i32x4& fa, const i32x4& fb, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb), c, 0, 0, 0);
}

template <int FIXMODE_, int TI, int SCHED = (TI == 4 ? 1 : 0), bool BF16 = false>     // 0: the product of the rewritten operands only (benchmarks), 1: with the exception add-back,
                            // 2: as 0, and workgroup 0 prints the clock it held over the K loop (diagnostic build)
__global__ __launch_bounds__(V8_NT, 1) void bfp_gemm_v8(const GemmArgs a_in, const float* __restrict__ sx,
                                                        const float* __restrict__ sw_in, const int* __restrict__ xlist,
                                                        const int* __restrict__ wlist_in, const uint8_t* __restrict__ xf,
                                                        const uint8_t* __restrict__ wf_in) {
    constexpr int FIXMODE = (FIXMODE_ == 1 || FIXMODE_ == 3) ? 1 : 0;      // 3: as 1, with phase timing printed by workgroup 0
    // TI = 16-row MFMA tiles per wave along M.  8: 256 x 256 workgroup tile, two MFMA phases per K-step, three 32-KiB
    // stages.  4: 128 x 256 tile (for shapes whose 256 x 256 tiles would leave compute units idle), one phase per K-step,
    // four 24-KiB stages (the steps are half as long, so the loads run three steps ahead).
    constexpr int TJ = 4, BM = 2 * TI * 16, WM = TI * 16, HALF = BM * 64, STAGE = HALF + 256 * 64;
    constexpr int NS = V8_S * V8_STAGE / STAGE, NPIECE = BM / 16 + 16, LPW = NPIECE / V8_NW;
    static_assert((TI == 8 && NS == 3) || (TI == 4 && NS == 4), "stage ring");
    __shared__ __attribute__((aligned(16))) unsigned char smem[V8_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, l16 = lane & 15, lq = lane >> 4;

    const unsigned long long kernel_t0 = FIXMODE_ == 3 ? __builtin_amdgcn_s_memrealtime() : 0ull;
    GemmArgs a = a_in;
    const float* __restrict__ sw = sw_in;
    const int* __restrict__ wlist = wlist_in;
    const uint8_t* __restrict__ wf = wf_in;
    // grouped launch: the column tiles of `ngroup` equally shaped weight operands side by side (one x, one grid)
    const int ngroup = a.ngroup > 1 ? a.ngroup : 1;
    const int tiles_m = (int)((a.M + BM - 1) / BM), tiles_n1 = (int)((a.N + V8_BN - 1) / V8_BN), tiles_n = tiles_n1 * ngroup;
    const int S = a.splits > 1 ? a.splits : 1;                 // workgroups per tile (split-K)
    const int nwg = tiles_m * tiles_n * S;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int split = pid % S, tile_id = pid / S;               // (a tile's slices are neighbours: same XCD, speed only)
    const int GM = 4, in_group = GM * tiles_n, group_id = tile_id / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (tile_id % in_group) % gsz;
    int tn = (tile_id % in_group) / gsz;
    if (ngroup > 1) {
        const int which = tn / tiles_n1;                        // (wave-uniform: scalar loads from the argument block)
        tn -= which * tiles_n1;
        // (selects over constant indices: a runtime index would put the argument block in scratch memory)
#define V8_PICK(f) (which == 0 ? a_in.f[0] : which == 1 ? a_in.f[1] : a_in.f[2])
        a.wm = V8_PICK(g_wm); a.we = V8_PICK(g_we); a.bias = V8_PICK(g_bias); a.y = V8_PICK(g_y);
        sw = V8_PICK(g_sw); wlist = V8_PICK(g_wlist); wf = V8_PICK(g_wf);
#undef V8_PICK
    }
    const long long m0 = (long long)tm * BM, n0 = (long long)tn * V8_BN;
    const int nsteps_all = (int)(a.K >> 6);
    const int kstep0 = (int)((long long)nsteps_all * split / S);          // this workgroup's slice of the K-steps
    const int nsteps = (int)((long long)nsteps_all * (split + 1) / S) - kstep0;
    const long long Mrows = a.M, Ncols = a.N;

    int* xb = reinterpret_cast<int*>(smem + V8_XB);
    int* wb = reinterpret_cast<int*>(smem + V8_WB);
    int* rowslot = reinterpret_cast<int*>(smem + V8_MAP);
    int* colslot = rowslot + 256;
    float* sxt = reinterpret_cast<float*>(smem + V8_SXT);
    float* swt = reinterpret_cast<float*>(smem + V8_SWT);
    float* bst = reinterpret_cast<float*>(smem + V8_BIAS);

    // ---- in front of the operand stream (same LDS-DMA queue, so landed by the first counted wait): the tile's
    //      scale slices and its two exception buckets; the bias slice goes through registers (no padding behind it)
    if (wave == 0 || wave == 1) {
        if (FIXMODE && !(wave == 0 && a.x_post)) {
            const int* b = wave == 0 ? row_bucket(xlist, m0) : row_bucket(wlist, n0);
            unsigned char* d = smem + (wave == 0 ? V8_XB : V8_WB);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q * 256 + lane * 4 < ROW_BUCKET_WORDS)
                    __builtin_amdgcn_global_load_lds((gptr_t)(b + q * 256 + lane * 4), (lptr_t)(d + q * 1024), 16, 0, 0);
        }
    } else if (!BF16 && wave == 2) {
        __builtin_amdgcn_global_load_lds((gptr_t)(sx + m0 + lane * 4), (lptr_t)(smem + V8_SXT), 16, 0, 0);
    } else if (!BF16 && wave == 3) {
        __builtin_amdgcn_global_load_lds((gptr_t)(sw + n0 + lane * 4), (lptr_t)(smem + V8_SWT), 16, 0, 0);
    } else if (wave == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long n = n0 + q * 64 + lane;
            bst[q * 64 + lane] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
        }
    } else if (FIXMODE && (wave == 5 || wave == 6)) {
        // the lists' overflow words arrive the same way (a scalar load here would put a memory round trip in
        // front of everything the workgroup does)
        __builtin_amdgcn_global_load_lds((gptr_t)((wave == 5 ? xlist : wlist) + lane), (lptr_t)(smem + V8_OVF + (wave - 5) * 256), 4, 0, 0);
    }

    const long long kp = a.K >> 6;
    const long long pa_max = ((a.M + 127) / 128) * 8 - 1, pb_max = ((a.N + 127) / 128) * 8 - 1;
    // piece p of a stage: p < BM / 16 -> 16 rows of A, else 16 rows of B; this wave stages pieces wave + 8 q
    // (bf16 flavour, a.x_segs > 1: x lies as column segments -- [segment][row piece][K-steps of the segment] -- where an
    //  all-gather of per-rank quantised slices left them; a K-step of a row piece is then found through its segment)
    const bool xseg = BF16 && a.x_segs > 1;
    const int sps = xseg ? (int)(kp / a.x_segs) : (int)kp;                  // K-steps per segment
    const int8_t* src[LPW];
    int dst[LPW];
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
        const int p = wave + V8_NW * q;
        src[q] = p < BM / 16 ? a.xm + min((m0 >> 4) + p, pa_max) * (long long)sps * 1024 + lane * 16 + (xseg ? 0ll : (long long)kstep0 * 1024)
                             : a.wm + min((n0 >> 4) + (p - BM / 16), pb_max) * kp * 1024 + lane * 16 + (long long)kstep0 * 1024;
        dst[q] = p * 1024;
    }
    auto piece_src = [&](int q, int step) -> const int8_t* {                // piece q of the slice's K-step `step`
        if (xseg && wave + V8_NW * q < BM / 16) {
            const int ks = kstep0 + step, sg = ks / sps;
            return src[q] + sg * a.x_seg_stride + (long long)(ks - sg * sps) * 1024;
        }
        return src[q] + (long long)step * 1024;
    };
    auto stage = [&](int step, int slot) {
#pragma unroll
        for (int q = 0; q < LPW; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)piece_src(q, step), (lptr_t)(smem + slot * STAGE + dst[q]), 16, 0, 0);
    };
    int aoff[TI], boff[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) aoff[i] = v8_off(wm * WM + i * 16 + l16, lq);
#pragma unroll
    for (int j = 0; j < TJ; ++j) boff[j] = HALF + v8_off(wn * 64 + j * 16 + l16, lq);

    // Pipelined schedule with the add-back: the second stage is requested behind the gathers of the correction vectors
    // (a wave's memory operations return in issue order: in front of them the gathers would queue behind 32 KiB more)
    constexpr bool LATE_STAGE1 = SCHED == 2 && FIXMODE;
    stage(0, 0);
    if (!LATE_STAGE1 && nsteps > 1) stage(1, 1);
    if (TI == 4 && SCHED == 1 && nsteps > 2) stage(2, 2);       // (the one-phase schedule runs three steps ahead)

    // ---- exception bookkeeping of this tile.  Its buckets rode in front of the operand stream; once they have landed
    //      (the first two stages stay in flight) the workgroup counts the entries, clears the row / column maps, links
    //      the entries of each tile row / column into a chain and requests every entry's correction vector -- 256
    //      floats the short launch in front of this kernel formed (mi355q_gemm_v6.hip) -- by ONE 1-KiB LDS-DMA each.
    unsigned long long pst[3] = {0, 0, 0};
    int cx = 0, cw = 0, mode = 0;         // mode 0: none, 1: vectors beside the stages, 2: in the stage area after
    float* corr = reinterpret_cast<float*>(smem + V8_CORR);          // the K loop, 3: added with atomics after the stores
    int* multi = reinterpret_cast<int*>(smem + V8_FLAGS);             // set when a row / column carries several entries
    // The correction vector of an entry: 256 floats, element p = the entry's block times the other operand's block at
    // the same K position in tile row / column p, scaled -- x entry (row r, block kb):
    //     v[p] = 2^(code - x_off) * sw[n0 + p] * dot16(entry, wm'[n0 + p, kb])          (w as stored: its own exceptions
    // are zero there and come in through the exception x exception terms); w entries the mirror image over the tile's
    // rows.  Formed HERE, by the tile that adds them: wave w takes entries w, w + 8, ...; a lane gathers the 16-byte
    // blocks of four tile rows / columns per entry (unconditional loads, clamped addresses), two entries in flight.
    // Two halves, so that other work can sit between request and use (the gathers cost a memory round trip): entries
    // base + wave + 8 u, u < U, of the combined list.
    auto gather_issue = [&](auto& qv, int base, auto ucount) {
        constexpr int U = decltype(ucount)::value;
        const int n = cx + cw;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + wave + V8_NW * u;
            if (i >= n) break;                                          // wave-uniform
            const bool is_x = i < cx;
            const int* e = v8_entry(xb, wb, cx, i);
            const long long kcol = (long long)e[1] * 16;
            const int8_t* qm = is_x ? +a.wm : +a.xm;           // (unary +: values, not a select of addresses in the argument block)
            const long long q0 = is_x ? n0 : m0, qmax = (is_x ? Ncols : Mrows) - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                qv[u][c] = *reinterpret_cast<const int4*>(qm + tiled_offset(min(q0 + c * 64 + lane, qmax), kcol, a.K));
        }
    };
    auto gather_finish = [&](auto& qv, float* area, int base, auto ucount) {
        constexpr int U = decltype(ucount)::value;
        const int n = cx + cw;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + wave + V8_NW * u;
            if (i < n) {
                const bool is_x = i < cx;
                const int* e = v8_entry(xb, wb, cx, i);
                const int4 pv = *reinterpret_cast<const int4*>(e + 4);
                const int sh = e[2] - (is_x ? +a.x_off : +a.w_off);
                const float* sc = is_x ? swt : sxt;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    area[i * 256 + c * 64 + lane] = __builtin_ldexpf((float)dot16(pv, qv[u][c]), sh) * sc[c * 64 + lane];
            }
        }
    };
    constexpr int U_PRO = (V8_FAST_MAX + V8_NW - 1) / V8_NW;        // all of a tile's entries in ONE round trip
    if (FIXMODE) {
        // (the buckets are older than the operand stages requested above)
        if (FIXMODE_ == 3) pst[0] = __builtin_amdgcn_s_memrealtime();
        // (what may still fly: the stages requested above, nothing older)
        if (TI == 4 && SCHED == 1 && nsteps > 2) V8_WAIT(3 * LPW); else if (nsteps > 1 && !LATE_STAGE1) V8_WAIT(2 * LPW); else V8_WAIT(LPW);
        __builtin_amdgcn_s_barrier();
        if (FIXMODE_ == 3) pst[1] = __builtin_amdgcn_s_memrealtime();
        // a bucket overflowed somewhere (uniform over the grid): the row-scale product does not apply; the workgroups of
        // this launch share the blockwise-exact product instead (v8_fallback).  The operand loads in flight land in LDS
        // only; nothing else is pending.
        const int* ovf = reinterpret_cast<const int*>(smem + V8_OVF);
        if (__builtin_amdgcn_readfirstlane(ovf[0] | ovf[64]) != 0) {
            V8_WAIT(0);
            __syncthreads();
            // (grouped launch: the workgroups of one weight operand share that operand's product)
            v8_fallback(a, xf, wf, xlist, wlist, smem, ngroup > 1 ? (tm * tiles_n1 + tn) * S + split : (int)blockIdx.x,
                        ngroup > 1 ? tiles_m * tiles_n1 * S : nwg);
            return;
        }
        cx = a.x_post ? 0 : __builtin_amdgcn_readfirstlane(min(xb[0], ROW_BCAP));
        cw = __builtin_amdgcn_readfirstlane(min(wb[0], ROW_BCAP));
        const int n = cx + cw;
        // (split-K: only the tile's last arriver adds the exceptions, but which slice that will be is not known here:
        // every slice forms the vectors in its prologue, where the gathers hide behind the first stages -- forming them
        // behind the K loop instead cost the reducer 5-8 us of exposed round trips)
        mode = n == 0 ? 0 : (n <= V8_FAST_MAX ? 1 : (n <= V8_SLOW_MAX ? 2 : 3));
        if (mode) {
            rowslot[tid & 255] = -1;                        // tid < 256: rowslot, else colslot (contiguous)
            if (tid >= 256) colslot[tid & 255] = -1;
            if (tid == 0) *multi = 0;
            int4 qv[U_PRO][4];
            if (mode == 1) gather_issue(qv, 0, std::integral_constant<int, U_PRO>{});
            if (LATE_STAGE1 && nsteps > 1) stage(1, 1);
            __builtin_amdgcn_s_barrier();                   // (maps cleared before the chain heads are written)
            // chains, while the gathers are in flight: the entries of one tile row / column, linked by DESCENDING block
            // index (head in rowslot / colslot, successor in word 3 of the entry's LDS copy, -2 marks a void entry).
            // The order is a property of the data, not of which workgroup reserved its list slots first: results are
            // reproducible.  16 lanes share an entry (each scans every 16th entry of the same operand).
            for (int i0 = 0; i0 < n; i0 += V8_NT / 16) {               // uniform
                const int i = i0 + (tid >> 4), sub = tid & 15;
                const bool valid = i < n;
                const bool is_x = i < cx;
                int* e = v8_entry(xb, wb, cx, valid ? i : 0);
                const int r = e[0], kb = e[1];
                const bool live = valid && (is_x ? (r >= m0 && r < m0 + BM && r < Mrows) : (r >= n0 && r < n0 + 256 && r < Ncols));
                const int lo = is_x ? 0 : cx, hi = is_x ? cx : n;
                int key = -1, later = 0;                                // key = (block << 8 | index) of the best predecessor
                if (live)
                    for (int j = lo + sub; j < hi; j += 16) {
                        const int* f = v8_entry(xb, wb, cx, j);
                        if (f[0] != r) continue;
                        const int kj = f[1];
                        if (kj < kb) key = max(key, (kj << 8) | (j - lo));
                        later |= kj > kb ? 1 : 0;
                    }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    key = max(key, __shfl_xor(key, o));
                    later |= __shfl_xor(later, o);
                }
                if (valid && sub == 0) {
                    if (!live) {
                        e[3] = -2;
                    } else {
                        e[3] = key >= 0 ? (key & 255) : -1;
                        if (!later) (is_x ? rowslot : colslot)[r - (int)(is_x ? m0 : n0)] = i - lo;
                        if (key >= 0) *multi = 1;
                    }
                }
            }
            if (mode == 1) gather_finish(qv, reinterpret_cast<float*>(smem + V8_CORR), 0, std::integral_constant<int, U_PRO>{});
            if (FIXMODE_ == 3) pst[2] = __builtin_amdgcn_s_memrealtime();
        } else if (LATE_STAGE1 && nsteps > 1) {
            stage(1, 1);
        }
    }

    static_assert(!BF16 || FIXMODE_ == 0, "the bf16 arithmetic has no exception lists");
    using acc_t = typename std::conditional<BF16, f32x4, i32x4>::type;
    acc_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    unsigned long long c0 = 0, r0 = 0;
    unsigned long long rt[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long rt_start = kernel_t0;
    if (FIXMODE_ == 2) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (FIXMODE_ == 3) { rt[0] = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime(); }

    // Two wave groups (wm = 0 / 1: the two waves of every SIMD) run ONE BARRIER apart: while a group issues its 16
    // MFMAs between two barriers, the other reads its next fragments from LDS and issues its LDS-DMA loads.
    // Rules both schedules below keep: a stage is re-filled only after a barrier that every wave reaches with its reads
    // of that stage retired; and every wave has waited for its own pieces of a step before a barrier that every
    // reader of that step passes first (LDS-DMA data are ordered by the issuing wave's vmcnt + a barrier only).
    auto dma_pieces = [&](int step, int sl, int q0, int q1) {
#pragma unroll
        for (int q = 0; q < LPW; ++q)
            if (q >= q0 && q < q1)
                __builtin_amdgcn_global_load_lds((gptr_t)piece_src(q, step), (lptr_t)(smem + sl * STAGE + dst[q]), 16, 0, 0);
    };
    int slot = 0;
    constexpr bool ONEPHASE = SCHED == 1;
    if constexpr (SCHED == 2) {        // (discarded, not merely dead, for the 128-row tile: its groups index acc[0..7])
        // PIPELINED schedule: ONE barrier per K-step, no staggered wave groups.  Every wave keeps its MFMA stream fed from
        // registers: while the 4 MFMAs of A fragment i issue, fragment i + 2 is being read (the last two reads of a step
        // and the four B reads fetch step t + 1, from the next stage), and the wave's LDS-DMA pieces of step t + 2 go out
        // one per MFMA group.  Both waves of a SIMD run the same stream; the matrix pipe alternates between them and
        // either one's LDS / DMA issue slots hide behind the other's MFMAs.
        //   barrier(t): every wave has waited for its own pieces of step t + 1 (requested a whole step earlier) and has
        //   finished every read of step t - 1 (all consumed by MFMAs it has issued)  =>  behind it stage t + 1 may be read
        //   and stage t - 1 = stage t + 2 of the ring of three may be re-filled.
        // Round 4: the 128 x 256 tile takes the same schedule (TI = 4: four MFMA groups a step, three DMA pieces a wave, a ring
        // of three of its four 24-KiB stages): its one-phase schedule ran 1400 clocks a K-step against 515 of MFMA work
        // (profiles/r04_shard_shapes.txt).  Reads of a step there: group i issues A(i + 2) and B(i) of the next step.
        static_assert(SCHED != 2 || (NS == 3 && TI == 8 && LPW == 4) || (NS == 4 && TI == 4 && LPW == 3), "pipelined schedule: ring of three");
        i32x4 fa[4], fb0[TJ], fb1[TJ];
        // Fragment reads are inline assembly with hand-counted waits (LDS reads of a wave return in issue order): left to
        // the compiler every read sinks to just in front of its first use behind an lgkmcnt(0).  Lane-constant part of
        // the addresses; fragment i is i KiB further (immediate offset), a stage STAGE bytes further.
        const int va = piece_lds_off(wm * WM + l16, lq), vb = HALF + piece_lds_off(wn * 64 + l16, lq);
#define V8_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define V8_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n))
        if (nsteps > 1) V8_WAIT(LPW); else V8_WAIT(0);
        __builtin_amdgcn_s_barrier();
        V8_DSR(fb0[0], vb, 0); V8_DSR(fb0[1], vb, 1024); V8_DSR(fb0[2], vb, 2048); V8_DSR(fb0[3], vb, 3072);
        V8_DSR(fa[0], va, 0); V8_DSR(fa[1], va, 1024);
        __builtin_amdgcn_sched_barrier(0);
        // reads issued in MFMA group i of a step:  i < 6: A(i+2);  i = 6, 7: A(0), A(1) of the next step;  i = 2..5 also
        // B(i-2) of the next step.  Reads still in flight when group i's MFMAs need A(i) -- the counted waits below.
        // ONE body for every step (two instances: the B fragment sets swap roles): the last step reads its "successor"
        // fragments from the stage it is on (unused values: the hand-counted waits stay the same) and the last two steps
        // request the final step once more (into the free stage of the ring; drained behind the loop, never read).
        const int dbg = FIXMODE_ == 3 ? a.dbg : 0;            // (diagnostic build: knock parts of the loop out, results invalid)
        constexpr int B0 = TI == 8 ? 2 : 0;                   // first MFMA group that reads a B fragment of the next step
        auto body = [&](i32x4 (&fb)[TJ], i32x4 (&fbn)[TJ], int t, int sc, int sn, int dslot) {
            V8_WAIT(0);
            if (!(dbg & 2)) __builtin_amdgcn_s_barrier();
            const int ac = va + sc, an = va + sn, bn = vb + sn;
            const int dstep = min(t + 2, nsteps - 1);
#define V8_GROUP(i, wait)                                                                                              \
            if (!(dbg & 4)) {                                                                                          \
            if (i < TI - 2) V8_DSR(fa[(i + 2) & 3], ac, (i + 2) * 1024);                                               \
            else V8_DSR(fa[(i + 2) & 3], an, (i + 2 - TI) * 1024);                                                     \
            if (i >= B0 && i < B0 + TJ) V8_DSR(fbn[(i - B0) & 3], bn, ((i - B0) & 3) * 1024);                          \
            }                                                                                                          \
            if (i < LPW && !(dbg & 1))                                                                                 \
                __builtin_amdgcn_global_load_lds((gptr_t)piece_src(i & 3, dstep), (lptr_t)(smem + dslot * STAGE + dst[i & 3]), \
                                                 16, 0, 0);                                                            \
            V8_LGKM(wait);                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                                             \
                acc[i][j] = v8_mma(fa[i & 3], fb[j], acc[i][j]);                                                                        \
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TI == 8) {
                V8_GROUP(0, 2) V8_GROUP(1, 2) V8_GROUP(2, 3) V8_GROUP(3, 4) V8_GROUP(4, 5) V8_GROUP(5, 5) V8_GROUP(6, 4) V8_GROUP(7, 3)
            } else {
                // reads in issue order: .. A0' B2' | A1' B3' || A2 B0" | A3 B1" | A0" B2" | A1" B3" ..  (' this step, " the next).
                // Group 0 needs the whole of B' (B3' is the youngest: two reads behind it), group 1 A1' (older still: the four
                // of groups 0 and 1 may fly), groups 2 / 3 their A issued two groups earlier (five behind it)
                V8_GROUP(0, 2) V8_GROUP(1, 4) V8_GROUP(2, 5) V8_GROUP(3, 5)
            }
#undef V8_GROUP
        };
        int s0 = 0, s1 = 1, s2 = 2;
        auto rot = [&]() { const int o = s0; s0 = s1; s1 = s2; s2 = o; };
        for (int t = 0; t < nsteps; t += 2) {                // (nsteps is even: K % 128 == 0)
            body(fb0, fb1, t, s0 * STAGE, s1 * STAGE, s2);
            rot();
            body(fb1, fb0, t + 1, s0 * STAGE, (t + 2 < nsteps ? s1 : s0) * STAGE, s2);
            rot();
        }
        V8_WAIT(0);
        V8_LGKM(0);                                          // (the compiler does not know these reads are in flight)
        __builtin_amdgcn_sched_barrier(0);
#undef V8_DSR
#undef V8_LGKM
        __builtin_amdgcn_s_barrier();                   // (the epilogue's LDS areas: every wave is out of the stages)
    } else if (!ONEPHASE) {
        // A K-step is two phases (A rows 0-63, then 64-127 of the wave tile).  The pieces of step t+2 are requested in
        // phase 1 of step t (two) and phase 0 of step t+1 (two): a stage is re-filled two barriers after its last read,
        // so fragment reads may retire behind the barrier, beside the other group's wait.
        int nslot = 2 % NS, pslot = 0;
        if (nsteps > 1) V8_WAIT(LPW); else V8_WAIT(0);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nsteps; ++t) {
            const unsigned char* sbase = smem + slot * STAGE;
            i32x4 fa[4], fb[TJ];
            // ---- phase 0
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[j] = *reinterpret_cast<const i32x4*>(sbase + boff[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const i32x4*>(sbase + aoff[i]);
            if (t >= 1 && t + 1 < nsteps) dma_pieces(t + 1, pslot, 2, 4);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = v8_mma(fa[i], fb[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const i32x4*>(sbase + aoff[TI - 4 + i]);
            if (t + 2 < nsteps) {
                dma_pieces(t + 2, nslot, 0, 2);
                pslot = nslot;
                nslot = nslot + 1 == NS ? 0 : nslot + 1;
            }
            if (t + 2 < nsteps) V8_WAIT(2); else V8_WAIT(0);    // step t+1 complete (two newest pieces: step t+2)
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[TI - 4 + i][j] = v8_mma(fa[i], fb[j], acc[TI - 4 + i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
    } else {
        // One phase per K-step; barriers G(2t) (before the leading group's MFMAs of step t) and G(2t+1) (behind them;
        // the lagging group is one barrier later).  Step t+D (D = stages - 1) is requested once every read of step t-1
        // has retired: behind G(2t), i.e. after its own MFMAs for the leading group, before them for the lagging one.
        // Either way a wave then waits for its pieces of step t+1 (the later steps stay in flight) before it arrives at
        // G(2t+1), the barrier the leading group passes before it reads step t+1.
        constexpr int D = NS - 1;
        int nslot = D % NS;
        if (D == 3 && nsteps > 2) V8_WAIT(2 * LPW); else if (nsteps > 1) V8_WAIT(LPW); else V8_WAIT(0);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        auto request_and_wait = [&](int t) {
            if (t + D < nsteps) {
                dma_pieces(t + D, nslot, 0, LPW);
                nslot = nslot + 1 == NS ? 0 : nslot + 1;
            }
            if (D == 3) {
                if (t + 3 < nsteps) V8_WAIT(2 * LPW); else if (t + 2 < nsteps) V8_WAIT(LPW); else V8_WAIT(0);
            } else {
                if (t + 2 < nsteps) V8_WAIT(LPW); else V8_WAIT(0);
            }
        };
        for (int t = 0; t < nsteps; ++t) {
            const unsigned char* sbase = smem + slot * STAGE;
            i32x4 fa[TI], fb[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[j] = *reinterpret_cast<const i32x4*>(sbase + boff[j]);
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[i] = *reinterpret_cast<const i32x4*>(sbase + aoff[i]);
            if (wm == 1) request_and_wait(t);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = v8_mma(fa[i], fb[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            if (wm == 0) request_and_wait(t);
            __builtin_amdgcn_s_barrier();
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
    }

    if (FIXMODE_ == 3) { rt[1] = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime() - c0; }
    if (FIXMODE_ == 2) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if ((blockIdx.x == 0 || blockIdx.x == 77) && tid == 0)
            printf("wg %d: K loop %llu shader clocks in %llu x 10 ns -> %.0f MHz, %.1f clocks per K-step\n", blockIdx.x, c1 - c0,
                   r1 - r0, (double)(c1 - c0) / (double)(r1 - r0) * 100.0, (double)(c1 - c0) / nsteps);
    }
    if (S > 1) {
        // ---- split-K: every slice leaves its raw accumulators in its slab (16 bytes a lane, 1 KiB a wave instruction);
        //      the slice that arrives last at the tile's ticket sums all slabs IN SLICE ORDER (reproducible for the fp32
        //      flavour too; the int32 sums are exact in any order) and goes on to the epilogue, the others leave.
        //      Hand-off: plain stores, every wave's vmcnt(0), workgroup barrier, agent-scope release by one lane, relaxed
        //      ticket; the reducer acquires once, then loads plainly (cdna guide, Guideline 16).
        constexpr long long SLAB = (long long)BM * 256 * 4;
        acc_t* slab = reinterpret_cast<acc_t*>(static_cast<unsigned char*>(a.slabs) + ((long long)tile_id * S + split) * SLAB);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) slab[((wave * TI + i) * TJ + j) * 64 + lane] = acc[i][j];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flagw = reinterpret_cast<int*>(smem + V8_FLAGS) + 8;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int t = __hip_atomic_fetch_add(&a.tickets[tile_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == S - 1 ? 1 : 0;
            if (last) {
                __hip_atomic_store(&a.tickets[tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // idle again
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flagw = last;
        }
        __syncthreads();
        if (*flagw == 0) return;
        const acc_t* tslabs = reinterpret_cast<const acc_t*>(static_cast<unsigned char*>(a.slabs) + (long long)tile_id * S * SLAB);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
        for (int sl = 0; sl < S; ++sl) {
            const acc_t* sp = tslabs + (long long)sl * (SLAB / 16);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] += sp[((wave * TI + i) * TJ + j) * 64 + lane];
        }
    }
    if (FIXMODE && mode) {
        __syncthreads();                                         // every wave is done with the stage area
        if (mode == 2) {
            corr = reinterpret_cast<float*>(smem);
            for (int base = 0; base < cx + cw; base += 2 * V8_NW) {        // (two entries a wave and round: the
                int4 qv2[2][4];                                             //  accumulators are live here)
                gather_issue(qv2, base, std::integral_constant<int, 2>{});
                gather_finish(qv2, corr, base, std::integral_constant<int, 2>{});
            }
            __syncthreads();
        }
        if (mode != 3) {
            for (int idx = tid; idx < cx * cw; idx += V8_NT) {   // exception x exception: one (vector, element) each
                const int ex = idx / cw;
                const int* e = xb + EXC_HEADER + EXC_ENTRY * ex;
                const int* f = wb + EXC_HEADER + EXC_ENTRY * (idx % cw);
                if (e[3] == -2 || f[3] == -2 || e[1] != f[1]) continue;
                const int d = dot16(*reinterpret_cast<const int4*>(e + 4), *reinterpret_cast<const int4*>(f + 4));
                corr[ex * 256 + (int)(f[0] - n0)] += __builtin_ldexpf((float)d, e[2] + f[2] - a.scale_bias);
            }
            __syncthreads();
            // rows / columns with two or more entries (flagged while the chains were pushed; uncommon): fold the
            // vectors of a chain into its head's, so that the epilogue reads ONE vector per row / column
            if (*multi != 0 && wave == 0) {
                for (int base = 0; base < cx + cw; base += 64) {
                    const int i = base + lane;
                    bool hm = false;
                    int succ = -1;
                    if (i < cx + cw) {
                        const int* e = v8_entry(xb, wb, cx, i);
                        const bool is_x = i < cx;
                        if (e[3] != -2) {
                            const int head = (is_x ? rowslot : colslot)[e[0] - (is_x ? m0 : n0)];
                            hm = head == (is_x ? i : i - cx) && e[3] >= 0;
                            succ = e[3];
                        }
                    }
                    unsigned long long todo = __ballot(hm);
                    while (todo) {
                        const int src = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        const int u = base + src, off = u < cx ? 0 : cx;
                        int sidx = __builtin_amdgcn_readlane(succ, src);
                        while (sidx >= 0) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) corr[u * 256 + c * 64 + lane] += corr[(off + sidx) * 256 + c * 64 + lane];
                            sidx = __builtin_amdgcn_readfirstlane(v8_entry(xb, wb, cx, off + sidx)[3]);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: y = float(acc) * sx[m] * sw[n] + bias[n] (+ correction vectors).  Nothing is loaded from
    //      global memory between the stores (vmcnt counts loads and stores alike: a load's wait would drain them).
    if (FIXMODE_ == 3) rt[2] = __builtin_amdgcn_s_memrealtime();
    const bool look = FIXMODE && (mode == 1 || mode == 2);
    // One 16-row fragment at a time -- scale, add the vectors of its rows / columns, store -- so that the first stores
    // leave a few hundred cycles after the K loop: the 64 MiB of y are bound by the HBM write rate chip-wide, and every
    // microsecond the stores start earlier comes off the kernel.
    float swv[TJ], bv[TJ];
    int sc_[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int cl = wn * 64 + j * 16 + l16;
        swv[j] = BF16 ? 1.f : swt[cl];
        bv[j] = bst[cl];
        sc_[j] = look ? colslot[cl] : -1;
    }
    if (FIXMODE_ == 3) rt[3] = rt[4] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        float val[TJ][4];
        const f32x4 sxv = BF16 ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(&sxt[wm * WM + i * 16 + lq * 4]);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) val[j][r] = BF16 ? (float)acc[i][j][r] + bv[j] : (float)acc[i][j][r] * sxv[r] * swv[j] + bv[j];
        if (look) {
            // one vector per row / column (chains were folded above): one predicated read each
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                if (__any(sc_[j] >= 0)) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(corr + (cx + max(sc_[j], 0)) * 256 + wm * WM + i * 16 + lq * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[j][r] += sc_[j] >= 0 ? c4[r] : 0.f;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sr = rowslot[wm * WM + i * 16 + lq * 4 + r];
                if (__any(sr >= 0)) {
                    const float* v = corr + max(sr, 0) * 256 + wn * 64 + l16;
#pragma unroll
                    for (int j = 0; j < TJ; ++j) val[j][r] += sr >= 0 ? v[j * 16] : 0.f;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long row = m0 + wm * WM + i * 16 + lq * 4 + r;
            float* yrow = a.y + row * a.ldy + n0 + wn * 64 + l16;
            const float* rrow = a.resid ? a.resid + row * a.ldr + n0 + wn * 64 + l16 : nullptr;      // (the caller's residual add, in the store)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                if (n0 + wn * 64 + j * 16 + l16 < a.N && row < a.M) yrow[j * 16] = rrow ? val[j][r] + rrow[j * 16] : val[j][r];
        }
    }
    if (FIXMODE_ == 3) {
        rt[5] = __builtin_amdgcn_s_memrealtime();
        if ((blockIdx.x == 0 || blockIdx.x == 77) && (tid == 0 || tid == 448))
            printf("wg %d wave %d: [issue %llu, buckets %llu, vectors %llu, chains+stage0 %llu] prologue %llu | loop %llu | cross %llu | scale %llu | lookups %llu | stores %llu (x 10 ns) cx %d cw %d mode %d; loop %.0f MHz, %.0f clocks per K-step\n", blockIdx.x, wave,
                   pst[0] - rt_start, pst[1] - pst[0], pst[2] - pst[1], rt[0] - pst[2], rt[0] - rt_start, rt[1] - rt[0], rt[2] - rt[1], rt[3] - rt[2], rt[4] - rt[3], rt[5] - rt[4], cx, cw, mode,
                   (double)c0 / (double)(rt[1] - rt[0]) * 100.0, (double)c0 / nsteps);
    }
    if (FIXMODE && mode == 3) {
        V8_WAIT(0);
        __syncthreads();
        v8_fix_atomic(a, xb, wb, cx, cw, sxt, swt, m0, n0, BM);
    }
}


// ---- split-K workspace: raw accumulator slabs + one ticket per tile, owned by the library, one per (device, stream),
//      grow-only; tickets are zero whenever no launch is in flight (the reducer of a tile clears its ticket).
//      Under stream capture (a HIP graph being recorded) nothing is allocated or freed: a shape that needs growth then
//      gets no workspace (null: the caller launches unsplit), and a workspace that a capture has been handed is never
//      freed afterwards -- an instantiated graph keeps its pointers -- growth retires the old buffers instead.
SplitWorkspace* split_workspace(hipStream_t st, size_t slab_bytes, int ntickets) {
    struct Owned { SplitWorkspace w; bool slabs_in_graph = false, tickets_in_graph = false; };      // (per buffer: ADVICE r3)
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Owned> all;
    static std::vector<void*> retired;                     // (buffers a recorded graph may still use: kept for good)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(mu);
    Owned& o = all[{dev, st}];
    SplitWorkspace& w = o.w;
    const bool grow = w.slab_bytes < slab_bytes || w.ntickets < ntickets;
    if (grow && capturing) return nullptr;
    if (w.slab_bytes < slab_bytes) {
        if (w.slabs) {
            if (o.slabs_in_graph) retired.push_back(w.slabs);
            else (void)hipFree(w.slabs);                  // (synchronises: nothing of this workspace is in flight after)
        }
        w.slabs = nullptr;
        w.slab_bytes = 0;
        o.slabs_in_graph = false;                         // (a fresh buffer: no graph knows it yet)
        if (hipMalloc(&w.slabs, slab_bytes) != hipSuccess) return nullptr;
        w.slab_bytes = slab_bytes;
    }
    if (w.ntickets < ntickets) {
        if (w.tickets) {
            if (o.tickets_in_graph) retired.push_back(w.tickets);
            else (void)hipFree(w.tickets);
        }
        w.tickets = nullptr;
        w.ntickets = 0;
        o.tickets_in_graph = false;
        const int n = (ntickets + 1023) / 1024 * 1024;
        if (hipMalloc(reinterpret_cast<void**>(&w.tickets), (size_t)n * 4) != hipSuccess) return nullptr;
        if (hipMemsetAsync(w.tickets, 0, (size_t)n * 4, st) != hipSuccess) return nullptr;
        w.ntickets = n;
    }
    if (capturing) o.slabs_in_graph = o.tickets_in_graph = true;
    return &w;
}
// slices per tile for an under-filled grid: the largest S with tiles * S <= 256 (one workgroup per compute unit), whole
// and, where the schedule needs it, even numbers of K-steps per slice, at least 8 of them
// `min_steps`: K-steps a slice must keep.  Splitting costs the slabs' round trip through memory (S x the output, written
// and read), the agent-scope release / acquire and, in the flavour that carries exception lists, every slice's own
// bookkeeping prologue: measured 16-22 us at 128 tiles x 2 slices, so a row-scale int8 product is split only while a
// slice keeps 32 steps (2048^3: 27.7 us unsplit, 38.3 split in two; 4096 x 4096 x 512: 46.7 unsplit, 40.2 in two)
int choose_splits(long long tiles, int nsteps_all, bool need_even, int min_steps) {
    static const int forced = getenv("MI355Q_V8_SPLITS") ? atoi(getenv("MI355Q_V8_SPLITS")) : 0;
    if (forced) min_steps = 8;
    int best = 1;
    for (int S = 2; S <= 16; ++S) {
        if (nsteps_all % S) continue;
        const int steps = nsteps_all / S;
        if (steps < min_steps) break;
        if (need_even && (steps & 1)) continue;
        if (forced ? S > forced : tiles * S > 256) break;
        best = S;
    }
    return best;
}

// the small tiles of mi355q_gemm_v10.hip under a forced geometry (MI355Q_V10 = 1 | 2 | 3 | 4; sweeps and tests) -- split-K as the
// environment pins it (MI355Q_V8_SPLITS) or none
static int v10_forced_launch(const GemmArgs& a_in, const float* sx, const float* sw, const int* xlist, const int* wlist, hipStream_t st,
                             const uint8_t* xf, const uint8_t* wf, bool bf16, int geom) {
    GemmArgs a = a_in;
    int bm, bn;
    v10_tile_shape(geom, bm, bn);
    const long long tiles = ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * (a.ngroup > 1 ? a.ngroup : 1);
    const int forced = getenv("MI355Q_V8_SPLITS") ? atoi(getenv("MI355Q_V8_SPLITS")) : 0;
    const int nsteps_all = (int)(a.K >> 6);
    int S = forced > 1 ? forced : 1;
    while (S > 1 && (nsteps_all % S || nsteps_all / S < 2)) --S;
    a.splits = 1;
    if (geom >= 5) {                                // (the K-group geometries: never split across workgroups; whole pairs of K-steps)
        S = 1;
        if (a.K % 128) geom = geom == 5 ? 3 : 1;
    }
    if (S > 1) {
        SplitWorkspace* w = split_workspace(st, (size_t)tiles * S * bm * bn * 4, 2 * (int)tiles);      // (two ticket words a tile)
        if (w) {
            a.splits = S;
            a.slabs = w->slabs;
            a.tickets = w->tickets;
        }
    }
    return launch_bfp_gemm_v10(a, sx, sw, xlist, wlist, st, xf, wf, bf16, geom);
}
static int v10_forced() {                       // (read per launch: the tests pin one geometry after the other in one process)
    const char* e = getenv("MI355Q_V10");
    const int g = e ? atoi(e) : 0;
    return g >= 1 && g <= 6 ? g : 0;
}

int launch_bfp_gemm_v8(const GemmArgs& a_in, const float* sx, const float* sw, const int* xlist, const int* wlist,
                       int list_cap, hipStream_t st, const uint8_t* xf, const uint8_t* wf) {
    (void)list_cap;
    if (v10_forced() && a_in.K % 64 == 0 && (!(xlist && wlist) || (xf && wf))) return v10_forced_launch(a_in, sx, sw, xlist, wlist, st, xf, wf, false, v10_forced());
    // Round 5: grids of at most 128 tiles of 256 x 256 -- half the compute units or fewer -- take 128 x 128 tiles, two
    // four-wave workgroups a compute unit, unsplit (mi355q_gemm_v10.hip; profiles/r05_small_tiles.txt: 4096 x 512 x 4096 36.7 ->
    // 27.7 us, Llama-7B v_proj 47.1 -> 41.7, 2048^3 23.6 -> 19.0).  MI355Q_V10_AUTO=0 keeps the round-4 choice for A/B runs; a
    // pinned tile height (MI355Q_V8_TILE_ROWS, tests of the other kernels) does too.
    {
        static const int v10_auto = getenv("MI355Q_V10_AUTO") ? atoi(getenv("MI355Q_V10_AUTO")) : 1;
        const long long t256 = ((a_in.M + 255) / 256) * ((a_in.N + 255) / 256) * (a_in.ngroup > 1 ? a_in.ngroup : 1);
        const char* pinned = getenv("MI355Q_V8_TILE_ROWS");
        if (v10_auto && t256 <= 128 && a_in.K % 64 == 0 && !(pinned && atoi(pinned)) && !getenv("MI355Q_V8_SPLITS") && (!(xlist && wlist) || (xf && wf)) &&
            !getenv("MI355Q_V8_CLOCK") && !getenv("MI355Q_V8_STAMPS")) {
            GemmArgs a = a_in;
            a.splits = 1;
            // (128 x 64 tiles where 128 x 128 ones would fill half the compute units or fewer -- end of round 5:
            //  4096 x 512 x 4096 26.4 -> 22.4 us at W6A6, 20.1 -> 16.7 at W4A4; level from ~176 tiles of 128 x 128 on)
            const long long g3 = ((a.M + 127) / 128) * ((a.N + 127) / 128) * (a.ngroup > 1 ? a.ngroup : 1);
            // (round 6: 129 .. 256 tiles of 128 x 128 -- one four-wave workgroup a compute unit, every wave alone on its SIMD -- as
            //  8-wave workgroups of two K-groups: 4096 x 1024 x 4096 30.2 -> 27.4 us, Llama up / P = 8 27.5 -> 25.1, 2048^3 20.2 -> 19.5;
            //  profiles/r06_shard_shapes.txt.  MI355Q_V10_KG=0: the round-5 choice, A/B runs)
            static const int kg_auto = getenv("MI355Q_V10_KG") ? atoi(getenv("MI355Q_V10_KG")) : 1;
            const int geom = g3 <= 128 ? 4 : (kg_auto && g3 <= 256 && a.K % 128 == 0 && a.ngroup <= 1 ? 5 : 3);
            return launch_bfp_gemm_v10(a, sx, sw, xlist, wlist, st, xf, wf, false, geom);
        }
    }
    GemmArgs a = a_in;
    const int ngroup = a.ngroup > 1 ? a.ngroup : 1;            // grouped launch: that many weight operands' column tiles
    // 256 x 256 tiles unless they would leave too many of the 256 compute units idle: a 128 x 256 tile does half the
    // work in 0.8 of the time (measured: 48 vs 58 us at 2048 x 4096 x 4096; the fragment reads and LDS-DMA issue of a
    // K-step are shared by half as many MFMAs)
    const long long tn = (a.N + V8_BN - 1) / V8_BN * ngroup;
    const long long t256 = ((a.M + 255) / 256) * tn, t128 = ((a.M + 127) / 128) * tn;
    const double cost256 = (double)((t256 + 255) / 256) * 1.0, cost128 = (double)((t128 + 255) / 256) * 0.82;
    const char* force = getenv("MI355Q_V8_TILE_ROWS");          // (tests pin either flavour)
    const bool small = force && atoi(force) ? atoi(force) == 128 : cost128 < cost256;
    // K-loop schedule of the 128 x 256 tile: 2 = the pipelined one of the 256 x 256 tile (round 4, default), 1 = one phase per step
    constexpr int small_sched = 2;          // (the one-phase schedule only serves K % 128 == 64 now; its switch went in round 5)
    unsigned tiles = (unsigned)(small ? t128 : t256);
    {   // under-filled grid: split K (the 128-row tile's schedule takes any slice length, the 256-row one even ones)
        constexpr int sched_ = 2;
        const int S = choose_splits(tiles, (int)(a.K >> 6), sched_ == 2 && (!small || small_sched == 2), xlist && wlist ? 32 : 8);
        a.splits = 1;
        if (S > 1) {
            SplitWorkspace* w = split_workspace(st, (size_t)tiles * S * (small ? 128 : 256) * 256 * 4, (int)tiles);
            if (w) {                                       // (none -- growth under graph capture, or no memory: unsplit)
                a.splits = S;
                a.slabs = w->slabs;
                a.tickets = w->tickets;
                tiles *= S;
            }
        }
    }
    // diagnostic builds: MI355Q_V8_CLOCK prints the clock held over the K loop (no add-back),
    // MI355Q_V8_STAMPS the duration of the kernel's phases
    static const bool want_clock = getenv("MI355Q_V8_CLOCK") != nullptr, want_stamps = getenv("MI355Q_V8_STAMPS") != nullptr;
    const bool fix = xlist && wlist;
    if (fix && (!xf || !wf)) return MI355Q_E_BADARG;
    a.dbg = 0;
    // K-loop schedule of the 256 x 256 tile: 2 = pipelined (one barrier per K-step, default: 71.0 vs 72.8 us at 4096^3),
    // 0 = two staggered wave groups, four barriers per K-step (kept for A/B runs: MI355Q_V8_SCHED=0)
    constexpr int sched = 2;             // (the four-barrier schedule only serves K % 128 == 64 now; its switch went in round 5)
    const unsigned grid = tiles;            // (on a bucket overflow the tile workgroups themselves form the product blockwise)
    // the 256 x 256 tile has its own kernel since round 3 (mi355q_gemm_v9.hip); MI355Q_V9=0 keeps the round-2 one for A/B runs
    static const int use_v9 = getenv("MI355Q_V9") ? atoi(getenv("MI355Q_V9")) : 1;
    static const int v9_dbg = getenv("MI355Q_V9_DBG") ? atoi(getenv("MI355Q_V9_DBG")) : 0;
    if (use_v9) a.dbg = v9_dbg;
    // (grouped launches stay here: their outputs are promised bit-identical to the separate calls, which may take 128-row tiles)
    // products WITH exception lists stay on the kernel below unless MI355Q_V9_FIX=1: on one box, bench.py, the two take 67.0
    // (here) and 70.0 us (there; profiles/r03_v9_exception_designs.txt); without lists the new kernel takes 55-57 against 57.4
    // launches WITH exception lists: the 256 x 256 kernel of mi355q_gemm_v9.hip (add-back behind its K loop, nothing in front
    // of it) where the lists are all but empty -- operands of <= 5 bits, whose int8 container leaves a window of >= 4
    // exponents: 54.0-55.0 us at 4096^3 W4A4 / W5A5 against 55.8-56.5 here -- and this kernel (add-back in the prologue,
    // hidden behind the first stages) where every tile has its twenty entries: W6A6 62.3 against 65.0 (back-to-back
    // launches, tools/dbg/v9_widths.py).  MI355Q_V9_FIX=0 / 1 pins either.
    // Round 4: the 256 x 256 kernel takes EVERY launch with lists -- its add-back now rides the K loop's tail (the gathers of the
    // tile's first 24 entries in the LDS-DMA slots of the three K-steps past the end, one vector per entry formed by all waves,
    // a one-pass store epilogue): W6A6 62.1 against 63.1 us here on one box, 60.7 against 62.0 on another
    // (profiles/r04_v9_tail_prefetch.txt).  MI355Q_V9_FIX=0 keeps the round-2 kernel for A/B runs.
    static const int v9_fix_env = getenv("MI355Q_V9_FIX") ? atoi(getenv("MI355Q_V9_FIX")) : -1;
    const bool v9_fix = v9_fix_env >= 0 ? v9_fix_env != 0 : true;
    // (grouped launches stay on the kernel below: their outputs are promised bit-identical to the separate calls, which may take
    //  128-row tiles there -- the two kernels add a row's corrections in different fp32 orders)
    // (round 5: grouped launches too -- the small-tile kernel their separate launches may take adds a row's corrections in this
    //  kernel's order, tests/test_gpu_gemm.py::test_small_tiles_equal_the_256_tile_bit_for_bit)
        if (use_v9 && (v9_fix || !fix) && !small && sched == 2 && a.K % 128 == 0 && (a.K >> 6) / (a.splits > 1 ? a.splits : 1) >= 4 && !want_clock && !want_stamps)
        return launch_bfp_gemm_v9(a, sx, sw, xlist, wlist, st, xf, wf, false);
    if (small) {
        const bool piped = small_sched == 2 && a.K % 128 == 0 && (((a.K >> 6) / (a.splits > 1 ? a.splits : 1)) & 1) == 0;
        if (fix && piped) hipLaunchKernelGGL((bfp_gemm_v8<1, 4, 2>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
        else if (piped) hipLaunchKernelGGL((bfp_gemm_v8<0, 4, 2>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
        else if (fix) hipLaunchKernelGGL((bfp_gemm_v8<1, 4>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
        else hipLaunchKernelGGL((bfp_gemm_v8<0, 4>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    } else if (fix && want_stamps && sched == 2) hipLaunchKernelGGL((bfp_gemm_v8<3, 8, 2>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else if (fix && want_stamps) hipLaunchKernelGGL((bfp_gemm_v8<3, 8>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else if (fix && sched == 2) hipLaunchKernelGGL((bfp_gemm_v8<1, 8, 2>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else if (fix) hipLaunchKernelGGL((bfp_gemm_v8<1, 8>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else if (want_clock && sched == 2) hipLaunchKernelGGL((bfp_gemm_v8<2, 8, 2>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else if (want_clock) hipLaunchKernelGGL((bfp_gemm_v8<2, 8>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else if (sched == 2 && a.K % 128 == 0) hipLaunchKernelGGL((bfp_gemm_v8<0, 8, 2>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    else hipLaunchKernelGGL((bfp_gemm_v8<0, 8>), grid, V8_NT, 0, st, a, sx, sw, xlist, wlist, xf, wf);
    return (int)hipGetLastError();
}

// y = x . w^T (+ bias) on TILED bf16 operands (the same 1-KiB pieces: 16 rows x 32 values): a.xm / a.wm point at the
// bf16 pieces and a.K is the contraction length IN BYTES (2 K).  K % 32 == 0.
int launch_bf16_gemm_tiled(const GemmArgs& a_in, hipStream_t st) {
    if (v10_forced() && a_in.K % 64 == 0) return v10_forced_launch(a_in, nullptr, nullptr, nullptr, nullptr, st, nullptr, nullptr, true, v10_forced());
    GemmArgs a = a_in;
    const long long tn = (a.N + V8_BN - 1) / V8_BN;
    const long long t256 = ((a.M + 255) / 256) * tn, t128 = ((a.M + 127) / 128) * tn;
    // Tile height and split together, by a small cost model fitted to tools/timing/sweep_bf16_tile_split.py (us): rounds x steps
    // per slice x 0.70 (256 rows) or 0.56 (128 rows) per K-step, plus, when split, 12 + 0.7 per MiB of slab traffic (S x
    // the fp32 output through memory twice).  It ranks the measured settings of the post-activation layer shapes
    // correctly: 2048 x 11008 x 4096 -> 256 rows in two slices (177 vs 192 us), 2048 x 8192 x 2048 -> 256 rows in four.
    const char* force = getenv("MI355Q_V8_TILE_ROWS");          // (sweeps pin either flavour)
    const int nsteps_all = (int)(a.K >> 6);
    const double out_mib = (double)a.M * (double)a.N * 4.0 / 1048576.0;
    bool small = false;
    int S = 1;
    double best_t = 1e30;
    for (int kind = 0; kind < 2; ++kind) {                      // 0: 256 rows, 1: 128 rows
        if (force && atoi(force) && (atoi(force) == 128) != (kind == 1)) continue;
        const long long t = kind ? t128 : t256;
        const int smax = choose_splits(t, nsteps_all, kind == 0 && a.K % 128 == 0);
        for (int sp = 1; sp <= smax; ++sp) {
            if (getenv("MI355Q_V8_SPLITS") && sp != smax) continue;                      // (sweeps pin the split too)
            if (nsteps_all % sp || (kind == 0 && a.K % 128 == 0 && ((nsteps_all / sp) & 1))) continue;
            const double rounds = (double)((t * sp + 255) / 256);
            const double est = rounds * (nsteps_all / sp) * (kind ? 0.56 : 0.70) + (sp > 1 ? 12.0 + 0.7 * out_mib * sp : 0.0);
            if (est < best_t) { best_t = est; small = kind == 1; S = sp; }
        }
    }
    // Round 5: the small tiles of mi355q_gemm_v10.hip, unsplit, where their estimate is lower -- per K-step and tile 0.19 us
    // (128 x 128, up to two a compute unit side by side) / 0.39 us (128 x 256, one a compute unit), + 8 us per launch
    // (profiles/r05_small_tiles.txt: 2048 x 2048 x 8192 81.3 -> 59.0 us, Llama-7B o_proj 74.8 -> 54.6, down_proj 156 -> 142)
    {
        static const int v10_auto = getenv("MI355Q_V10_AUTO") ? atoi(getenv("MI355Q_V10_AUTO")) : 1;
        if (v10_auto && !force && !getenv("MI355Q_V8_SPLITS") && a.K % 64 == 0) {
            const long long g3 = ((a.M + 127) / 128) * ((a.N + 127) / 128), g1 = ((a.M + 127) / 128) * ((a.N + 255) / 256);
            // (128 x 128: 0.23 us a K-step alone on a compute unit, 0.43 for two side by side -- rounds of 512 tiles; beyond ~1000 tiles the
            //  256 x 256 kernel is ahead again although the line says otherwise: 2048 x 11008 x 4096 took 200 us here against its 165-179,
            //  hence the margin.  profiles/r05_shard_shapes.txt, r05_column_offsets.txt)
            // (end of round 5: 128 x 64 tiles where 128 x 128 ones fill half the compute units or fewer -- 0.12 us a K-step:
            //  4096 x 512 x 4096 30.1 -> 23.1 us, 2048 x 768 x 3072 22.7 -> 17.4, 2048 x 256 x 2048 16.0 -> 11.9)
            const double est3 = (g3 <= 128 ? nsteps_all * 0.12 : g3 <= 256 ? nsteps_all * 0.23 : nsteps_all * 0.43 * (double)((g3 + 511) / 512)) + 8.0;
            const double est1 = g1 <= 256 ? nsteps_all * 0.39 + 8.0 : 1e30;
            // (round 6: two K-groups in an 8-wave workgroup for 129 .. 256 tiles of 128 x 128 -- 0.20 us a K-step: 2048 x 2048 x 8192
            //  64.0 -> 58.4 us, profiles/r06_shard_shapes.txt)
            static const int kg_auto = getenv("MI355Q_V10_KG") ? atoi(getenv("MI355Q_V10_KG")) : 1;
            const double est5 = kg_auto && g3 > 128 && g3 <= 256 && a.K % 128 == 0 && a.x_segs <= 1 ? nsteps_all * 0.20 + 8.0 : 1e30;
            if (est3 * 1.12 < best_t || est1 * 1.12 < best_t || est5 * 1.12 < best_t) {
                a.splits = 1;
                const int geom = est5 < est3 && est5 < est1 ? 5 : (est1 < est3 ? 1 : (g3 <= 128 ? 4 : 3));
                return launch_bfp_gemm_v10(a, nullptr, nullptr, nullptr, nullptr, st, nullptr, nullptr, true, geom);
            }
        }
    }
    unsigned tiles = (unsigned)(small ? t128 : t256);
    {
        a.splits = 1;
        if (S > 1) {
            SplitWorkspace* w = split_workspace(st, (size_t)tiles * S * (small ? 128 : 256) * 256 * 4, (int)tiles);
            if (w) {                                       // (none -- growth under graph capture, or no memory: unsplit)
                a.splits = S;
                a.slabs = w->slabs;
                a.tickets = w->tickets;
                tiles *= S;
            }
        }
    }
    constexpr int small_sched = 2;          // (the one-phase schedule only serves K % 128 == 64 now; its switch went in round 5)
    static const bool use_v9_bf16 = !(getenv("MI355Q_V9") && atoi(getenv("MI355Q_V9")) == 0);        // (read once: ADVICE r4)
    if (small && small_sched == 2 && a.K % 128 == 0 && (((a.K >> 6) / (a.splits > 1 ? a.splits : 1)) & 1) == 0)
        hipLaunchKernelGGL((bfp_gemm_v8<0, 4, 2, true>), tiles, V8_NT, 0, st, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    else if (small) hipLaunchKernelGGL((bfp_gemm_v8<0, 4, 1, true>), tiles, V8_NT, 0, st, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    else if (a.x_segs <= 1 && a.K % 128 == 0 && (a.K >> 6) / (a.splits > 1 ? a.splits : 1) >= 4 && use_v9_bf16) return launch_bfp_gemm_v9(a, nullptr, nullptr, nullptr, nullptr, st, nullptr, nullptr, true);
    else if (a.K % 128 == 0) hipLaunchKernelGGL((bfp_gemm_v8<0, 8, 2, true>), tiles, V8_NT, 0, st, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((bfp_gemm_v8<0, 8, 0, true>), tiles, V8_NT, 0, st, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    return (int)hipGetLastError();
}

}  // namespace mi355q
