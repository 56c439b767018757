// mi355q_gemm_v10.hip -- the SMALL tiles of the row-scale block-floating-point GEMM (gfx950), round 5.
//
//     y[m,n] = sx[m] * sw[n] * ( sum_k xm'[m,k] * wm'[n,k] )  (+ bias[n])  (+ exception blocks),  K <= 16384, K % 64 == 0
// on ROW-aligned tiled operands (mi355q_align_row.h, mi355q_gemm_v2.h) -- the arithmetic of mi355q_gemm_v9.hip (and its bf16
// flavour: tiled bf16 operands, fp32 accumulators, no scales) for the shapes whose 256 x 256 tiles leave compute units idle:
// per-rank shards of a row-sharded layer (4096 x 512 x 4096), OPT-1.3B's 2048-wide projections, Llama-7B's v_proj at 2048
// tokens.  Reference path: quantized_modules/linear.py:59-76 (F.linear on the fake-quantised operands).
//
// One kernel template, geometry as parameters: NWM x NWN waves of (TI x 16) x (TJ x 16) outputs each
//     128 x 256:  1 x 4 waves of 128 x 64  (the 256 x 256 kernel's wave tile: 384 B of LDS fragment reads per MFMA)
//     256 x 128:  2 x 2 waves of 128 x 64
//     128 x 128:  2 x 2 waves of  64 x 64  (512 B per MFMA: for grids that would otherwise leave SIMDs idle)
// KG = 2 (round 6): TWO K-GROUPS in one 8-wave workgroup -- waves 0-3 take the even K-steps, waves 4-7 the odd ones of the SAME tile,
// each group through its own ring on the same barriers; behind the loop the groups swap half of their accumulators through LDS
// (int32 sums: exact in any order; fp32: two terms) and each stores its half of the tile.  For grids of at most one workgroup a
// compute unit, where a four-wave workgroup leaves every wave ALONE on its SIMD (355 clocks a K-step against 256 of MFMA,
// tools/ubench/vmem_issue.hip) and splitting K ACROSS workgroups costs a 4-us exchange between XCDs: two waves a SIMD, one prologue,
// one epilogue, the exchange inside the compute unit.
// FOUR waves a workgroup and <= 80 KiB of LDS, so that TWO workgroups share a compute unit (two waves per SIMD, as in the
// 256 x 256 kernel, but each pair of tiles runs its own barriers: one workgroup's prologue / store epilogue hides behind the
// other's K loop), THREE for the 128 x 128 tile.  The 128-row tile of mi355q_gemm_v8.hip (8 waves of 64 x 64, one workgroup a
// compute unit: 64 KiB of fragment reads per K-step for 515 clocks of MFMA) measured 1050 clocks a K-step
// (profiles/r04_shard_shapes.txt).
//
// K loop: the pipelined one-barrier schedule of the 256 x 256 kernel, generated from the geometry -- a ring of NS stages
// (A pieces then B pieces of one K-step of 64 bytes, 1-KiB pieces filled by buffer_load ... lds, the K-step in the scalar
// offset, steps past the end through a descriptor of zero bytes), fragment i + 2 read while group i's MFMAs issue, the next
// step's B fragments read into the other register set, one LDS-DMA piece per MFMA group, every wait counted
// (V10Sched::wait is the issue-order arithmetic the 256 x 256 kernel's hand-written counts came from).
//
// Exceptions (blocks outside their row's exponent window): added back behind the K loop from the ring area, one vector per
// entry formed by all waves with ONE round trip of gathers per batch, exception x exception terms by the wave that wrote the
// vector, rows with several entries folded into their first in ascending block order (reproducible), one-pass store epilogue.
// Bucket overflow: the launch's workgroups share the blockwise-exact product (v8_fallback).  Split-K: slabs + tickets.
// Roofline: int8 / bf16 MFMA, 2*M*N*K ops.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "mi355q_gemm_tile.h"

namespace mi355q {

typedef __bf16 v10_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x4 v10_mma(const i32x4& fw, const i32x4& fx, const i32x4& c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(fw, fx, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 v10_mma(const i32x4& fw, const i32x4& fx, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v10_bf16x8, fw), __builtin_bit_cast(v10_bf16x8, fx), c, 0, 0, 0);
}

template <int I, int N, class F>
__device__ __forceinline__ void v10_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        v10_for<I + 1, N>(f);
    }
}
template <int N> __device__ __forceinline__ void v10_waitv() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void v10_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ void v10_dsr(i32x4& d, int addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
#define V10_SB() __builtin_amdgcn_sched_barrier(0)
// split-K slabs: 16-byte stores / loads at AGENT scope (sc1: through the workgroup's own L2 to the level every XCD sees) -- the
// release / acquire fences they replace write back and invalidate the whole L2 of the XCD (buffer_wbl2 / buffer_inv), with every
// workgroup of the grid doing so at once: ~10 us of a 24-us launch (tools/dbg/v10_stamps.py, round 5)
template <class T> __device__ __forceinline__ void v10_store_agent(T* p, const T& v) {
    static_assert(sizeof(T) == 16, "one dwordx4");
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <class T> __device__ __forceinline__ void v10_load_agent(T& v, const T* p) {
    static_assert(sizeof(T) == 16, "one dwordx4");
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
// LDS-DMA as inline assembly (the compiler must not know it is pending: mi355q_gemm_v9.hip); M0 written in the statement
#define V10_BLDS16(vo, rs, so, lds) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so), "s"(lds) : "memory")
#define V10_BLDS4(vo, rs, so, lds) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so), "s"(lds) : "memory")
__device__ __forceinline__ i32x4 v10_desc(const void* base, int bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    return i32x4{(int)(unsigned)b, (int)(unsigned)(b >> 32), bytes, 0x00020000};
}

// Issue order of a K-step's LDS reads (per wave; LDS reads return in issue order, so a wait is a count):
//   group g:  read A fragment g + 2 (of this step, or fragment g + 2 - TI of the next) | WAIT | MFMA 0 | next step's B
//   fragments placed behind MFMA 0 | MFMA 1 | LDS-DMA piece | MFMA 2 | next step's B fragments placed behind MFMA 2 | MFMA 3
// B fragment j of the next step is read in group gb(j): TI = 8: groups 2..5, one each; TI = 4: groups 0 and 1, two each.
template <int TI, int TJ>
struct V10Sched {
    static_assert((TJ == 4 && (TI == 8 || TI == 4)) || (TJ == 2 && TI == 4), "geometry (TI = 4 runs the DEEP schedule, not this one)");
    static constexpr int gb(int j) { return TI == 8 ? 2 + j : j / 2; }
    static constexpr int sb(int j) { return TI == 8 ? 0 : (j & 1) * 2; }       // behind which MFMA of its group
    static constexpr int nb_before(int g) {                                   // B reads issued in groups < g
        int n = 0;
        for (int j = 0; j < TJ; ++j) n += gb(j) < g ? 1 : 0;
        return n;
    }
    static constexpr int L = TI + TJ;                                          // reads a step
    static constexpr int ea(int g) { return g + nb_before(g); }                // index (in its step) of group g's A read
    static constexpr int eb(int j) {                                           // ... of the read of next-step B fragment j
        int n = 0;
        for (int q = 0; q < TJ; ++q) n += (gb(q) == gb(j) && sb(q) < sb(j)) ? 1 : 0;
        return ea(gb(j)) + 1 + n;
    }
    // reads that may still be in flight at group g's wait: everything issued behind the read of A fragment g (two groups
    // earlier), group 0 also behind the last B fragment of THIS step (read during the previous one)
    static constexpr int wait(int g) {
        const int at = ea(g);
        const int need_a = g >= 2 ? ea(g - 2) : ea(TI - 2 + g) - L;
        int w = at - need_a;
        if (g == 0) {
            int last_b = 0;
            for (int j = 0; j < TJ; ++j) last_b = eb(j) > last_b ? eb(j) : last_b;
            const int wb = at - (last_b - L);
            w = wb < w ? wb : w;
        }
        return w;
    }
};
static_assert(V10Sched<8, 4>::wait(0) == 2 && V10Sched<8, 4>::wait(1) == 2 && V10Sched<8, 4>::wait(2) == 2 && V10Sched<8, 4>::wait(3) == 3 &&
              V10Sched<8, 4>::wait(4) == 4 && V10Sched<8, 4>::wait(5) == 4 && V10Sched<8, 4>::wait(6) == 4 && V10Sched<8, 4>::wait(7) == 3,
              "the 256 x 256 kernel's hand-counted waits");

// LDS beside the ring (byte offsets): scale / bias slices (1 KiB each whatever the tile: the loads are 1 KiB),
// the two maps, flag words, the two lists' header words and the two buckets' first words
constexpr int V10_SXT = 0, V10_SWT = 1024, V10_BIAS = 2048, V10_FLAGS = 3072, V10_OVF = V10_FLAGS + 256;
constexpr int V10_HEAD = V10_OVF + 512, V10_LIVE = V10_HEAD + 512, V10_MAP = V10_LIVE + 512;     // (the maps: (BM + BN) words, last)

template <int NWM, int NWN, int TI, int TJ, int NS, int OCC, int FIX, bool BF16, int KG = 1>
__global__ __launch_bounds__(64 * NWM * NWN * KG) __attribute__((amdgpu_waves_per_eu((OCC < 2 ? 2 : OCC), (OCC < 2 ? 2 : OCC)))) void bfp_gemm_v10(const GemmArgs a_in, const float* __restrict__ sx,
                                                                    const float* __restrict__ sw_in, const int* __restrict__ xlist,
                                                                    const int* __restrict__ wlist_in, const uint8_t* __restrict__ xf,
                                                                    const uint8_t* __restrict__ wf_in) {
    static_assert(!BF16 || FIX == 0, "the bf16 arithmetic has no exception lists");
    static_assert(KG == 1 || (KG == 2 && OCC == 1 && TI % 2 == 0), "K-groups: two, one workgroup a compute unit");
    // NWG: waves of one K-group (they own the tile's wave tiles and stage its ring); NW / NT: all waves / threads of the workgroup
    // (the bookkeeping, the exception vectors and the fallback are shared work)
    constexpr int NWG = NWM * NWN, NW = NWG * KG, NT = 64 * NW, WM = TI * 16, WN = TJ * 16, BM = NWM * WM, BN = NWN * WN;
    constexpr int PA = BM / 16, PB = BN / 16, NP = PA + PB, LPW = NP / NWG, LPA = PA / NWG;
    static_assert(NP % NWG == 0 && PA % NWG == 0, "a wave stages whole pieces of one operand per index");
    constexpr int STAGE = NP * 1024, RING = NS * STAGE;
    static_assert(NS >= 3 && (NS - 3) * LPW <= 63, "ring: NS - 2 K-steps of LDS-DMA in flight, counted by vmcnt");
    constexpr int V10_SIDE = V10_MAP + (BM + BN) * 4;
    static_assert(KG * RING + V10_SIDE <= 160 * 1024 / OCC, "LDS for OCC workgroups a compute unit");
    static_assert(KG == 1 || NW * (TI / 2) * TJ * 1024 <= KG * RING, "the accumulator exchange of the two K-groups fits the rings");
    static_assert(RING >= 40 * 1024, "the blockwise fallback's LDS, the buckets and vectors behind the K loop");
    static_assert(NT == 256 * KG, "the blockwise fallback runs a 256-thread team (the waves beyond it leave)");
    using S = V10Sched<TI, TJ>;
    constexpr int VLEN = BM > BN ? BM : BN;                     // floats of a correction vector
    // EARLY: LDS to spare for the tile's two exception buckets beside the ring -- they arrive in front of the operand stream,
    // the bookkeeping runs in the shadow of the first stages' flight and (UE > 0: the 128 x 128 tile, which has the registers)
    // every wave requests the gathers of its first UE entries BEFORE the K loop and keeps them in registers across it: behind the
    // loop only the vectors remain to be formed.  Otherwise the buckets are fetched behind the loop into the ring (two dependent
    // round trips exposed: ~10 of the 28 us of 4096 x 512 x 4096 were this, profiles/r05_small_tiles.txt).
    constexpr bool EARLY = FIX && KG * RING + V10_SIDE + 8192 <= 160 * 1024 / OCC;
    constexpr int UE = EARLY && TI == 4 ? 6 : 0, VC = VLEN / 64, G = UE * VC;      // G: gather loads a wave issues in front of the loop
    constexpr int VOFF = EARLY ? 0 : 8192;                      // vectors behind the K loop: the ring (behind the bucket copies)
    constexpr int VCAP = (RING - VOFF) / (VLEN * 4) < 128 ? (RING - VOFF) / (VLEN * 4) : 128;
    // The ring is DYNAMIC shared memory (RING bytes, v10_launch): a kernel whose static LDS admits only one workgroup a compute
    // unit is compiled for one wave per SIMD whatever its launch bounds say -- the accumulators then move to the accumulation
    // registers and are copied in and out around every MFMA of the K loop (NS = 4: 65 us where NS = 3 took 48).
    extern __shared__ __attribute__((aligned(16))) unsigned char v10_ring[];
    unsigned char* const ring = v10_ring;
    __shared__ __attribute__((aligned(16))) unsigned char side[V10_SIDE];
    __shared__ __attribute__((aligned(16))) unsigned char bk[EARLY ? 8192 : 16];
    static_assert(V10_SIDE % 16 == 0, "the dynamic part starts 16-byte aligned");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = KG > 1 ? wave / NWG : 0, wv = KG > 1 ? wave % NWG : wave;       // K-group, wave inside it
    const int wm = wv / NWN, wn = wv % NWN, l16 = lane & 15, lq = lane >> 4;

#ifdef V10_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define V10_STAMP(k) st_[k] = __builtin_amdgcn_s_memrealtime()
#else
#define V10_STAMP(k)
#endif
    V10_STAMP(0);
    GemmArgs a = a_in;
    const float* __restrict__ sw = sw_in;
    const int* __restrict__ wlist = wlist_in;
    const uint8_t* __restrict__ wf = wf_in;
    const int ngroup = a.ngroup > 1 ? a.ngroup : 1;
    const int Mi = (int)a.M, Ni = (int)a.N;
    const int tiles_m = (Mi + BM - 1) / BM, tiles_n1 = (Ni + BN - 1) / BN, tiles_n = tiles_n1 * ngroup;
    const int S_ = a.splits > 1 ? a.splits : 1;                // workgroups per tile (split-K)
    const int nwg = tiles_m * tiles_n * S_;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int split = S_ > 1 ? pid % S_ : 0, tile_id = S_ > 1 ? pid / S_ : pid;
    const int GM = 256 / BM * 4, in_group = GM * tiles_n, group_id = tile_id / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (tile_id % in_group) % gsz;
    int tn = (tile_id % in_group) / gsz;
    if (ngroup > 1) {
        const int which = tn / tiles_n1;                        // (wave-uniform: scalar loads from the argument block)
        tn -= which * tiles_n1;
        // (selects over constant indices: a runtime index would put the argument block in scratch memory)
#define V10_PICK(f) (which == 0 ? a_in.f[0] : which == 1 ? a_in.f[1] : a_in.f[2])
        a.wm = V10_PICK(g_wm); a.we = V10_PICK(g_we); a.bias = V10_PICK(g_bias); a.y = V10_PICK(g_y);
        sw = V10_PICK(g_sw); wlist = V10_PICK(g_wlist); wf = V10_PICK(g_wf);
#undef V10_PICK
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kp = (int)(a.K >> 6);                             // 1-KiB pieces per 16 rows
    const int kstep0 = S_ > 1 ? (int)((long long)kp * split / S_) : 0;      // this workgroup's slice of the K-steps
    // (KG = 2: never split across workgroups; this group's K-steps are kstep0 + 2 s + grp, nsteps of them -- K % 128 == 0)
    const int nsteps = (S_ > 1 ? (int)((long long)kp * (split + 1) / S_) - kstep0 : kp) / KG;

    float* sxt = reinterpret_cast<float*>(side + V10_SXT);
    float* swt = reinterpret_cast<float*>(side + V10_SWT);
    float* bst = reinterpret_cast<float*>(side + V10_BIAS);
    int* rowslot = reinterpret_cast<int*>(side + V10_MAP);
    int* colslot = rowslot + BM;
    int* flags = reinterpret_cast<int*>(side + V10_FLAGS);
    const int ring_lds = (int)(size_t)(lptr_t)ring, side_lds = (int)(size_t)(lptr_t)side;

    // ---- in front of the operand stream: the tile's scale / bias slices (bounds-checked by their descriptors: rows and
    //      columns past the operand read as zero), the lists' header words, the first words of the tile's two buckets
    if (grp == 0) {
        const int w4 = wave & 3;
        if (!BF16 && w4 == 0) V10_BLDS16(lane * 16, v10_desc(sx + m0, (Mi - m0) * 4), 0, side_lds + V10_SXT);
        if (!BF16 && w4 == 1) V10_BLDS16(lane * 16, v10_desc(sw + n0, (Ni - n0) * 4), 0, side_lds + V10_SWT);
        if (w4 == 2) {
            if (a.bias) V10_BLDS16(lane * 16, v10_desc(a.bias + n0, (Ni - n0) * 4), 0, side_lds + V10_BIAS);
            else *reinterpret_cast<f32x4*>(bst + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (FIX && w4 == 3) {
            V10_BLDS4(lane * 4, v10_desc(xlist, 256), 0, side_lds + V10_OVF);
            V10_BLDS4(lane * 4, v10_desc(wlist, 256), 0, side_lds + V10_OVF + 256);
        }
        if constexpr (EARLY) {
            const int bk_lds = (int)(size_t)(lptr_t)bk;
            if (w4 == 0 || (w4 == 1 && !a.x_post)) {
                const i32x4 bd = v10_desc(w4 == 0 ? row_bucket(wlist, n0) : row_bucket(xlist, m0), ROW_BUCKET_WORDS * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) V10_BLDS16(lane * 16 + q * 1024, bd, 0, bk_lds + (w4 == 0 ? 4096 : 0) + q * 1024);
            }
        } else {
            if (FIX && w4 == 0) V10_BLDS4(lane * 4, v10_desc(row_bucket(wlist, n0), 256), 0, side_lds + V10_HEAD + 256);
            if (FIX && w4 == 1 && !a.x_post) V10_BLDS4(lane * 4, v10_desc(row_bucket(xlist, m0), 256), 0, side_lds + V10_HEAD);
        }
    }

    // ---- the operand stream.  Piece p of a K-step: 16 rows of A (p < PA) or of B; this wave stages pieces wave + NW q.
    const long long row_bytes = (long long)kp * 1024;           // one piece row (16 rows x K bytes)
    // (bf16 flavour, a.x_segs > 1: x lies as column segments -- [segment][row piece][K-steps of the segment], a.x_seg_stride bytes
    //  apart -- where an all-gather of per-rank quantised slices left them (mi355q_bf16_gemm_tiled_seg; mi355q_gemm_v8.hip had the only
    //  kernel that read them until the end of round 5): a K-step of a row piece is found through its segment)
    const bool xseg = BF16 && a.x_segs > 1;
    const int sps = xseg ? kp / a.x_segs : kp;                  // K-steps per segment
    const unsigned sps_inv = xseg && sps > 1 ? 0xFFFFFFFFu / (unsigned)sps + 1u : 0u;   // (ks / sps == umulhi(ks, sps_inv) for the few thousand K-steps there are)
    const long long row_bytes_a = (long long)sps * 1024;        // (== row_bytes without segments)
    const int pa_rows = min(PA, ((Mi + 127) >> 7) * 8 - (m0 >> 4)), pb_rows = min(PB, ((Ni + 127) >> 7) * 8 - (n0 >> 4));
    const int8_t* xbase = a.xm + (long long)(m0 >> 4) * row_bytes_a + (xseg ? 0ll : (long long)kstep0 * 1024);
    const int8_t* wbase = a.wm + (long long)(n0 >> 4) * row_bytes + (long long)kstep0 * 1024;
    const int x_nrec = (int)(pa_rows * row_bytes_a), w_nrec = (int)(pb_rows * row_bytes);
    int voff[LPW];
#pragma unroll
    for (int q = 0; q < LPW; ++q) voff[q] = (wv + NWG * (q < LPA ? q : q - LPA)) * (int)(q < LPA ? row_bytes_a : row_bytes) + lane * 16;
    // piece q (literal) of K-step `step` into the stage at byte offset `so`; a step past the slice's end is requested through
    // descriptors of zero bytes (no memory traffic, zeros land)
    auto piece = [&](auto qi, int step, int so) {
        constexpr int q = decltype(qi)::value;
        constexpr bool isa = q < LPA;
        // (operands of the asm statement as locals of the lambda: an asm operand does not capture by itself)
        const int vo = voff[q];
        const int8_t* src = isa ? xbase : wbase;
        int soff = (KG * step + grp) * 1024;
        if (BF16 && isa && xseg) {                              // (uniform)
            const unsigned ks = (unsigned)(kstep0 + KG * step + grp), sg = sps > 1 ? __umulhi(ks, sps_inv) : ks;
            src = xbase + (long long)sg * a.x_seg_stride;
            soff = (int)(ks - sg * (unsigned)sps) * 1024;
        }
        const i32x4 rd = v10_desc(src, step < nsteps ? (isa ? x_nrec : w_nrec) : 0);
        const int dst = ring_lds + grp * RING + so + ((isa ? 0 : PA) + wv + NWG * (isa ? q : q - LPA)) * 1024;
        V10_BLDS16(vo, rd, soff, dst);
    };
    v10_for<0, NS - 1>([&](auto si) {
        constexpr int s = decltype(si)::value;
        v10_for<0, LPW>([&](auto qi) { piece(qi, s, s * STAGE); });
    });

    using acc_t = typename std::conditional<BF16, f32x4, i32x4>::type;
    // lane-constant part of the fragment addresses; fragment i is i KiB further (immediate offset)
    const int va = ring_lds + grp * RING + piece_lds_off(wm * WM + l16, lq), vb = ring_lds + grp * RING + PA * 1024 + piece_lds_off(wn * WN + l16, lq);
    // DEEP (the 64-row wave tile): ALL of the next K-step's fragments are read during this step -- a wave that is alone on its SIMD
    // (grids of <= 256 workgroups: every launch this tile is chosen for) has no partner whose MFMAs cover the ~130 clocks of an LDS
    // round trip, and a two-group window leaves them exposed (tools/ubench/vmem_issue.hip: 455 -> 355 clocks a K-step)
    constexpr bool DEEP = TI == 4;
#ifndef V10_PAIR
#define V10_PAIR 1
#endif
    constexpr bool PAIR = V10_PAIR && DEEP && NS >= 6;          // (see `body`)
    i32x4 fa[DEEP ? 2 * TI : 4], fb[2][TJ];

    // ---- exception bookkeeping state.  Buckets: LDS copies of the tile's two buckets (EARLY: beside the ring, else in the ring
    //      behind the K loop); vectors: VLEN floats per entry in the ring behind the K loop
    int* const xb = reinterpret_cast<int*>(EARLY ? bk : ring);
    int* const wb = reinterpret_cast<int*>(EARLY ? bk + 4096 : ring + 4096);
    float* const vecs = reinterpret_cast<float*>(ring + VOFF);
    int cx = 0, cw = 0, nent = 0, mode = 0, nlive = 0;
    int x_off_s = a.x_off, w_off_s = a.w_off;                   // (opaque scalars: see mi355q_gemm_v9.hip)
    asm volatile("" : "+s"(x_off_s), "+s"(w_off_s));
    // (1) of the add-back, the bookkeeping: 16 lanes share an entry: slot = the list index of the entry of the same tile row /
    //     column with the SMALLEST BLOCK (a property of the data, not of the order in which rows reserved their list slots); -2
    //     marks an entry outside the tile; rows / columns without a vector keep -1 in the maps.  Barriers: the caller's.
    int* const live_idx = reinterpret_cast<int*>(side + V10_LIVE);      // list positions of the entries inside this tile
    auto clear_maps = [&]() {
        for (int o = tid; o < BM + BN; o += NT) rowslot[o] = -1;
        if (tid == 0) flags[16] = flags[17] = 0;               // (set by the bookkeeping: a row with several entries; live entries)
    };
    auto bookkeep = [&]() {
        for (int i0 = 0; i0 < nent; i0 += NT / 16) {           // uniform
            const int i = i0 + (tid >> 4), sub = tid & 15;
            const bool valid = i < nent, is_x = i < cx;
            int* e = v8_entry(xb, wb, cx, valid ? i : 0);
            const int r = e[0], base = is_x ? m0 : n0;
            const bool live = valid && (is_x ? (r >= m0 && r < m0 + BM && r < Mi) : (r >= n0 && r < n0 + BN && r < Ni));
            const int lo = is_x ? 0 : cx, hi = is_x ? cx : nent;
            int skey = (e[1] << 8) | i;
            if (live)
                for (int j = lo + sub; j < hi; j += 16) {
                    const int* ej = v8_entry(xb, wb, cx, j);
                    if (ej[0] == r) skey = min(skey, (ej[1] << 8) | j);
                }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) skey = min(skey, __shfl_xor(skey, o));
            const int slot = skey & 255;
            if (valid && sub == 0) {
                e[3] = live ? slot : -2;
                if (live && slot == i) (is_x ? rowslot : colslot)[r - base] = i;
                if (live && slot != i) flags[16] = 1;
                // the entries that lie in this tile, compacted (a bucket covers 256 rows: half of a 128-row tile's entries are
                // its neighbour's).  Which wave forms which vector does not change any sum.
                if (live) live_idx[atomicAdd(&flags[17], 1)] = i;
            }
        }
    };
    // (2) the gathers of entry i: the other operand's 16-byte blocks at the entry's K position for the tile's rows / columns (16
    //     rows' blocks at one K position are 256 contiguous bytes).  An entry that does not exist or lies outside the tile loads
    //     the operand's first bytes: the number of loads a wave issues must not depend on the data (counted waits).
    auto gather = [&](int4 (&qv)[VC], int i) {
        const bool on = i < nent;
        const bool is_x = i < cx;
        const int* e = v8_entry(xb, wb, cx, on ? i : 0);
        const bool use = on && e[3] != -2;
        const long long kcol = use ? (long long)e[1] * 16 : 0;
        const int8_t* qm = is_x ? +a.wm : +a.xm;                // (unary +: values, not a select of addresses in the argument block)
        const long long q0 = is_x ? n0 : m0, qmax = (is_x ? Ni : Mi) - 1;
#pragma unroll
        for (int c = 0; c < VC; ++c) {
            const long long q = use && c * 64 < (is_x ? BN : BM) ? min(q0 + c * 64 + lane, qmax) : 0;
            qv[c] = *reinterpret_cast<const int4*>(qm + tiled_offset(q, kcol, a.K));
        }
    };
    // (3) one vector per live entry: x entry (row r, block kb): v[n] = 2^(code - x_off) * sw[n] * dot16(entry, w'[n, kb]) over the
    //     tile's columns; w entries the mirror image over its rows; an x entry also takes the exception x exception terms (w
    //     entries at the same K position) -- by the wave that wrote the vector (LDS operations of a wave complete in order)
    auto form = [&](const int4 (&qv)[VC], int i) {
        if (i >= nent) return;                                  // (uniform)
        const bool is_x = i < cx;
        const int* e = v8_entry(xb, wb, cx, i);
        if (e[3] == -2) return;                                 // (uniform: outside the tile -- no vector)
        const int kb = e[1], code = e[2];
        const int4 pv = *reinterpret_cast<const int4*>(e + 4);
        const int sh = code - (is_x ? x_off_s : w_off_s);
        const float* sc = is_x ? swt : sxt;
        float* v = vecs + i * VLEN;
#pragma unroll
        for (int c = 0; c < VC; ++c)
            if (c * 64 < (is_x ? BN : BM)) v[c * 64 + lane] = __builtin_ldexpf((float)dot16(pv, qv[c]), sh) * sc[c * 64 + lane];
        if (is_x)
            for (int f0 = 0; f0 < cw; f0 += 64) {               // (uniform)
                const int fi = f0 + lane;
                if (fi < cw) {
                    const int* f = wb + EXC_HEADER + EXC_ENTRY * fi;
                    if (f[3] != -2 && f[1] == kb)
                        v[f[0] - n0] += __builtin_ldexpf((float)dot16(pv, *reinterpret_cast<const int4*>(f + 4)), code + f[2] - a.scale_bias);
                }
            }
    };
    auto overflowed = [&]() {
        // a bucket overflowed somewhere (uniform over the grid): the row-scale product does not apply; the workgroups of
        // this launch share the blockwise-exact product instead (the operand loads in flight land in LDS only)
        const int* ovf = reinterpret_cast<const int*>(side + V10_OVF);
        if (__builtin_amdgcn_readfirstlane(ovf[0] | ovf[64]) == 0) return false;
        v10_waitv<0>();
        __syncthreads();
        v8_fallback(a, xf, wf, xlist, wlist, ring, ngroup > 1 ? (tm * tiles_n1 + tn) * S_ + split : (int)blockIdx.x,
                    ngroup > 1 ? tiles_m * tiles_n1 * S_ : nwg);
        return true;
    };
    int4 pq[UE > 0 ? UE : 1][VC];                               // the gathers requested in front of the K loop
    V10_STAMP(1);
    if constexpr (EARLY) {
        clear_maps();
        v10_lgkm<0>();
        v10_waitv<(NS - 1) * LPW>();                            // scales, header words and buckets (older than every operand piece)
        __builtin_amdgcn_s_barrier();
        if (overflowed()) return;
        cx = a.x_post ? 0 : __builtin_amdgcn_readfirstlane(min(xb[0], ROW_BCAP));
        cw = __builtin_amdgcn_readfirstlane(min(wb[0], ROW_BCAP));
        nent = cx + cw;
        mode = nent == 0 ? 0 : (nent <= VCAP ? 1 : 3);
        if (mode) {                                             // (uniform over the workgroup; the maps were cleared at the start)
            bookkeep();
            v10_lgkm<0>();
            __builtin_amdgcn_s_barrier();
            nlive = __builtin_amdgcn_readfirstlane(flags[17]);
        }
        if constexpr (UE > 0) {
#pragma unroll
            for (int u = 0; u < UE; ++u) gather(pq[u], mode == 1 && wave + NW * u < nlive ? live_idx[wave + NW * u] : nent);
        }
    }
    V10_STAMP(2);
    v10_waitv<(NS - 2) * LPW + G>();                            // everything but the pieces of K-steps 1 .. NS - 2 (and the gathers)
    __builtin_amdgcn_s_barrier();
    V10_STAMP(3);
    if constexpr (FIX && !EARLY) {
        if (overflowed()) return;
    }
    v10_for<0, TJ>([&](auto ji) { constexpr int j = decltype(ji)::value; v10_dsr<j * 1024>(fb[0][j], vb); });
    v10_dsr<0>(fa[0], va);
    v10_dsr<1024>(fa[1], va);
    if constexpr (DEEP) {
        v10_dsr<2048>(fa[2], va);
        v10_dsr<3072>(fa[3], va);
    }
    V10_SB();
    acc_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    // K-step t between barrier(t) and barrier(t + 1).  At barrier(t) every wave has waited for its own pieces of step t + 1
    // (the later steps' stay in flight) and has retired every read of step t - 1's stage: that is the stage step t + NS - 1
    // goes to.  C: which B register set holds step t's fragments.
    // (the gathers requested in front of the loop are younger than the first stages' pieces: the first step's counted wait covers
    //  them too -- a peeled first step with its own count cost more registers than the half microsecond it would save)
    // PH (PAIR: the 64-row wave tile with a ring of >= 6 stages): K-steps two to a barrier.  0: a step on its own (as before);
    // 1: the first of a pair -- waits for this wave's pieces of BOTH following steps; 2: the second -- no wait for pieces, no barrier
    // (what it reads landed before the pair's barrier; the stage it refills, step t - 1's, was read into registers by every wave
    // before that barrier too).  A lone wave per SIMD pays ~170 clocks of barrier and LDS round trip a K-step whatever its MFMAs
    // (128 x 64 tile: 300 clocks a step for 128 of MFMA); pairs halve the number of those.
    auto body = [&](auto ci, auto phi, int t, int sc, int sn, int sd) {
        constexpr int C = decltype(ci)::value, PH = decltype(phi)::value;
        if constexpr (PH == 1) v10_waitv<(NS - 4 > 0 ? NS - 4 : 0) * LPW>();
        else if constexpr (PH == 0) v10_waitv<(NS - 3) * LPW>();
        if constexpr (DEEP) v10_lgkm<0>();                      // this step's fragments, read during the last one
        if constexpr (PH != 2) __builtin_amdgcn_s_barrier();
        const int ac = va + sc, an = va + sn, bn = vb + sn;
        if constexpr (DEEP) {
            // group 0: the next step's four B fragments, one behind each MFMA; group 1: its four A fragments; one LDS-DMA piece a
            // group behind MFMA 1
            (void)ac;
            v10_for<0, TI>([&](auto gi) {
                constexpr int g = decltype(gi)::value;
                v10_for<0, TJ>([&](auto ji) {
                    constexpr int j = decltype(ji)::value;
                    V10_SB();
                    acc[g][j] = v10_mma(fb[C][j], fa[C * TI + g], acc[g][j]);
                    V10_SB();
                    // the next step's TJ + TI fragment reads, in order B then A, behind the first MFMAs: one a MFMA with TJ = 4 (B in
                    // group 0, A in group 1), two a MFMA with TJ = 2 (the 128 x 64 tile: all six behind its first three MFMAs of
                    // eight -- a lone wave has only its own MFMAs to cover the reads' way back)
                    constexpr int RPS = TJ == 2 ? 2 : 1, slot = g * TJ + j;
                    v10_for<0, RPS>([&](auto ri) {
                        constexpr int r = slot * RPS + decltype(ri)::value;
                        if constexpr (r < TJ) v10_dsr<(r < TJ ? r : 0) * 1024>(fb[1 - C][r < TJ ? r : 0], bn);
                        else if constexpr (r < TJ + TI) v10_dsr<(r >= TJ && r < TJ + TI ? r - TJ : 0) * 1024>(fa[(1 - C) * TI + (r >= TJ && r < TJ + TI ? r - TJ : 0)], an);
                    });
                    if constexpr (j == 1 && g < LPW) piece(std::integral_constant<int, (g < LPW ? g : 0)>{}, t + NS - 1, sd);
                    if constexpr (j == 3 && g + TI < LPW) piece(std::integral_constant<int, (g + TI < LPW ? g + TI : 0)>{}, t + NS - 1, sd);
                });
            });
            V10_SB();
        } else
        v10_for<0, TI>([&](auto gi) {
            constexpr int g = decltype(gi)::value;
            if constexpr (g + 2 < TI) v10_dsr<(g + 2) * 1024>(fa[(g + 2) & 3], ac);
            else v10_dsr<(g + 2 - TI) * 1024>(fa[(g + 2) & 3], an);
            v10_lgkm<S::wait(g)>();
            V10_SB();
            acc[g][0] = v10_mma(fb[C][0], fa[g & 3], acc[g][0]);
            V10_SB();
            v10_for<0, TJ>([&](auto ji) {
                constexpr int j = decltype(ji)::value;
                if constexpr (S::gb(j) == g && S::sb(j) == 0) v10_dsr<j * 1024>(fb[1 - C][j], bn);
            });
            V10_SB();
            acc[g][1] = v10_mma(fb[C][1], fa[g & 3], acc[g][1]);
            V10_SB();
            if constexpr (g < LPW) piece(std::integral_constant<int, g>{}, t + NS - 1, sd);
            V10_SB();
            acc[g][2] = v10_mma(fb[C][2], fa[g & 3], acc[g][2]);
            V10_SB();
            v10_for<0, TJ>([&](auto ji) {
                constexpr int j = decltype(ji)::value;
                if constexpr (S::gb(j) == g && S::sb(j) == 2) v10_dsr<j * 1024>(fb[1 - C][j], bn);
            });
            V10_SB();
            acc[g][3] = v10_mma(fb[C][3], fa[g & 3], acc[g][3]);
            V10_SB();
            if constexpr (g + TI < LPW) piece(std::integral_constant<int, g + TI>{}, t + NS - 1, sd);
            V10_SB();
        });
    };
    {
        // ring positions (stage indices): `cur` holds step t, cur + 1 step t + 1, ..., cur - 1 is the one step t + NS - 1 goes to
        int cur = 0;
        auto nxt = [&](int s_) { return s_ + 1 == NS ? 0 : s_ + 1; };
        auto prv = [&](int s_) { return s_ == 0 ? NS - 1 : s_ - 1; };
        int t = 0;
        constexpr int P1 = PAIR ? 1 : 0, P2 = PAIR ? 2 : 0;
        for (; t + 1 < nsteps; t += 2) {
            body(std::integral_constant<int, 0>{}, std::integral_constant<int, P1>{}, t, cur * STAGE, nxt(cur) * STAGE, prv(cur) * STAGE);
            cur = nxt(cur);
            body(std::integral_constant<int, 1>{}, std::integral_constant<int, P2>{}, t + 1, cur * STAGE, nxt(cur) * STAGE, prv(cur) * STAGE);
            cur = nxt(cur);
        }
        if (t < nsteps) body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, t, cur * STAGE, nxt(cur) * STAGE, prv(cur) * STAGE);   // (odd slices: K % 128 == 64)
    }
    v10_waitv<0>();
    v10_lgkm<0>();                                              // (the compiler does not know these reads are in flight)
    V10_SB();
    V10_STAMP(4);
    __builtin_amdgcn_s_barrier();                               // (every wave is out of the ring)

    if constexpr (KG == 2) {
        // ---- the two K-groups' accumulators: group g keeps fragment rows g TI / 2 .. and hands the others to its partner wave
        //      (same wave tile, other group) through the ring area; every wave then owns complete sums for half of its wave tile
        constexpr int HI = TI / 2;
        acc_t* const xbuf = reinterpret_cast<acc_t*>(ring);
#pragma unroll
        for (int i = 0; i < HI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) xbuf[((wave * HI + i) * TJ + j) * 64 + lane] = grp == 0 ? acc[HI + i][j] : acc[i][j];
        __syncthreads();
        const int pw = grp == 0 ? wave + NWG : wave - NWG;
#pragma unroll
        for (int i = 0; i < HI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const acc_t part = xbuf[((pw * HI + i) * TJ + j) * 64 + lane];
                if (grp == 0) acc[i][j] += part;
                else acc[HI + i][j] += part;
            }
        __syncthreads();                                        // (the ring area is about to hold buckets / vectors)
    }
    if constexpr (KG == 1)
    if (S_ > 1) {
        // ---- split-K.  a.tickets holds two words a tile: arrivals, published slabs.
        // Order-free sums (int32: exact whatever the order; two slices of fp32: a + b == b + a): a slice takes its ticket FIRST; every
        // slice but the last to arrive publishes its raw accumulators in the slab of its arrival number and leaves; the last one keeps
        // its own in registers, waits until the others' slabs are published -- they took their tickets before it, so they are past their
        // K loops and publish without waiting for anybody: no co-residency is assumed -- and adds them.  Half the slab traffic of
        // everybody-stores, and no store in front of the finisher's ticket.
        // Ordered sums (fp32, more than two slices): every slice stores, the last arriver sums all slabs IN SLICE ORDER
        // (reproducible; mi355q_gemm_v9.hip, Guideline 16).
        // Slab accesses at agent scope (v10_store_agent / v10_load_agent) instead of release / acquire fences.
        constexpr long long SLAB = (long long)BM * BN * 4;
        acc_t* tslabs = reinterpret_cast<acc_t*>(static_cast<unsigned char*>(a.slabs) + (long long)tile_id * S_ * SLAB);
        int* arrivals = a.tickets + 2 * tile_id;
        int* published = arrivals + 1;
        int* flagw = flags + 8;
        const bool ordered = BF16 && S_ > 2;
        if (ordered) {
            acc_t* slab = tslabs + (long long)split * (SLAB / 16);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) v10_store_agent(&slab[((wave * TI + i) * TJ + j) * 64 + lane], acc[i][j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (agent-scope stores: acknowledged where every XCD reads them)
            __syncthreads();
            if (tid == 0) {
                const int tk = __hip_atomic_fetch_add(arrivals, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tk == S_ - 1) __hip_atomic_store(arrivals, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // idle again
                *flagw = tk;
            }
            __syncthreads();
            if (*flagw != S_ - 1) return;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
        } else {
            if (tid == 0) *flagw = __hip_atomic_fetch_add(arrivals, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int tk = __builtin_amdgcn_readfirstlane(*flagw);
            if (tk != S_ - 1) {
                acc_t* slab = tslabs + (long long)tk * (SLAB / 16);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) v10_store_agent(&slab[((wave * TI + i) * TJ + j) * 64 + lane], acc[i][j]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) (void)__hip_atomic_fetch_add(published, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (tid == 0) {
                __hip_atomic_store(arrivals, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                     // idle again
                while (__hip_atomic_load(published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != S_ - 1) __builtin_amdgcn_s_sleep(2);
                __hip_atomic_store(published, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        const int nslabs = ordered ? S_ : S_ - 1;
        for (int sl = 0; sl < nslabs; ++sl) {
            const acc_t* sp = tslabs + (long long)sl * (SLAB / 16);
            constexpr int CH = TI > 4 ? 2 : TI;                  // rows of tiles in flight together (registers: 16 CH)
            v10_for<0, TI / CH>([&](auto ci) {
                constexpr int i0 = decltype(ci)::value * CH;
                acc_t part[CH][TJ];
#pragma unroll
                for (int i = 0; i < CH; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) v10_load_agent(part[i][j], &sp[((wave * TI + i0 + i) * TJ + j) * 64 + lane]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                V10_SB();                                       // (the compiler does not know the loads were in flight)
#pragma unroll
                for (int i = 0; i < CH; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[i0 + i][j] += part[i][j];
                V10_SB();
            });
        }
    }

    // ---- exception add-back, behind the K loop: vectors in the ring area (not EARLY: behind the two bucket copies fetched now)
    bool look = false;
    if constexpr (FIX && !EARLY) {
        const int* head = reinterpret_cast<const int*>(side + V10_HEAD);
        cx = a.x_post ? 0 : __builtin_amdgcn_readfirstlane(min(head[0], ROW_BCAP));
        cw = __builtin_amdgcn_readfirstlane(min(head[64], ROW_BCAP));
        nent = cx + cw;
        mode = nent == 0 ? 0 : (nent <= VCAP ? 1 : 3);
        if (mode) {                                             // (uniform over the workgroup)
            const int* bx = row_bucket(xlist, m0);
            const int* bw = row_bucket(wlist, n0);
            const int nx4 = cx ? (EXC_HEADER + EXC_ENTRY * cx) / 4 : 0, nw4 = (EXC_HEADER + EXC_ENTRY * cw) / 4;
            for (int o = tid; o < nx4; o += NT) reinterpret_cast<int4*>(xb)[o] = reinterpret_cast<const int4*>(bx)[o];
            for (int o = tid; o < nw4; o += NT) reinterpret_cast<int4*>(wb)[o] = reinterpret_cast<const int4*>(bw)[o];
            clear_maps();
            __syncthreads();
            bookkeep();
            __syncthreads();
            nlive = __builtin_amdgcn_readfirstlane(flags[17]);
        }
    }
    if (FIX && mode == 1) {
        look = true;
        // the entries whose gathers were requested in front of the loop, then batches of U entries a wave: ONE round trip a batch
        if constexpr (UE > 0) {
#pragma unroll
            for (int u = 0; u < UE; ++u) form(pq[u], wave + NW * u < nlive ? live_idx[wave + NW * u] : nent);
        }
        constexpr int U = 2;
        for (int base = NW * UE; base < nlive; base += NW * U) {    // uniform
            int4 qv[U][VC];
            int idx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                idx[u] = base + wave + NW * u < nlive ? live_idx[base + wave + NW * u] : nent;
                gather(qv[u], idx[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) form(qv[u], idx[u]);
        }
        // rows / columns with several entries (flagged by the bookkeeping): the others added to the first
        // (wave = slot % NW, ascending block) once every vector is complete
        __syncthreads();
        if (flags[16] != 0) {                                   // (uniform)
            auto follower_key = [&](int j) {
                if (j >= nent) return 0x7fffffff;
                const int* f = v8_entry(xb, wb, cx, j);
                const int s3 = f[3];
                return s3 >= 0 && s3 != j && (s3 % NW) == wave ? (s3 << 18) | (f[1] << 8) | j : 0x7fffffff;
            };
            const int k0 = follower_key(lane), k1 = follower_key(lane + 64);
            if (__any(k0 != 0x7fffffff || k1 != 0x7fffffff)) {
                int last = -1;
                for (;;) {
                    int best = k0 > last ? k0 : 0x7fffffff;
                    if (k1 > last) best = min(best, k1);
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) best = min(best, __shfl_xor(best, o));
                    const int key = __builtin_amdgcn_readfirstlane(best);
                    if (key == 0x7fffffff) break;
                    last = key;
                    float* h = vecs + (key >> 18) * VLEN;
                    const float* o = vecs + (key & 255) * VLEN;
#pragma unroll
                    for (int c = 0; c < VLEN / 64; ++c) h[c * 64 + lane] += o[c * 64 + lane];
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: y = float(acc) * sx[m] * sw[n] + bias[n] (+ vectors).  The lane holds, for tile (i, j) of its wave, row
    //      wm * WM + 16 i + l16 and the four columns wn * WN + 16 j + 4 lq + 0..3: one 16-byte store.
    V10_STAMP(5);
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.y) | (uintptr_t)(a.ldy * 4)) & 15) == 0;
    auto store4 = [&](float* yrow, f32x4 val, int col, const float* rrow) {
        if (vec_ok && col + 3 < Ni) {
            if (rrow) val += *reinterpret_cast<const f32x4*>(rrow);
            *reinterpret_cast<f32x4*>(yrow) = val;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (col + r < Ni) yrow[r] = rrow ? val[r] + rrow[r] : val[r];
        }
    };
    unsigned cany = 0;                  // bit j: some column of the wave's column fragment j has a vector
    if (FIX && look) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int4 c = *reinterpret_cast<const int4*>(&colslot[wn * WN + j * 16 + lq * 4]);
            if (__any(c.x >= 0 || c.y >= 0 || c.z >= 0 || c.w >= 0)) cany |= 1u << j;
        }
    }
    {
        f32x4 swr[TJ], bvr[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int cl = wn * WN + j * 16 + lq * 4;
            swr[j] = BF16 ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(&swt[cl]);
            bvr[j] = *reinterpret_cast<const f32x4*>(&bst[cl]);
        }
        int rsv[TI];                    // (the rows' slots and scales up front: one LDS round trip for the wave's fragments, not one each)
        float sxr[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            rsv[i] = FIX && look ? rowslot[wm * WM + i * 16 + l16] : -1;
            sxr[i] = BF16 ? 1.f : sxt[wm * WM + i * 16 + l16];
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (KG == 2 && (i / (TI / 2)) != grp) continue;     // (uniform: the fragment rows this K-group holds complete sums for)
            const int rl = wm * WM + i * 16 + l16;
            const long long row = (long long)m0 + rl;
            const float sxv = sxr[i];
            const int rs = rsv[i];
            const bool rowv = FIX && look && __any(rs >= 0);
            float* yrow = a.y + row * a.ldy + n0 + wn * WN + lq * 4;
            f32x4 val[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) val[j][r] = BF16 ? (float)acc[i][j][r] + bvr[j][r] : (float)acc[i][j][r] * sxv * swr[j][r] + bvr[j][r];
            }
            if (FIX && rowv) {                                   // the row's vector: one product per tile column
                const float* rv = vecs + max(rs, 0) * VLEN + wn * WN + lq * 4;
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(rv + j * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[j][r] += rs >= 0 ? c4[r] : 0.f;
                }
            }
            if (FIX && cany) {                                   // the columns' vectors: one product per tile row
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    if ((cany >> j) & 1) {                       // (uniform)
                        const int4 c = *reinterpret_cast<const int4*>(&colslot[wn * WN + j * 16 + lq * 4]);
                        const int c4[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) val[j][r] += c4[r] >= 0 ? vecs[max(c4[r], 0) * VLEN + rl] : 0.f;
                    }
            }
            if (row < a.M) {
                const float* rrow = a.resid ? a.resid + row * a.ldr + n0 + wn * WN + lq * 4 : nullptr;
#pragma unroll
                for (int j = 0; j < TJ; ++j) store4(yrow + j * 16, val[j], n0 + wn * WN + j * 16 + lq * 4, rrow ? rrow + j * 16 : nullptr);
            }
            V10_SB();
        }
    }
#ifdef V10_STAMPS
    V10_STAMP(6);
    if (a.stamps && lane == 0 && (wave == 0 || wave == NW - 1)) {
        unsigned long long* d = a.stamps + ((long long)blockIdx.x * 2 + (wave != 0)) * 8;
#pragma unroll
        for (int q = 0; q < 7; ++q) d[q] = st_[q];
        d[7] = ((unsigned long long)(unsigned)nent << 32) | ((unsigned)nlive << 8) | (unsigned)mode;
    }
#endif
    if (FIX && mode == 3) {
        // more entries than the ring holds vectors for: the products added to the tile with atomics after its stores
        // (order not fixed: the one route of this kernel that is not reproducible to the last bit)
        __threadfence();
        __syncthreads();
        for (int it = wave; it < nent * 4; it += NW) {          // (uniform: an entry's quarter of 64 rows / columns)
            const int i = it >> 2, c = it & 3;
            const bool is_x = i < cx;
            const int* e = v8_entry(xb, wb, cx, i);
            const int rl = c * 64 + lane;
            if (e[3] == -2 || rl >= (is_x ? BN : BM)) continue;
            const long long q = (is_x ? n0 : m0) + rl, r = e[0];
            if (q >= (is_x ? Ni : Mi)) continue;
            const int4 qv = *reinterpret_cast<const int4*>((is_x ? +a.wm : +a.xm) + tiled_offset(q, (long long)e[1] * 16, a.K));
            const int d = dot16(*reinterpret_cast<const int4*>(e + 4), qv);
            if (d != 0)
                atomicAdd(&a.y[(is_x ? r : q) * a.ldy + (is_x ? q : r)],
                          __builtin_ldexpf((float)d, e[2] - (is_x ? x_off_s : w_off_s)) * (is_x ? swt : sxt)[rl]);
        }
        for (int idx = tid; idx < cx * cw; idx += NT) {
            const int* e = xb + EXC_HEADER + EXC_ENTRY * (idx / cw);
            const int* f = wb + EXC_HEADER + EXC_ENTRY * (idx % cw);
            if (e[3] == -2 || f[3] == -2 || e[1] != f[1]) continue;
            const int d = dot16(*reinterpret_cast<const int4*>(e + 4), *reinterpret_cast<const int4*>(f + 4));
            if (d != 0) atomicAdd(&a.y[(long long)e[0] * a.ldy + f[0]], __builtin_ldexpf((float)d, e[2] + f[2] - a.scale_bias));
        }
    }
}

// geometry: 1 = 128 x 256 (1 x 4 waves of 128 x 64), 2 = 256 x 128 (2 x 2 of 128 x 64), 3 = 128 x 128 (2 x 2 of 64 x 64),
// 4 = 128 x 64 (2 x 2 of 64 x 32: grids that 128 x 128 tiles leave on half the compute units or fewer)
// 5 / 6 (round 6): 128 x 128 / 128 x 256 with TWO K-GROUPS in an 8-wave workgroup -- grids of at most one workgroup a compute unit
void v10_tile_shape(int geom, int& bm, int& bn) {
    bm = geom == 2 ? 256 : 128;
    bn = (geom == 1 || geom == 6) ? 256 : (geom == 4 ? 64 : 128);
}

template <int NWM, int NWN, int TI, int NS, int OCC, int TJ = 4, int KG = 1>
static int v10_launch(const GemmArgs& a, const float* sx, const float* sw, const int* xlist, const int* wlist, hipStream_t st,
                      const uint8_t* xf, const uint8_t* wf, bool bf16, unsigned grid) {
    const bool fix = xlist && wlist;
    constexpr int RING = KG * NS * (NWM * TI + NWN * TJ) * 1024;          // (KG = 2: one ring a K-group)
    // (dynamic shared memory beyond 64 KiB has to be asked for once per kernel)
    static const bool ready = [] {
        bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&bfp_gemm_v10<NWM, NWN, TI, TJ, NS, OCC, 0, true, KG>), hipFuncAttributeMaxDynamicSharedMemorySize, RING) == hipSuccess;
        ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&bfp_gemm_v10<NWM, NWN, TI, TJ, NS, OCC, 1, false, KG>), hipFuncAttributeMaxDynamicSharedMemorySize, RING) == hipSuccess && ok;
        ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&bfp_gemm_v10<NWM, NWN, TI, TJ, NS, OCC, 0, false, KG>), hipFuncAttributeMaxDynamicSharedMemorySize, RING) == hipSuccess && ok;
        return ok;
    }();
    if (!ready) return (int)hipErrorInvalidValue;
    if (bf16) hipLaunchKernelGGL((bfp_gemm_v10<NWM, NWN, TI, TJ, NS, OCC, 0, true, KG>), grid, 256 * KG, RING, st, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    else if (fix) hipLaunchKernelGGL((bfp_gemm_v10<NWM, NWN, TI, TJ, NS, OCC, 1, false, KG>), grid, 256 * KG, RING, st, a, sx, sw, xlist, wlist, xf, wf);
    else hipLaunchKernelGGL((bfp_gemm_v10<NWM, NWN, TI, TJ, NS, OCC, 0, false, KG>), grid, 256 * KG, RING, st, a, sx, sw, xlist, wlist, xf, wf);
    return (int)hipGetLastError();
}

// a.splits / a.slabs / a.tickets set by the caller (slabs of BM x BN x 4 bytes); K % 64 == 0
static unsigned long long* g_v10_stamps = nullptr;      // diagnostic (-DV10_STAMPS builds, tools/dbg/v10_stamps.py)

int launch_bfp_gemm_v10(const GemmArgs& a_in, const float* sx, const float* sw, const int* xlist, const int* wlist, hipStream_t st,
                        const uint8_t* xf, const uint8_t* wf, bool bf16, int geom) {
    GemmArgs a = a_in;
    a.stamps = g_v10_stamps;
    const bool fix = xlist && wlist;
    if (fix && (!xf || !wf)) return MI355Q_E_BADARG;
    int bm, bn;
    v10_tile_shape(geom, bm, bn);
    const unsigned tiles = (unsigned)(((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn));
    const unsigned grid = tiles * (a.ngroup > 1 ? a.ngroup : 1) * (a.splits > 1 ? a.splits : 1);
    // ring depth: NS - 2 K-steps of LDS-DMA stay in flight across a barrier.  Two (three) workgroups a compute unit cover each
    // other's waits with a shallow ring; a workgroup that has its compute unit to itself (grids of <= 256 tiles: the shard and
    // projection shapes this kernel is for) needs the flight time of an L2 round trip under load in K-steps of 256-512 clocks:
    // profiles/r05_small_tiles.txt.  MI355Q_V10_NS pins a depth for sweeps.
    const int ns_env = getenv("MI355Q_V10_NS") ? atoi(getenv("MI355Q_V10_NS")) : 0;
    const bool deep = ns_env ? ns_env > 4 : grid <= 256;       // (one workgroup a compute unit at most)
    const int ns = ns_env ? ns_env : 0;
#ifndef V10_ONLY_G3N4
    if (geom == 1) {
        if (ns == 6) return v10_launch<1, 4, 8, 6, 1>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid);
        return deep ? v10_launch<1, 4, 8, 4, 1>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid)
                    : v10_launch<1, 4, 8, 3, 2>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid);
    }
    if (geom == 2) {
        return deep ? v10_launch<2, 2, 8, 4, 1>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid)
                    : v10_launch<2, 2, 8, 3, 2>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid);
    }
    if (geom == 3) {
        if (ns == 8) return v10_launch<2, 2, 4, 8, 1>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid);
        return deep ? v10_launch<2, 2, 4, 6, 1>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid)
                    : v10_launch<2, 2, 4, 4, 2>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid);
    }
    if (geom == 4) {
        return deep ? v10_launch<2, 2, 4, 6, 1, 2>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid)
                    : v10_launch<2, 2, 4, 4, 2, 2>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid);
    }
    if (geom == 5 || geom == 6) {
        if (a.splits > 1 || (a.K % 128) != 0) return MI355Q_E_UNSUPPORTED;
        return geom == 5 ? v10_launch<2, 2, 4, 4, 1, 4, 2>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid)
                         : v10_launch<1, 4, 8, 3, 1, 4, 2>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid);
    }
#else
    (void)deep; (void)ns;
    if (geom == 3) return v10_launch<2, 2, 4, 4, 2>(a, sx, sw, xlist, wlist, st, xf, wf, bf16, grid);
#endif
    return MI355Q_E_BADARG;
}

}  // namespace mi355q

// diagnostic hook, not part of include/mi355q.h: the buffer ([workgroups][2][8] 64-bit words) a -DV10_STAMPS build fills
extern "C" __attribute__((visibility("default"))) void mi355q_debug_v10_stamps(void* buf) { mi355q::g_v10_stamps = static_cast<unsigned long long*>(buf); }
