// mi355q_quant.hip -- block quantisers for gfx950 (MI355X), HBM-bound streaming kernels.
//
// One pass over x produces, per block of b0 x b1 values, the shared exponent / bias and,
// per element, the fake-quantised fp32 value and (block_fp) the packed int8 mantissa.
//   block_fp        : reference quantizers/block_fp.py:21-96
//   block_minifloat : reference quantizers/block_minifloat.py:22-74 + minifloat.py:134-196
//   block_log       : reference quantizers/block_log.py:23-69 + log.py:22-56
// Bit-exactness notes
//   * every fp32 operation of the reference is done as the same IEEE fp32 operation here
//     (scalings by 2^n through v_ldexp_f32, which rounds like the division/multiplication it
//     replaces, subnormals included);
//   * ceil/floor/rint of log2 are taken from integer threshold tables on the fp32 fraction
//     (log2_tables.inc, tools/gen_log2_tables.py) -- no logarithm is evaluated on the device;
//   * all-zero blocks take the reference's tensor-global fill (block_fp.py:54-58) in a second
//     launch that returns immediately when kernel 1 met no such block.
//
// Data layout in HBM: x/y/mant are the caller's contiguous row-major tensors; one wave reads
// 1 KiB of x per instruction (float4 per lane), a block of 16 values is held by 4 adjacent
// lanes and its abs-max is formed by two DPP quad permutes -- no LDS traffic on the hot loop
// except the threshold lookup for the block exponent.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <utility>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_align_row.h"

#include "mi355q_quant_dev.h"

namespace mi355q {

// Zero-block state of kernel 1 (exact mode only) for the fix-up launch that follows on the same stream: each workgroup
// folds the smallest non-zero block max it met -- as max(~bits) -- into workspace slot blockIdx % WS_SLOTS with one
// result-less atomic max (2048 slots: a handful of workgroups per slot over the kernel's lifetime; atomics of thousands of
// waves on ONE word serialised at the L2 and cost 6-10 us per call), and a workgroup that met an all-zero block raises the
// flag word (a plain store of 1: every writer writes the same value).  The slots are zero between calls: the fix-up
// launch clears them (zero_fixup_kernel).
// The fill kernel 1 writes all-zero blocks with.  The reference's is the smallest non-zero block maximum of the WHOLE tensor
// (block_fp.py:54-58), known only when kernel 1 is through: kernel 1 writes them with the fill the fix-up launch of the LAST
// tensor with all-zero blocks found (a word of the workspace; 1.0 before there was one, and always in the fast mode), the
// fix-up launch compares the block parameters that fill gives with the true ones and rewrites the blocks only if they
// differ.  Attention probabilities under a causal mask -- half of the blocks all zero, the same tiny fill layer after layer
// -- then cost one pass instead of two (r03: block_log on [32, 2048, 2048] 355 -> 2xx us); any other sequence of tensors
// costs what it did.  Exact either way.
__device__ __forceinline__ float zero_fill_guess(const unsigned* ws, bool exact) {
    const unsigned g = exact ? ws[WS_FILL_GUESS] : 0u;
    return g ? __uint_as_float(g) : 1.0f;
}

__device__ __forceinline__ void publish_zero_state(unsigned* ws, bool saw_zero, unsigned inv) {
    __shared__ unsigned wg_inv[4];
    __shared__ int wg_zero[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)inv, o);
        inv = other > inv ? other : inv;
    }
    const bool any_zero = __any(saw_zero);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { wg_inv[wave] = inv; wg_zero[wave] = any_zero ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = wg_inv[0];
        int z = wg_zero[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { m = wg_inv[w] > m ? wg_inv[w] : m; z |= wg_zero[w]; }
        if (m) (void)__hip_atomic_fetch_max(&ws[WS_SLOT0 + (blockIdx.x & (WS_SLOTS - 1))], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (z) ws[WS_ZERO_FLAG] = 1u;
    }
}

// ---------------------------------------------------------------------------------------
// kernel 1, vector path: b0 == 1, cols % b1 == 0, b1 = 4 * LPB, 16-byte aligned x / y.
// Flat over all elements: float4 slot i belongs to block i / LPB.
// ---------------------------------------------------------------------------------------
// PIECES > 0 (tensors beyond the memory-side cache: 256 MiB and more): a workgroup owns PIECES consecutive 4-KiB pieces of x
// (256 float4 each) -- no grid-stride loop: with one piece per workgroup a 512-MiB copy runs at 6.2 TB/s on this memory
// system, as a grid-stride loop of 2048 workgroups at 4.9, a pure write stream at 6.8 vs 4.35 (tools/ubench/stream.hip,
// profiles/r03_stream_limits.txt).  PIECES == 0: the grid-stride loop, for everything smaller.
template <int FMT, int LPB, int PIECES>
__global__ __launch_bounds__(256) void quant_vec_kernel(const QuantArgs a) {
    __shared__ Lut lut;
    const long long n4 = a.n_elems >> 2;
    const long long n4_pad = (n4 + 63) & ~63ll;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x);
    float4* __restrict__ y4 = reinterpret_cast<float4*>(a.y);
    unsigned* __restrict__ m4 = reinterpret_cast<unsigned*>(a.mant);
    const bool exact = (a.flags & MI355Q_ZERO_BLOCK_FAST) == 0u;
    bool saw_zero = false;
    unsigned inv = 0u;          // max over non-zero blocks of ~bits(block max) = the smallest non-zero block max
    const float guess = zero_fill_guess(a.ws, exact);
    const int mbits = FMT == FMT_BM ? (int)__builtin_log2f(a.shift) : 0;

    // (tried: block_log staging only the ceil table and reading its two rounding tables from memory on the rare table walk -- one
    //  wave-step in ~260 takes that walk, and eight dependent L2 round trips there cost more than the staging saves: 16.7 -> 18.1 us)
    auto stage_lut = [&]() { load_lut<FMT>(lut); };
    auto process = [&](const long long i, const float4 v) {
        const bool valid = i < n4;
        float bmax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        bmax = group_max<LPB>(bmax);
        if (LPB == 4 && a.zmap) {   // (uniform) one ballot per wave iteration: which of its sixteen blocks are all zero
            const unsigned long long zb = __ballot(bmax == 0.f && valid);
            if ((threadIdx.x & 63) == 0) a.zmap[i >> 6] = zb;
            // (the fix-up pass finds the all-zero blocks through the map, without reading x again, should the fill they
            //  are written with below turn out not to be the tensor's)
        }
        if (bmax == 0.f) {          // all-zero block: provisional fill -- the last tensor's (exact mode), else 1.0
            saw_zero = saw_zero || valid;
            bmax = guess;
        } else if (exact) {
            const unsigned vb = ~__float_as_uint(bmax);
            inv = vb > inv ? vb : inv;
        }
        unsigned code;
        const BlockParam bp = block_param<FMT>(bmax, a, lut, code);
        int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        float4 o;
        if constexpr (FMT == FMT_BFP) {
            o.x = quant_elem<FMT>(v.x, bp, a, lut, q0);
            o.y = quant_elem<FMT>(v.y, bp, a, lut, q1);
            o.z = quant_elem<FMT>(v.z, bp, a, lut, q2);
            o.w = quant_elem<FMT>(v.w, bp, a, lut, q3);
        } else {
            // minifloat / log elements by the short forms (14-18 operations, no table: mi355q_quant_dev.h); a wave with an element
            // inside one of the bands where torch's fp32 log2 decides (a few ulps under a power of two / the 125 fractions around
            // sqrt(2); subnormals) redoes its four with the table walk, one after the other (about one wave-step in a thousand;
            // kept out of each other's way: interleaved they cost the kernel 84 registers and three waves a SIMD)
            bool near = false;
            const int eminb = 127 - bp.p, emaxb = 127 + a.span - bp.p;
            if constexpr (FMT == FMT_BM) {
                o.x = bm_elem_fused(v.x, eminb, emaxb, mbits, a.shift, a.mant_max, near);
                o.y = bm_elem_fused(v.y, eminb, emaxb, mbits, a.shift, a.mant_max, near);
                o.z = bm_elem_fused(v.z, eminb, emaxb, mbits, a.shift, a.mant_max, near);
                o.w = bm_elem_fused(v.w, eminb, emaxb, mbits, a.shift, a.mant_max, near);
                // (the reference's mask arithmetic turns a passed-through -0.0 into +0.0; a value flushed to zero keeps its sign)
                o.x = fabsf(v.x) <= ATOL ? v.x + 0.0f : o.x;
                o.y = fabsf(v.y) <= ATOL ? v.y + 0.0f : o.y;
                o.z = fabsf(v.z) <= ATOL ? v.z + 0.0f : o.z;
                o.w = fabsf(v.w) <= ATOL ? v.w + 0.0f : o.w;
            } else {
                o.x = bl_elem_fused(v.x, bp.eps, eminb, emaxb, near);
                o.y = bl_elem_fused(v.y, bp.eps, eminb, emaxb, near);
                o.z = bl_elem_fused(v.z, bp.eps, eminb, emaxb, near);
                o.w = bl_elem_fused(v.w, bp.eps, eminb, emaxb, near);
            }
            if (__any(near)) {
                o.x = quant_elem<FMT>(v.x, bp, a, lut, q0);
                __builtin_amdgcn_sched_barrier(0);
                o.y = quant_elem<FMT>(v.y, bp, a, lut, q1);
                __builtin_amdgcn_sched_barrier(0);
                o.z = quant_elem<FMT>(v.z, bp, a, lut, q2);
                __builtin_amdgcn_sched_barrier(0);
                o.w = quant_elem<FMT>(v.w, bp, a, lut, q3);
            }
        }
        if (valid) {
            if (a.y) y4[i] = o;
            if (a.ybf)          // (block_fp of <= 9 bits, minifloats of <= 7 mantissa bits, signed powers of two: exact in bf16)
                reinterpret_cast<uint2*>(a.ybf)[i] = make_uint2(pack_bf16(o.x, o.y), pack_bf16(o.z, o.w));
            if (FMT == FMT_BFP && a.mant)
                m4[i] = (unsigned)(q0 & 0xFF) | ((unsigned)(q1 & 0xFF) << 8) | ((unsigned)(q2 & 0xFF) << 16) |
                        ((unsigned)(q3 & 0xFF) << 24);
            if (a.code && (i & (LPB - 1)) == 0) a.code[i / LPB] = (uint8_t)code;
        }
    };
    if constexpr (PIECES == 0) {
        // tensors that stay in the 256-MiB memory-side cache between calls (activations, weights of one layer): the
        // grid-stride loop of <= 2048 workgroups -- tables and zero-state once per workgroup, which is what these
        // launch-bound sizes feel
        stage_lut();
        const long long stride = (long long)gridDim.x * blockDim.x;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4_pad; i += stride) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n4) v = x4[i];
            process(i, v);
        }
    } else {
        constexpr int P = PIECES > 0 ? PIECES : 1;
        const long long i = (long long)blockIdx.x * (256 * P) + threadIdx.x;
        float4 v[P];
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const long long iu = i + u * 256;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iu < n4) v[u] = x4[iu];
        }
        stage_lut();         // (behind the loads of x: the tables' round trip -- L2 hits -- hides under theirs)
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const long long iu = i + u * 256;           // (a wave's 64 slots are in or out of the padded range together)
            if (iu < n4_pad) process(iu, v[u]);
        }
    }
    if (exact) publish_zero_state(a.ws, saw_zero, inv);
}

// ---------------------------------------------------------------------------------------
// generic block walker: a 16-lane group owns one b0 x b1 block at a time (any layout).
// ---------------------------------------------------------------------------------------
struct BlockCursor {
    long long base;   // element offset of the block's (0,0)
    int h, w;         // valid rows / cols of this (possibly ragged) block
};
__device__ __forceinline__ BlockCursor locate(const QuantArgs& a, long long bid) {
    const long long per_plane = a.nbr * a.nbc;
    const long long l = bid / per_plane, r = bid - l * per_plane;
    const long long br = r / a.nbc, bc = r - br * a.nbc;
    BlockCursor c;
    const long long r0 = br * a.b0, c0 = bc * a.b1;
    c.base = (l * a.rows + r0) * a.cols + c0;
    c.h = (int)((a.rows - r0) < a.b0 ? (a.rows - r0) : a.b0);
    c.w = (int)((a.cols - c0) < a.b1 ? (a.cols - c0) : a.b1);
    return c;
}
__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ float block_absmax16(const QuantArgs& a, const BlockCursor& c, int lane16) {
    float m = 0.f;
    const int n = c.h * c.w;
    for (int i = lane16; i < n; i += 16) {
        const int ii = i / c.w, jj = i - ii * c.w;
        m = fmaxf(m, fabsf(a.x[c.base + (long long)ii * a.cols + jj]));
    }
    return group16_max(m);
}

template <int FMT>
__global__ __launch_bounds__(256) void quant_generic_kernel(const QuantArgs a) {
    __shared__ Lut lut;
    load_lut<FMT>(lut);
    const int lane16 = threadIdx.x & 15;
    const long long groups = ((long long)gridDim.x * blockDim.x) >> 4;
    const bool exact = (a.flags & MI355Q_ZERO_BLOCK_FAST) == 0u;
    bool saw_zero = false;
    unsigned inv = 0u;
    const float guess = zero_fill_guess(a.ws, exact);
    // every lane of a group runs the same trip count (bid is group-uniform)
    for (long long bid = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; bid < a.n_blocks; bid += groups) {
        const BlockCursor c = locate(a, bid);
        float bmax = block_absmax16(a, c, lane16);
        if (bmax == 0.f) { saw_zero = true; bmax = guess; }
        else if (exact) { const unsigned v = ~__float_as_uint(bmax); inv = v > inv ? v : inv; }
        unsigned code;
        const BlockParam bp = block_param<FMT>(bmax, a, lut, code);
        const int n = c.h * c.w;
        for (int i = lane16; i < n; i += 16) {
            const int ii = i / c.w, jj = i - ii * c.w;
            const long long at = c.base + (long long)ii * a.cols + jj;
            int q;
            const float o = quant_elem<FMT>(a.x[at], bp, a, lut, q);
            if (a.y) a.y[at] = o;
            if (FMT == FMT_BFP && a.mant) a.mant[at] = (int8_t)q;
        }
        if (a.code && lane16 == 0) a.code[bid] = (uint8_t)code;
    }
    if (exact) publish_zero_state(a.ws, saw_zero, inv);
}

// ---------------------------------------------------------------------------------------
// kernel 2: all-zero blocks take the tensor-global fill (block_fp.py:54-58).  Kernel 1 has left, in the
// workspace, whether it met an all-zero block and the smallest non-zero block max of the whole tensor
// (publish_zero_state: one slot per workgroup of kernel 1); the stream orders the two launches, so there is no in-kernel
// grid barrier.  Returns at once when kernel 1 met no zero block.  The last workgroup out lowers the flag.
// ---------------------------------------------------------------------------------------
constexpr int FIXUP_GRID = 2048;     // (exits at once when kernel 1 met no all-zero block)
constexpr int FIXUP_GRID_MAP = 2048; // with the zero-block map (tensors >= 128 MiB): 16 KiB of output per workgroup and trip

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int FMT>
__global__ __launch_bounds__(256) void zero_fixup_kernel(const QuantArgs a, int n_slots) {
    __shared__ Lut lut;
    __shared__ unsigned wg_inv[4];
    if (ld_agent(&a.ws[WS_ZERO_FLAG]) == 0u) {           // uniform over the grid: nothing to rewrite
        // (nobody of this launch reads the slots then: one workgroup clears them for the next call's atomic maxima)
        if (blockIdx.x == 0) for (int i = threadIdx.x; i < n_slots; i += blockDim.x) a.ws[WS_SLOT0 + i] = 0u;
        return;
    }
    load_lut<FMT>(lut);
    const int lane16 = threadIdx.x & 15;
    const long long groups = ((long long)gridDim.x * blockDim.x) >> 4;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    // the tensor's smallest non-zero block max: every workgroup reduces kernel 1's slots for itself
    unsigned inv_all = 0u;
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
        const unsigned v = a.ws[WS_SLOT0 + i];
        inv_all = v > inv_all ? v : inv_all;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)inv_all, o);
        inv_all = other > inv_all ? other : inv_all;
    }
    if ((threadIdx.x & 63) == 0) wg_inv[threadIdx.x >> 6] = inv_all;
    __syncthreads();
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) inv_all = wg_inv[w] > inv_all ? wg_inv[w] : inv_all;
    const float fill = inv_all == 0u ? 1.0f : __uint_as_float(~inv_all);   // all blocks zero -> 1
    unsigned code;
    const BlockParam bp = block_param<FMT>(fill, a, lut, code);
    // what kernel 1 wrote the all-zero blocks with: the same block parameters -> the same values, nothing to rewrite
    unsigned gcode;
    const BlockParam gbp = block_param<FMT>(zero_fill_guess(a.ws, true), a, lut, gcode);
    const bool hit = gcode == code && gbp.p == bp.p && __float_as_uint(gbp.eps) == __float_as_uint(bp.eps);
    if (!hit) {

    // Row-vector blocks of 16 (every shipped configuration): one lane per block, four 16-byte loads, and -- every element
    // of an all-zero block being (+/-)0 -- one value for all of them.  A causal attention-probability tensor has half
    // of its blocks here: this pass is then a second stream over the tensor, not a scalar walk (2.5 ms -> 0.2 ms at
    // [32, 2048, 2048]).
    if (a.b0 == 1 && a.b1 == 16 && (a.cols & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.mant)) & 15) == 0) {
        int q0;
        const float zv = quant_elem<FMT>(0.0f, bp, a, lut, q0);
        const float4 z4 = make_float4(zv, zv, zv, zv);
        if (a.zmap) {
            // kernel 1 left a map of the all-zero blocks: x is not read again (a causal probability tensor: 0.46 -> 0.2x ms)
            const long long nthreads = (long long)gridDim.x * blockDim.x;
            const int m4 = (q0 & 255) * 0x01010101;
            // one lane per float4 slot, as kernel 1 walked them (a wave and its map word cover the same sixteen blocks):
            // the stores of a wave are 1 KiB of consecutive bytes
            const long long n4 = a.n_elems >> 2;
            float4* __restrict__ y4 = reinterpret_cast<float4*>(a.y);
            unsigned* __restrict__ mant4 = reinterpret_cast<unsigned*>(a.mant);
            // the non-empty words are broadcast from the lanes that loaded them -- no load in front of any store
            (void)nthreads;
            const long long words = (n4 + 63) >> 6;
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            // a workgroup takes 16 consecutive map words per trip (16 KiB of output, 4 words per wave: lanes 0-3 load them)
            for (long long base = (long long)blockIdx.x * 16 + wave * 4; base < words; base += (long long)gridDim.x * 16) {
                const unsigned long long mine = lane < 4 && base + lane < words ? a.zmap[base + lane] : 0ull;
                unsigned long long todo = __ballot(mine != 0ull);
                while (todo) {
                    const int j = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, j);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), j);
                    const unsigned long long w = ((unsigned long long)hi << 32) | lo;
                    const long long i = (base + j) * 64 + lane;
                    if (((w >> (lane & 60)) & 1ull) == 0ull || i >= n4) continue;
                    if (a.y) y4[i] = z4;
                    if (a.ybf) reinterpret_cast<uint2*>(a.ybf)[i] = make_uint2(pack_bf16(zv, zv), pack_bf16(zv, zv));
                    if (FMT == FMT_BFP && a.mant) mant4[i] = (unsigned)m4;
                    if (a.code && (i & 3) == 0) a.code[i >> 2] = (uint8_t)code;
                }
            }
        } else {
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x);
        const long long nthreads = (long long)gridDim.x * blockDim.x;
        for (long long bid = (long long)blockIdx.x * blockDim.x + threadIdx.x; bid < a.n_blocks; bid += nthreads) {
            const float4 v0 = x4[bid * 4], v1 = x4[bid * 4 + 1], v2 = x4[bid * 4 + 2], v3 = x4[bid * 4 + 3];
            const bool nz = v0.x != 0.f || v0.y != 0.f || v0.z != 0.f || v0.w != 0.f || v1.x != 0.f || v1.y != 0.f ||
                            v1.z != 0.f || v1.w != 0.f || v2.x != 0.f || v2.y != 0.f || v2.z != 0.f || v2.w != 0.f ||
                            v3.x != 0.f || v3.y != 0.f || v3.z != 0.f || v3.w != 0.f;
            if (nz) continue;
            if (a.y) {
                float4* __restrict__ y4 = reinterpret_cast<float4*>(a.y) + bid * 4;
                y4[0] = z4; y4[1] = z4; y4[2] = z4; y4[3] = z4;
            }
            if (a.ybf) {
                const unsigned zz = pack_bf16(zv, zv);
                uint4* __restrict__ yb = reinterpret_cast<uint4*>(a.ybf + bid * 16);
                yb[0] = make_uint4(zz, zz, zz, zz); yb[1] = make_uint4(zz, zz, zz, zz);
            }
            // (the mantissa of a zero element under the global fill is round(1e-9 * 2^(mbits - e)): not zero when the
            //  fill's exponent is tiny)
            if (FMT == FMT_BFP && a.mant) {
                const int m4 = (q0 & 255) * 0x01010101;
                *reinterpret_cast<int4*>(a.mant + bid * 16) = make_int4(m4, m4, m4, m4);
            }
            if (a.code) a.code[bid] = (uint8_t)code;
        }
        }
    } else
    for (long long bid = g0; bid < a.n_blocks; bid += groups) {
        const BlockCursor c = locate(a, bid);
        if (block_absmax16(a, c, lane16) != 0.f) continue;
        const int n = c.h * c.w;
        for (int i = lane16; i < n; i += 16) {
            const int ii = i / c.w, jj = i - ii * c.w;
            const long long at = c.base + (long long)ii * a.cols + jj;
            int q;
            const float o = quant_elem<FMT>(a.x[at], bp, a, lut, q);
            if (a.y) a.y[at] = o;
            if (FMT == FMT_BFP && a.mant) a.mant[at] = (int8_t)q;
        }
        if (a.code && lane16 == 0) a.code[bid] = (uint8_t)code;
    }
    }
    // exit ticket: the last workgroup out lowers the flag, clears the slots and leaves this tensor's fill for the next call
    // (every workgroup has read them before it takes its ticket)
    __shared__ int last_out;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned c = blockIdx.x % WS_TICKET_L1_N;
        const unsigned mates = (gridDim.x + WS_TICKET_L1_N - 1u - c) / WS_TICKET_L1_N;          // workgroups on counter c
        unsigned* l1 = &a.ws[WS_TICKET_L1 + c * WS_TICKET_L1_STRIDE];
        last_out = 0;
        if (__hip_atomic_fetch_add(l1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mates - 1u) {
            __hip_atomic_store(l1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned groups = gridDim.x < (unsigned)WS_TICKET_L1_N ? gridDim.x : (unsigned)WS_TICKET_L1_N;
            last_out = __hip_atomic_fetch_add(&a.ws[WS_TICKET], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u;
        }
    }
    __syncthreads();
    if (last_out) {                                       // (every workgroup has read the slots before its ticket)
        for (int i = threadIdx.x; i < n_slots; i += blockDim.x) a.ws[WS_SLOT0 + i] = 0u;
        if (threadIdx.x == 0) {
            a.ws[WS_FILL_GUESS] = __float_as_uint(fill);
            __hip_atomic_store(&a.ws[WS_TICKET], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.ws[WS_ZERO_FLAG], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------
// The un-blocked siblings (minifloat.py:21-196, log.py:22-56): every element by itself with ONE fixed bias -- the
// element functions of block_minifloat / block_log with the bias a parameter, plus the denormal minifloat (no implicit
// leading one: exponent ceil(log2(|x| + 1e-9)) per element, mantissa |x| / 2^e without the 1e-9).  Flat float4 stream.
// ---------------------------------------------------------------------------------------
constexpr int FMT_DN = 3;
__device__ __forceinline__ float quant_denorm(float x, const QuantArgs& a, const Lut& lut) {
    const float ax = fabsf(x);
    const float s = sgn(x + EPS9);
    const int e = clampi(ceil_log2(ax + EPS9, lut), a.e_min, a.e_max);
    const float m = clampf(__builtin_rintf(__builtin_ldexpf(ax, -e) * a.shift), 0.f, a.mant_max);
    const float q = __builtin_ldexpf(s, e) * (m * a.inv_shift);
    return ax <= ATOL ? x + 0.0f : q;       // (+ 0: the reference's mask arithmetic turns -0.0 into +0.0)
}

template <int FMT>
__global__ __launch_bounds__(256) void quant_flat_kernel(const QuantArgs a, int bias) {
    __shared__ Lut lut;
    load_lut<FMT == FMT_DN ? FMT_BFP : FMT>(lut);
    BlockParam bp;
    bp.p = bias;
    bp.eps = FMT == FMT_BL ? (float)(__builtin_ldexp(1.0, -bias) * 0.1) : 0.f;     // (python float arithmetic: log.py:45-48)
    const long long n4 = a.n_elems >> 2, stride = (long long)gridDim.x * blockDim.x;
    auto one = [&](float v) {
        int mant;
        return FMT == FMT_DN ? quant_denorm(v, a, lut) : quant_elem<FMT>(v, bp, a, lut, mant);
    };
    if ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y)) % 16 == 0) {
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x);
        float4* __restrict__ y4 = reinterpret_cast<float4*>(a.y);
        // (the rare-lookup shortcuts of floor_log2 / rint_log2 vote over the wave: every lane takes every trip)
        const long long n4_pad = (n4 + 63) & ~63ll;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4_pad; i += stride) {
            float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
            if (i < n4) v = x4[i];
            const float4 o = make_float4(one(v.x), one(v.y), one(v.z), one(v.w));
            if (i < n4) y4[i] = o;
        }
        const long long tail0 = n4 << 2, ntail = a.n_elems - tail0;
        if (blockIdx.x == 0 && threadIdx.x < 64) {
            const float v = (long long)threadIdx.x < ntail ? a.x[tail0 + threadIdx.x] : 1.f;
            const float o = one(v);
            if ((long long)threadIdx.x < ntail) a.y[tail0 + threadIdx.x] = o;
        }
    } else {
        const long long n_pad = (a.n_elems + 63) & ~63ll;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += stride) {
            const float v = i < a.n_elems ? a.x[i] : 1.f;
            const float o = one(v);
            if (i < a.n_elems) a.y[i] = o;
        }
    }
}

static int grid_for(long long work_items, int per_block);
int launch_quant_flat(const QuantArgs& a, int fmt, int bias, hipStream_t st) {
    const int grid = grid_for((a.n_elems >> 2) + 64, 256);
    if (fmt == FMT_BM) hipLaunchKernelGGL((quant_flat_kernel<FMT_BM>), grid, 256, 0, st, a, bias);
    else if (fmt == FMT_BL) hipLaunchKernelGGL((quant_flat_kernel<FMT_BL>), grid, 256, 0, st, a, bias);
    else hipLaunchKernelGGL((quant_flat_kernel<FMT_DN>), grid, 256, 0, st, a, bias);
    return (int)hipGetLastError();
}

// integer fixed point (integer.py:49-52): clamp(rint(x * 2^f), lo, hi) / 2^f
__global__ __launch_bounds__(256) void integer_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      long long n, float scale, float lo, float hi) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = clampf(__builtin_rintf(x[i] * scale), lo, hi) / scale;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static int grid_for(long long work_items, int per_block) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > WS_SLOTS) g = WS_SLOTS;   // 256 CUs x 8 blocks, grid-stride beyond (guide: Guideline 11); one workspace slot each
    return (int)g;
}

// the zero-block map of the exact mode: library-owned, one per (device, stream), grow-only (like the split-K workspace of
// mi355q_gemm_v8.hip; a stream's launches are ordered, so one map per stream is enough)
static unsigned long long* zmap_workspace(hipStream_t st, size_t bytes) {
    struct Buf { void* p = nullptr; size_t n = 0; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Buf> all;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Buf& b = all[{dev, st}];
    if (b.n < bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;   // (no allocation under capture)
        if (b.p) (void)hipFree(b.p);                      // (synchronises: nothing of this map is in flight after)
        b.p = nullptr;
        b.n = 0;
        const size_t want = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        if (hipMalloc(&b.p, want) != hipSuccess) return nullptr;
        b.n = want;
    }
    return static_cast<unsigned long long*>(b.p);
}

template <int FMT>
static int launch_format(const QuantArgs& a_in, bool needs_fixup, hipStream_t st) {
    // kernel 1 leaves the zero-block state in the workspace only when the fix-up launch follows (and clears it)
    QuantArgs a = a_in;
    if (!needs_fixup) a.flags |= MI355Q_ZERO_BLOCK_FAST;
    const bool vec_ok = a.b0 == 1 && (a.cols % a.b1) == 0 && (a.b1 % 4) == 0 &&
                        (reinterpret_cast<uintptr_t>(a.x) % 16) == 0 &&
                        (a.y == nullptr || reinterpret_cast<uintptr_t>(a.y) % 16 == 0) &&
                        (a.ybf == nullptr || reinterpret_cast<uintptr_t>(a.ybf) % 8 == 0) &&
                        (a.mant == nullptr || reinterpret_cast<uintptr_t>(a.mant) % 4 == 0);
    const int lpb = a.b1 / 4;
    int grid;
    a.zmap = nullptr;
    if (vec_ok && lpb == 4 && needs_fixup && (a.flags & MI355Q_ZERO_BLOCK_FAST) == 0u && (a.cols & 15) == 0 &&
        (a.y == nullptr || reinterpret_cast<uintptr_t>(a.y) % 16 == 0) &&
        (a.mant == nullptr || reinterpret_cast<uintptr_t>(a.mant) % 16 == 0) && a.n_elems >= (1 << 25))
        // (tensors of 128 MiB and more -- the attention-probability class, where half of the blocks can be zero; below that
        //  the map's bookkeeping on the host costs a launch-bound call more than a second read of x would)
        a.zmap = zmap_workspace(st, (size_t)((((a.n_elems >> 2) + 63) & ~63ll) >> 6) * 8);      // (null: the pass reads x again)
    if (vec_ok && (lpb == 1 || lpb == 2 || lpb == 4 || lpb == 8 || lpb == 16 || lpb == 32 || lpb == 64)) {
        // (measured, profiles/r03_quantizers_bench.json: the 512-MiB probability tensors gain 10-25 % from owned pieces --
        //  block_log, whose three tables cost most per workgroup, with 4 of them, the others with 2; up to 172 MiB the
        //  grid-stride loop is as fast or faster)
        const long long n4_pad = ((a.n_elems >> 2) + 63) & ~63ll;
        const int pieces = a.n_elems >= (1ll << 26) ? (FMT == FMT_BL ? 4 : 2) : 0;
        grid = pieces ? (int)((n4_pad + 256ll * pieces - 1) / (256ll * pieces)) : grid_for(a.n_elems >> 2, 256);
#define QV_LAUNCH(L) \
        if (pieces == 0) hipLaunchKernelGGL((quant_vec_kernel<FMT, L, 0>), grid, 256, 0, st, a); \
        else if (pieces == 1) hipLaunchKernelGGL((quant_vec_kernel<FMT, L, 1>), grid, 256, 0, st, a); \
        else if (pieces == 2) hipLaunchKernelGGL((quant_vec_kernel<FMT, L, 2>), grid, 256, 0, st, a); \
        else hipLaunchKernelGGL((quant_vec_kernel<FMT, L, 4>), grid, 256, 0, st, a)
        if (lpb != 4) grid = grid_for(a.n_elems >> 2, 256);
        switch (lpb) {
            case 1: hipLaunchKernelGGL((quant_vec_kernel<FMT, 1, 0>), grid, 256, 0, st, a); break;
            case 2: hipLaunchKernelGGL((quant_vec_kernel<FMT, 2, 0>), grid, 256, 0, st, a); break;
            case 4: QV_LAUNCH(4); break;
            case 8: hipLaunchKernelGGL((quant_vec_kernel<FMT, 8, 0>), grid, 256, 0, st, a); break;
            case 16: hipLaunchKernelGGL((quant_vec_kernel<FMT, 16, 0>), grid, 256, 0, st, a); break;
            case 32: hipLaunchKernelGGL((quant_vec_kernel<FMT, 32, 0>), grid, 256, 0, st, a); break;
            default: hipLaunchKernelGGL((quant_vec_kernel<FMT, 64, 0>), grid, 256, 0, st, a); break;
        }
#undef QV_LAUNCH
    } else {
        if (a.ybf) return MI355Q_E_UNSUPPORTED;            // bf16 output: vector path only
        grid = grid_for(a.n_blocks, 16);
        hipLaunchKernelGGL((quant_generic_kernel<FMT>), grid, 256, 0, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (needs_fixup && (a.flags & MI355Q_ZERO_BLOCK_FAST) == 0u) {
        // (a grid that exits at once costs its dispatch: 256 workgroups for the launch-bound sizes)
        hipLaunchKernelGGL((zero_fixup_kernel<FMT>), a.zmap ? FIXUP_GRID_MAP : a.n_elems >= (1 << 24) ? FIXUP_GRID : 256, 256, 0, st, a,
                           grid < WS_SLOTS ? grid : WS_SLOTS);
        e = hipGetLastError();
    }
    return (int)e;
}



// ceil(log2(v)) for finite v > 0 without branches: v_frexp splits normals and subnormals alike
__device__ __forceinline__ int ceil_log2_frexp(float v, const Lut& lut) {
    const int k = __builtin_amdgcn_frexp_expf(v) - 1;                                   // v = 2^k (1 + m 2^-23)
    const unsigned m = __float_as_uint(__builtin_amdgcn_frexp_mantf(v)) & 0x7FFFFFu;
    return k + ((m != 0u && m >= lut.a[lut_index(k)]) ? 1 : 0);
}
// signed integer mantissa of one element (block_fp.py:69-79): sign(x+1e-9) * min(rne((|x|+1e-9) 2^(mb-e)), 2^mb-1)
__device__ __forceinline__ float mant_f(float x, int up, float mmax) {
    const float t = x + EPS9;
    const float m = fminf(__builtin_rintf(__builtin_ldexpf(fabsf(x) + EPS9, up)), mmax);
    const float sm = __builtin_copysignf(m, t);
    return t == 0.f ? 0.f : sm;
}
// DPP helpers on ints: rotate within a row of 16 lanes
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }

// ---------------------------------------------------------------------------------------
// Fused activation path, ROW-aligned flavour (mi355q_align_row.h): one 256-thread workgroup per row keeps the
// row's packed mantissas in registers (4 values per lane and 1024-value slab), decides the row's exponent,
// then writes the tiled mantissas, the effective exponents, rowflag[row] and the row scale.  cols % 64 == 0,
// cols <= 1024 * MAXIT.
// ---------------------------------------------------------------------------------------
// MX flavour (round 5, mi355q_mx.hip): the same row arithmetic, the operand written for v_mfma_scale_f32_16x16x128_f8f6f4
// instead -- FP6 e2m3 codes of the integer mantissas (32 values = 24 bytes per lane of the MFMA: 16 bytes in one plane, 8 in
// another, both in the tile order of the product's LDS stages) and one E8M0 scale byte per 32 values.  The two [1,16] blocks
// of a 32-group share the scale: the block with the larger exponent carries its mantissas shifted left by the difference
// (|m| << s <= 60 with <= 4 significant bits is exact in e2m3: s <= 3 at W4, s <= 2 at W5).  A pair that does not fit raises
// *mx.bad: the product launch then forms the exact product from the fp32 tensors itself (uniform over its grid).
struct MxOut {
    uint8_t* c16;     // [rows / 16][K / 128] pieces of 1 KiB: [32-group 0..3][row 0..15][16 bytes]
    uint8_t* c8;      // [rows / 32][K / 128] pieces of 1 KiB: [16-row half 0..1][32-group 0..3][row 0..15][8 bytes]
    uint8_t* sc;      // [rows / 64][K / 128] pieces of 256 B: [32-group 0..3][row 0..15][16-row quarter 0..3]
    int* bad;
    int* bad_clear;   // the flag word of the NEXT call (callers alternate between two): cleared here, by one thread
};

// PRE: which step in front of the quantiser is COMPILED IN -- 0 none, 1 any (a.pre_op decides), 2 LlamaRMSNorm only, 3 LayerNorm only,
// 4 the elementwise ones (relu, silu(x) * x2) only.  Round 6: with every path compiled in, the register count was the maximum over
// them (silu_mul's second row, LayerNorm's two vectors): a row behind RMSNorm paid for all of it.
template <int MAXIT, bool FULL, bool SEG, bool MX = false, int PRE = 1, bool ST16 = false, int WPS = 1>
__global__ __launch_bounds__(256, WPS) void bfp_quant_align_rows_kernel(const QuantArgs a, int8_t* __restrict__ mt,
                                                                   uint8_t* __restrict__ flag, float* __restrict__ rscale,
                                                                   int exp_offset, int* __restrict__ list,
                                                                   int* __restrict__ list_to_clear, int bcap, const MxOut mx = MxOut{}) {
    __shared__ Lut lut;
    __shared__ RowAlignSmem rsm;
    __shared__ float norm_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long K = a.cols;
    const int nkb = (int)(K >> 4), nit = (nkb + 63) >> 6;
    const long long rows16 = a.rows & ~127ll;               // rows covered by whole groups of 8 piece rows
    // Workgroup -> row: the 16 rows of one piece row (they share every 1-KiB piece they write, 16 bytes each per block)
    // go to workgroups b, b + 8, b + 16, ... -- the ones that share an XCD and therefore an L2, which merges their
    // partial-line stores (consecutive workgroup ids are dealt round-robin over the 8 XCDs, whose L2s are not coherent:
    // lines written from two of them leave as two masked partial writes).  Speed only; any mapping is correct.
    auto row_of = [&](long long wi) {
        if (wi >= rows16) return wi;
        const long long grp = wi >> 7, in = wi & 127;       // 128 rows = 8 piece rows, one per XCD
        return (grp << 7) + ((in & 7) << 4) + (in >> 3);
    };
    // float4 f of a row of `base`.  Plain rows: base[row][K].  Segmented rows (a.seg_len > 0: the rank-major result of an
    // all-gather over out_features shards, [P][rows][seg_len] with the segments a.seg_stride elements apart): element k of
    // the row sits in segment k / seg_len -- the consumer reads the P pieces where they lie instead of a permuted copy.
    auto row_f4 = [&](const float* base, long long row, int f) -> const float4* {
        if (!SEG) return reinterpret_cast<const float4*>(base + row * K) + f;      // (compile-time: the plain rows' addressing costs 3 us at 4096 x 4096 otherwise)
        const long long k = (long long)f * 4, sgm = k / a.seg_len;
        return reinterpret_cast<const float4*>(base + sgm * a.seg_stride + row * a.seg_len + (k - sgm * a.seg_len));
    };
    auto load_raw = [&](float4 (&v)[MAXIT], long long row) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int kb = it * 64 + wave * 16 + (lane >> 2);
            v[it] = (FULL || (it < nit && kb < nkb)) ? *row_f4(a.x, row, it * 256 + tid) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto load_row = [&](float4 (&v)[MAXIT], long long row) {
        load_raw(v, row);
        if constexpr (PRE == 0) return;                        // (plain rows: none of the pre-op code, nor its registers)
        const int pre_op = PRE == 2 ? MI355Q_PRE_RMSNORM : (PRE == 3 ? MI355Q_PRE_LAYERNORM : a.pre_op);
        if ((PRE == 1 || PRE == 2) && pre_op == MI355Q_PRE_RMSNORM) {                          // (uniform)
            // LlamaRMSNorm (modeling_llama.py:88-92): x * rsqrt(mean(x^2) + eps), then weight * that, each product rounded
            // to fp32 like the reference's separate ops.  The mean is summed in a fixed order (lane partials over the
            // row's chunks, xor tree over the wave, the four waves in turn): the same bits run after run, not the bits of
            // torch's reduction kernel (no two of its back ends agree on those either).
            float ss = 0.f;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) ss += (v[it].x * v[it].x + v[it].y * v[it].y) + (v[it].z * v[it].z + v[it].w * v[it].w);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
            __syncthreads();                                   // (the previous row's readers are done with norm_part)
            if (lane == 0) norm_part[wave] = ss;
            __syncthreads();
            const float var = (((norm_part[0] + norm_part[1]) + norm_part[2]) + norm_part[3]) * (1.0f / (float)K);
            const float rs = rsqrtf(var + a.pre_eps);
            const float4* __restrict__ w4 = reinterpret_cast<const float4*>(a.x2);
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int kb = it * 64 + wave * 16 + (lane >> 2);
                if (FULL || (it < nit && kb < nkb)) {
                    const float4 w = w4[it * 256 + tid];
                    float4 h = make_float4(v[it].x * rs, v[it].y * rs, v[it].z * rs, v[it].w * rs);
                    asm volatile("" : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w));       // (two roundings, as two ops)
                    v[it] = make_float4(w.x * h.x, w.y * h.y, w.z * h.z, w.w * h.w);
                }
            }
        } else if ((PRE == 1 || PRE == 3) && pre_op == MI355Q_PRE_LAYERNORM) {                  // (uniform)
            // nn.LayerNorm over the row (OPT's self_attn_layer_norm / final_layer_norm, modeling_opt.py:391-415):
            // (x - mean) * rsqrt(var + eps) * weight + bias, biased variance, mean first and the centred squares after it
            // (the row sits in registers), both summed in the fixed order described above.
            auto row_sum = [&](float part) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
                __syncthreads();
                if (lane == 0) norm_part[wave] = part;
                __syncthreads();
                return ((norm_part[0] + norm_part[1]) + norm_part[2]) + norm_part[3];
            };
            float s1 = 0.f;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) s1 += (v[it].x + v[it].y) + (v[it].z + v[it].w);
            const float mean = row_sum(s1) * (1.0f / (float)K);
            float s2 = 0.f;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int kb = it * 64 + wave * 16 + (lane >> 2);
                if (FULL || (it < nit && kb < nkb)) {
                    v[it] = make_float4(v[it].x - mean, v[it].y - mean, v[it].z - mean, v[it].w - mean);
                    s2 += (v[it].x * v[it].x + v[it].y * v[it].y) + (v[it].z * v[it].z + v[it].w * v[it].w);
                }
            }
            const float rs = rsqrtf(row_sum(s2) * (1.0f / (float)K) + a.pre_eps);
            const float4* __restrict__ w4 = reinterpret_cast<const float4*>(a.x2);
            const float4* __restrict__ b4 = reinterpret_cast<const float4*>(a.x3);
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int kb = it * 64 + wave * 16 + (lane >> 2);
                if (FULL || (it < nit && kb < nkb)) {
                    const float4 w = w4[it * 256 + tid];
                    const float4 b = b4 ? b4[it * 256 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[it] = make_float4(v[it].x * rs * w.x + b.x, v[it].y * rs * w.y + b.y, v[it].z * rs * w.z + b.z,
                                        v[it].w * rs * w.w + b.w);
                }
            }
        } else if ((PRE == 1 || PRE == 4) && pre_op) {                                          // (uniform)
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int kb = it * 64 + wave * 16 + (lane >> 2);
                // (the second input of silu_mul lies like x)
                if (FULL || (it < nit && kb < nkb)) v[it] = apply_pre(a, v[it], row_f4(a.x2, row, it * 256 + tid) - (it * 256 + tid), it * 256 + tid);
            }
        }
    };
    // the first row is requested before anything else: the threshold table's own round trip (global -> LDS) and the
    // list clearing below run while it is in flight (a workgroup usually has exactly one row)
    float4 v[MAXIT];
    if ((long long)blockIdx.x < a.rows) load_row(v, row_of(blockIdx.x));
    load_lut<FMT_BFP>(lut);
    if (list_to_clear && blockIdx.x == 0) {
        const int cb = bcap < 0 ? ROW_BCAP : bcap;
        const long long words = row_list_words(a.rows, cb), bw = row_bucket_words(cb);
        if (tid < EXC_HEADER) list_to_clear[tid] = 0;
        for (long long b = EXC_HEADER + (long long)tid * bw; b < words; b += 256ll * bw) {
            list_to_clear[b] = 0;
            list_to_clear[b + 1] = 0;
        }
    }
    const int mbits_int = (int)__builtin_log2f(a.shift);
    __syncthreads();
    // Several rows per workgroup (round 4, plain rows: launch_quant_align_rows): the NEXT row is requested as soon as this one's
    // values are dead -- behind the mantissas, in front of the exponent decision, its barriers and the stores -- into the same
    // registers: a row's life was load round trip + arithmetic + decision + stores in sequence, 4 workgroups a compute unit.
    const bool early = (PRE == 0 || a.pre_op == 0) && !SEG;
    for (long long wi = blockIdx.x; wi < a.rows; wi += gridDim.x) {
        const long long row = row_of(wi);
        if (wi != (long long)blockIdx.x && !early) load_row(v, row);
        unsigned pk[MAXIT];
        int amax[MAXIT], code[MAXIT];
        // Pass 1: block maxima, shared exponents, scale exponents (the threshold table is read unconditionally: no
        // branch around an LDS read).  Block maxima are compared as unsigned bit patterns of |x| (monotone for floats).
        unsigned bmb[MAXIT];
        int up[MAXIT];
        bool big = false;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const unsigned b0 = __float_as_uint(v[it].x) & 0x7FFFFFFFu, b1 = __float_as_uint(v[it].y) & 0x7FFFFFFFu;
            const unsigned b2 = __float_as_uint(v[it].z) & 0x7FFFFFFFu, b3 = __float_as_uint(v[it].w) & 0x7FFFFFFFu;
            unsigned m = max(max(b0, b1), max(b2, b3));
            m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
            bmb[it] = m;
            const float bm1 = m != 0u ? __uint_as_float(m) : 1.0f;     // all-zero block: fill 1 (MI355Q_ZERO_BLOCK_FAST)
            const int k = __builtin_amdgcn_frexp_expf(bm1) - 1;        // bm1 = 2^k (1 + f 2^-23)
            const unsigned f = __float_as_uint(__builtin_amdgcn_frexp_mantf(bm1)) & 0x7FFFFFu;
            const unsigned thr = lut.a[lut_index(k)];
            const int e = clampi(k + ((f != 0u && f >= thr) ? 1 : 0), a.e_min, a.e_max);
            up[it] = mbits_int - e;
            code[it] = e + a.code_bias;
            big = big || up[it] >= 28;
        }
        if (__any(big)) {
            // some block of the row lies below 2^-23: the exact rule (an element equal to -1e-9 has mantissa 0 whatever
            // its magnitude, block_fp.py:69) for the whole row
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int q0 = (int)mant_f(v[it].x, up[it], a.mant_max), q1 = (int)mant_f(v[it].y, up[it], a.mant_max);
                const int q2 = (int)mant_f(v[it].z, up[it], a.mant_max), q3 = (int)mant_f(v[it].w, up[it], a.mant_max);
                const int am = (int)group_max<4>((float)max(max(abs(q0), abs(q1)), max(abs(q2), abs(q3))));
                amax[it] = bmb[it] != 0u ? am : 0;
                const unsigned lo = __builtin_amdgcn_perm((unsigned)q1, (unsigned)q0, 0x0c0c0400u);
                const unsigned hi = __builtin_amdgcn_perm((unsigned)q3, (unsigned)q2, 0x04000c0cu);
                pk[it] = lo | hi;
            }
        } else {
            // Fast path (every scale exponent < 28).  Signed mantissa of x:  sign(x + 1e-9) min(rne((|x| + 1e-9) 2^up), mmax)
            //   = rne(clamp(fma(x, 2^up, copysign(1e-9 * 2^up, x)), -mmax, mmax)):  scaling by a power of two commutes
            //   with the fp32 rounding of |x| + 1e-9, an element in [-1e-9, 0) rounds to 0 here whatever sign it carries,
            //   clamping to an integer bound commutes with rounding.  The rounding itself is the fp32 add of 1.5 * 2^23,
            //   which leaves the two's-complement mantissa in the low byte of the sum's bit pattern (|m| <= 127).
            constexpr float MAGIC = 12582912.0f;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const float sc = __builtin_ldexpf(1.0f, up[it]);
                const float es = EPS9 * sc;                             // (exact: a power-of-two multiple of 1e-9f)
                auto mant_bits = [&](float x) {
                    const float r = __builtin_fmaf(x, sc, __builtin_copysignf(es, x));
                    return __float_as_uint(__builtin_amdgcn_fmed3f(r, -a.mant_max, a.mant_max) + MAGIC);
                };
                const unsigned t0 = mant_bits(v[it].x), t1 = mant_bits(v[it].y), t2 = mant_bits(v[it].z), t3 = mant_bits(v[it].w);
                // largest |mantissa| of the block = mantissa of its largest element (rounding is monotone)
                amax[it] = (int)(mant_bits(__uint_as_float(bmb[it])) - 0x4B400000u);
                if (bmb[it] == 0u) amax[it] = 0;
                const unsigned lo = __builtin_amdgcn_perm(t1, t0, 0x0c0c0400u);
                const unsigned hi = __builtin_amdgcn_perm(t3, t2, 0x04000c0cu);
                pk[it] = lo | hi;
            }
        }
        if (early && wi + gridDim.x < a.rows) load_raw(v, row_of(wi + gridDim.x));
        if constexpr (MX) {
            if (wi == 0 && tid == 0 && mx.bad_clear) *mx.bad_clear = 0;
            const long long kp = K >> 7;
            const int q = lane & 3;
            bool bad = false;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int kb = it * 64 + wave * 16 + (lane >> 2);
                // the partner block kb ^ 1 of the 32-group lives four lanes away
                const int pc = __shfl_xor(code[it], 4), pa = __shfl_xor(amax[it], 4);
                const bool nz = amax[it] > 0, pnz = pa > 0;
                const int elo = nz ? (pnz ? min(code[it], pc) : code[it]) : (pnz ? pc : code[it]);
                const int sft = nz ? code[it] - elo : 0;
                int S = 127 + elo - a.code_bias - mbits_int + 3;          // E8M0: code 2^s m / 8 times 2^(S - 127) = m 2^(e - mbits)
                const bool ok = (!nz || (sft <= 3 && (amax[it] << sft) <= 60)) && S >= 0 && S <= 254;
                bad = bad || ((FULL || (it < nit && kb < nkb)) && !ok);
                S = clampi(S, 0, 254);
                const int sh = min(sft, 3);
                unsigned c24 = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = (int)(signed char)(pk[it] >> (8 * j));
                    // e2m3 code of the integer vv = |m| << s (<= 60, <= 4 significant bits: exact): vv / 8 below 1, else exponent
                    // and top three fraction bits of float(vv), re-biased (a group that does not fit encodes garbage: it is flagged)
                    const unsigned vv = (unsigned)abs(m) << sh;
                    const unsigned enc = vv < 8u ? vv : (__float_as_uint((float)vv) >> 20) - 1032u;
                    c24 |= ((m < 0 ? 32u : 0u) | enc) << (6 * j);
                }
                // the block's 12 bytes as three dwords, dword q by lane q of the block: bits [32 q, 32 q + 32) of the 96
                const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)c24, 0x39, 0xF, 0xF, true);      // quad_perm [1,2,3,0]: lane q + 1
                const unsigned dq = (c24 >> (8 * q)) | (nb << (24 - 8 * q));
                if (FULL || (it < nit && kb < nkb)) {
                    const int blk = kb & 1, g4 = (kb >> 1) & 3;
                    const long long ks = kb >> 3;
                    uint8_t* p16 = mx.c16 + (((row >> 4) * kp + ks) << 10) + g4 * 256 + (row & 15) * 16;
                    uint8_t* p8 = mx.c8 + (((row >> 5) * kp + ks) << 10) + ((row >> 4) & 1) * 512 + g4 * 128 + (row & 15) * 8;
                    if (blk == 0) {
                        if (q < 3) *reinterpret_cast<unsigned*>(p16 + q * 4) = dq;
                        else mx.sc[(((row >> 6) * kp + ks) << 8) + g4 * 64 + (row & 15) * 4 + ((row >> 4) & 3)] = (uint8_t)S;
                    } else {
                        if (q == 0) *reinterpret_cast<unsigned*>(p16 + 12) = dq;
                        else if (q < 3) *reinterpret_cast<unsigned*>(p8 + (q - 1) * 4) = dq;
                    }
                }
            }
            if (__any(bad) && lane == 0) *mx.bad = 1;                   // (plain store of the same value by whoever saw one)
            if (wi + gridDim.x < a.rows) __syncthreads();
            continue;
        }
        int E = 0;
        // bcap < 0: no alignment at all -- every block keeps its own exponent, rowflag 0 (operands for the blockwise
        // kernel: inputs whose block exponents spread too far for any row window, e.g. SiLU-gated MLP activations)
        const bool flagged = bcap < 0 ? false : align_row<MAXIT, FULL>(pk, amax, code, nit, nkb, row, list, rsm, E, bcap);
        // tiled address of this lane's 4 bytes in slab 0 (block 16 wave + lane / 4: K-step 4 wave + lane / 16, 16-byte
        // block (lane / 4) & 3 of the piece, 256 bytes apart); a slab further on is 16 K-steps = 16 KiB further
        int8_t* dst = mt + ((row >> 4) * (K >> 6) + wave * 4 + (lane >> 4)) * 1024 + ((lane >> 2) & 3) * 256 +
                      (row & 15) * 16 + (lane & 3) * 4;
        uint8_t* cdst = a.code + row * nkb + wave * 16 + (lane >> 2);
        if constexpr (ST16) {
            // 16-byte stores (round 6): the quad's four lanes hold bytes 4 q .. 4 q + 3 of the quad's block in each of four
            // consecutive slabs; a 4 x 4 transpose inside the quad (two butterfly stages of DPP quad permutes) leaves lane q with the
            // WHOLE block of slab 4 g + q: one dwordx4 store per lane and four slabs instead of four dword stores (a quarter of the
            // store instructions, every one of them 16 bytes a lane), and one exponent byte per lane instead of four per quad
            // leader (20.6 -> 19.5 us at 4096 x 4096, profiles/r06_qrows_variants.txt).
            const int q = lane & 3;
            int8_t* d16 = mt + ((row >> 4) * (K >> 6) + wave * 4 + (lane >> 4)) * 1024 + ((lane >> 2) & 3) * 256 + (row & 15) * 16;
#pragma unroll
            for (int g = 0; g < MAXIT / 4; ++g) {
                unsigned r0 = pk[4 * g], r1 = pk[4 * g + 1], r2 = pk[4 * g + 2], r3 = pk[4 * g + 3];
                {   // stage 1: lanes q ^ 1 exchange the off-diagonal halves of (r0, r1) and of (r2, r3)
                    const bool odd = q & 1;
                    const unsigned s01 = odd ? r0 : r1, s23 = odd ? r2 : r3;
                    const unsigned g01 = (unsigned)__builtin_amdgcn_mov_dpp((int)s01, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
                    const unsigned g23 = (unsigned)__builtin_amdgcn_mov_dpp((int)s23, 0xB1, 0xF, 0xF, true);
                    if (odd) { r0 = g01; r2 = g23; } else { r1 = g01; r3 = g23; }
                }
                {   // stage 2: lanes q ^ 2 exchange (r0, r2) and (r1, r3)
                    const bool hi = q & 2;
                    const unsigned s02 = hi ? r0 : r2, s13 = hi ? r1 : r3;
                    const unsigned g02 = (unsigned)__builtin_amdgcn_mov_dpp((int)s02, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
                    const unsigned g13 = (unsigned)__builtin_amdgcn_mov_dpp((int)s13, 0x4E, 0xF, 0xF, true);
                    if (hi) { r0 = g02; r1 = g13; } else { r2 = g02; r3 = g13; }
                }
                // lane q now holds dwords 0..3 (= bytes 0..15) of the block of slab 4 g + q
                const int it = 4 * g + q, kb = it * 64 + wave * 16 + (lane >> 2);
                const int cq = q == 0 ? code[4 * g] : q == 1 ? code[4 * g + 1] : q == 2 ? code[4 * g + 2] : code[4 * g + 3];
                if (FULL || (it < nit && kb < nkb)) {
                    *reinterpret_cast<uint4*>(d16 + it * 16384) = make_uint4(r0, r1, r2, r3);
                    cdst[it * 64] = (uint8_t)(flagged ? E : cq);
                }
            }
        } else {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int kb = it * 64 + wave * 16 + (lane >> 2);
            if (FULL || (it < nit && kb < nkb)) {
                *reinterpret_cast<unsigned*>(dst + it * 16384) = pk[it];
                if ((lane & 3) == 0) cdst[it * 64] = (uint8_t)(flagged ? E : code[it]);
            }
        }
        }
        if (tid == 0) {
            flag[row] = flagged ? 1 : 0;
            rscale[row] = flagged ? __builtin_ldexpf(1.0f, E - exp_offset) : 0.0f;
        }
        if (wi + gridDim.x < a.rows) __syncthreads();       // (the next row reuses the decision words in LDS)
    }
}

int launch_quant_align_rows(const QuantArgs& a, int8_t* mt, uint8_t* flag, float* rscale, int exp_offset, int* list,
                            int* list_to_clear, hipStream_t st, int bcap) {
    long long grid = a.rows;
    if (grid > 65536) grid = 65536;
    // plain rows: a fixed grid of SIX workgroups per compute unit (what the build without the pre-op code admits: 77 registers),
    // several rows each with the next row's loads in flight.  Round 6, tools/dbg/qrows_ab.py -> profiles/r06_qrows_variants.txt, at
    // 4096 x 4096: 20.6 us (round 5: 121 registers, 4 workgroups a compute unit, dword stores) -> 19.2 (no pre-op code, 16-byte
    // stores) -> 18.0 (six workgroups a compute unit).  MI355Q_QROWS_GRID / MI355Q_QROWS_VARIANT = 0 (round 5's kernel): A/B runs.
    const int qgrid_env = getenv("MI355Q_QROWS_GRID") ? atoi(getenv("MI355Q_QROWS_GRID")) : 0;
    const bool old = getenv("MI355Q_QROWS_VARIANT") && atoi(getenv("MI355Q_QROWS_VARIANT")) == 0;
    const int qgrid = qgrid_env > 0 ? qgrid_env : (old ? 1024 : 1536);
    if (qgrid > 0 && a.pre_op == 0 && !a.seg_len && grid > qgrid) grid = qgrid;
    if (qgrid_env > 0 && a.pre_op != 0 && !a.seg_len && grid > qgrid_env) grid = qgrid_env;       // (A/B: rows behind a pre-op take one workgroup each)
    if (grid < 1) grid = 1;
    // segmented rows: the round-5 build; rows with a pre-op in front: its code compiled in; plain rows: the lean build.  16-byte
    // stores where every lane holds a block in every slab (the guarded flavour of that transpose put its scalars in scratch memory)
#define MI355Q_LAUNCH_ROWS_T(...)                                                                                     \
    hipLaunchKernelGGL((bfp_quant_align_rows_kernel<__VA_ARGS__>), (unsigned)grid, 256, 0, st, a, mt, flag, rscale, exp_offset, list, \
                       list_to_clear, bcap)
#define MI355Q_LAUNCH_ROWS(MAXIT_, FULL_)                                                                             \
    if (a.seg_len && a.pre_op == MI355Q_PRE_RMSNORM) MI355Q_LAUNCH_ROWS_T(MAXIT_, false, true, false, 2);             \
    else if (a.seg_len && a.pre_op == MI355Q_PRE_LAYERNORM) MI355Q_LAUNCH_ROWS_T(MAXIT_, false, true, false, 3);      \
    else if (a.seg_len && a.pre_op) MI355Q_LAUNCH_ROWS_T(MAXIT_, false, true, false, 4);                              \
    else if (a.seg_len) MI355Q_LAUNCH_ROWS_T(MAXIT_, false, true, false, 0);                                          \
    else if (old) MI355Q_LAUNCH_ROWS_T(MAXIT_, FULL_, false);                                                         \
    else if (a.pre_op == MI355Q_PRE_RMSNORM) MI355Q_LAUNCH_ROWS_T(MAXIT_, FULL_, false, false, 2, FULL_);             \
    else if (a.pre_op == MI355Q_PRE_LAYERNORM) MI355Q_LAUNCH_ROWS_T(MAXIT_, FULL_, false, false, 3, FULL_);           \
    else if (a.pre_op) MI355Q_LAUNCH_ROWS_T(MAXIT_, FULL_, false, false, 4, FULL_);                                   \
    else MI355Q_LAUNCH_ROWS_T(MAXIT_, FULL_, false, false, 0, FULL_)
    // (round 6: rows of <= 1024 / <= 2048 values -- OPT-125m / 350m / 1.3B widths -- hold one / two slabs a lane instead of four guarded
    //  ones: fewer registers, more rows resident on a compute unit.  MI355Q_QROWS_SHORT=0: the four-slab build for them too, A/B runs)
    static const int short_rows = getenv("MI355Q_QROWS_SHORT") ? atoi(getenv("MI355Q_QROWS_SHORT")) : 1;
    if (a.cols == 4096) MI355Q_LAUNCH_ROWS(4, true);            // every lane holds a block in every slab: no guards
    else if (short_rows && !a.seg_len && !old && a.cols <= 1024) MI355Q_LAUNCH_ROWS(1, false);
    else if (short_rows && !a.seg_len && !old && a.cols <= 2048) MI355Q_LAUNCH_ROWS(2, false);
    else if (a.cols <= 4096) MI355Q_LAUNCH_ROWS(4, false);
    else if (a.cols == 8192) MI355Q_LAUNCH_ROWS(8, true);
    else if (a.cols <= 8192) MI355Q_LAUNCH_ROWS(8, false);
    else if (short_rows && !a.seg_len && !old && a.cols <= 12288) MI355Q_LAUNCH_ROWS(12, false);      // (Llama-7B's 11008: 11 slabs of 16)
    else if (a.cols == 16384) MI355Q_LAUNCH_ROWS(16, true);
    else if (a.cols <= 16384) MI355Q_LAUNCH_ROWS(16, false);
    else
        return MI355Q_E_UNSUPPORTED;
#undef MI355Q_LAUNCH_ROWS
#undef MI355Q_LAUNCH_ROWS_T
    return (int)hipGetLastError();
}

// block_fp -> MX operand (mi355q_mx.hip): rows x K fp32, K % 128 == 0, width <= 5; `bad` is RAISED, never cleared here
int launch_quant_mx_rows(const QuantArgs& a, uint8_t* c16, uint8_t* c8, uint8_t* sc, int* bad, int* bad_clear, hipStream_t st) {
    long long grid = a.rows;
    constexpr int qgrid = 1536;                             // (six resident workgroups a compute unit at 77 registers, like the int8 flavour)
    if (qgrid > 0 && a.pre_op == 0 && grid > qgrid) grid = qgrid;
    if (grid > 65536) grid = 65536;
    if (grid < 1) grid = 1;
    const MxOut mx{c16, c8, sc, bad, bad_clear};
#define MI355Q_LAUNCH_MX(MAXIT_, FULL_)                                                                               \
    hipLaunchKernelGGL((bfp_quant_align_rows_kernel<MAXIT_, FULL_, false, true, 0>), (unsigned)grid, 256, 0, st, a, nullptr, nullptr, nullptr, \
                       0, nullptr, nullptr, -1, mx)
    // (round 6: the build without the pre-op code -- this entry never has one -- and the short-row builds, as the int8 flavour)
    if (a.cols == 4096) MI355Q_LAUNCH_MX(4, true);
    else if (a.cols <= 1024) MI355Q_LAUNCH_MX(1, false);
    else if (a.cols <= 2048) MI355Q_LAUNCH_MX(2, false);
    else if (a.cols <= 4096) MI355Q_LAUNCH_MX(4, false);
    else if (a.cols == 8192) MI355Q_LAUNCH_MX(8, true);
    else if (a.cols <= 8192) MI355Q_LAUNCH_MX(8, false);
    else if (a.cols == 16384) MI355Q_LAUNCH_MX(16, true);
    else if (a.cols <= 16384) MI355Q_LAUNCH_MX(16, false);
    else
        return MI355Q_E_UNSUPPORTED;
#undef MI355Q_LAUNCH_MX
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// block_fp quantise ([1,16] blocks along the last dim) straight into TILED bf16: the operand of the bf16 flavour of the
// tile GEMM (mi355q_gemm_v8.hip) -- operands whose blocks keep their own exponents.  1-KiB pieces of 16 rows x 32 values,
// [8-value group 0..3][row 0..15][16 bytes] inside (mi355q_gemm_v2.h with K counted in bytes).  One workgroup per row;
// optionally also the fp32 fake-quantised values in place of / next to x (the weights' first-forward overwrite).
// ---------------------------------------------------------------------------------------
template <int FMT>          // FMT_BFP, or FMT_BM: block_minifloat with the same [1,16] blocks (values of <= 7 mantissa bits: exact in bf16)
__global__ __launch_bounds__(256) void bfp_quant_bf16_tiled_kernel(const QuantArgs a, uint16_t* __restrict__ yt, int cast_only) {
    __shared__ Lut lut;
    __shared__ float norm_part[4];
    load_lut<FMT>(lut);
    const long long K = a.cols, kp = (K * 2) >> 6;
    const int nhalf = (int)(K >> 3);                      // 8-value half blocks: one lane each, 16 bytes of bf16
    const int mbits_int = (int)__builtin_log2f(a.shift);
    const long long rows16 = a.rows & ~127ll;
    for (long long wi = blockIdx.x; wi < a.rows; wi += gridDim.x) {
        // (rows of one 16-row piece row on workgroups that share an XCD: bfp_quant_align_rows_kernel)
        long long row = wi;
        if (wi < rows16) {
            const long long grp = wi >> 7, in = wi & 127;
            row = (grp << 7) + ((in & 7) << 4) + (in >> 3);
        }
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x + row * K);
        float4* __restrict__ y4 = a.y ? reinterpret_cast<float4*>(a.y + row * K) : nullptr;
        unsigned char* prow = reinterpret_cast<unsigned char*>(yt) + (row >> 4) * kp * 1024 + (row & 15) * 16;
        // LlamaRMSNorm in front of the quantiser (MI355Q_PRE_RMSNORM; a.x2 = the norm's weight [K]): the row's mean of squares first --
        // the row quantiser's arithmetic and summation order (lane partials over float4 tid, tid + 256, ..., xor tree over the wave,
        // the four waves in turn: bfp_quant_align_rows_kernel), so both routes of a layer see the same normalised values -- then
        // w * (x * rsqrt(mean + eps)), two roundings, on the row's second pass (an L2 hit)
        float rs = 0.f;
        if (a.pre_op == MI355Q_PRE_RMSNORM) {                          // (uniform)
            float ss = 0.f;
            for (int f = (int)threadIdx.x; f < (int)(K >> 2); f += 256) {
                const float4 v = x4[f];
                ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) norm_part[threadIdx.x >> 6] = ss;
            __syncthreads();
            rs = rsqrtf((((norm_part[0] + norm_part[1]) + norm_part[2]) + norm_part[3]) * (1.0f / (float)K) + a.pre_eps);
        }
        // (the next trip's inputs -- x, and the second input of silu_mul / the norm's weight -- are requested before this trip's arithmetic:
        //  a row is 2-6 trips, and load -> arithmetic -> store in sequence left the memory system idle half the time: 2048 x 11008 with
        //  silu_mul 66 -> 5x us)
        const float4* __restrict__ u4 = a.pre_op == MI355Q_PRE_SILU_MUL ? reinterpret_cast<const float4*>(a.x2 + row * K)
                                        : (a.pre_op == MI355Q_PRE_RMSNORM ? reinterpret_cast<const float4*>(a.x2) : nullptr);
        auto fetch = [&](int j0, float4& p0, float4& p1, float4& q0, float4& q1) {
            const int j = j0 + (int)threadIdx.x;
            p0 = p1 = q0 = q1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < nhalf) {
                p0 = x4[2 * j]; p1 = x4[2 * j + 1];
                if (u4) { q0 = u4[2 * j]; q1 = u4[2 * j + 1]; }
            }
        };
        float4 nv0, nv1, nu0, nu1;
        fetch(0, nv0, nv1, nu0, nu1);
        for (int j0 = 0; j0 < nhalf; j0 += 256) {                     // uniform trip count (pair exchange inside)
            const int j = j0 + (int)threadIdx.x;
            const bool valid = j < nhalf;
            float4 v0 = nv0, v1 = nv1;
            const float4 u0 = nu0, u1 = nu1;
            if (j0 + 256 < nhalf) fetch(j0 + 256, nv0, nv1, nu0, nu1);
            if (valid && a.pre_op == MI355Q_PRE_RMSNORM) {
                const float4 w0 = u0, w1 = u1;
                float4 h0 = make_float4(v0.x * rs, v0.y * rs, v0.z * rs, v0.w * rs), h1 = make_float4(v1.x * rs, v1.y * rs, v1.z * rs, v1.w * rs);
                asm volatile("" : "+v"(h0.x), "+v"(h0.y), "+v"(h0.z), "+v"(h0.w), "+v"(h1.x), "+v"(h1.y), "+v"(h1.z), "+v"(h1.w));   // (two roundings, as two ops)
                v0 = make_float4(w0.x * h0.x, w0.y * h0.y, w0.z * h0.z, w0.w * h0.w);
                v1 = make_float4(w1.x * h1.x, w1.y * h1.y, w1.z * h1.z, w1.w * h1.w);
            } else if (valid && a.pre_op == MI355Q_PRE_RELU) {
                v0 = make_float4(pre_relu(v0.x), pre_relu(v0.y), pre_relu(v0.z), pre_relu(v0.w));
                v1 = make_float4(pre_relu(v1.x), pre_relu(v1.y), pre_relu(v1.z), pre_relu(v1.w));
            } else if (valid && a.pre_op == MI355Q_PRE_SILU_MUL) {
                v0 = make_float4(pre_silu_mul(v0.x, u0.x), pre_silu_mul(v0.y, u0.y), pre_silu_mul(v0.z, u0.z), pre_silu_mul(v0.w, u0.w));
                v1 = make_float4(pre_silu_mul(v1.x, u1.x), pre_silu_mul(v1.y, u1.y), pre_silu_mul(v1.z, u1.z), pre_silu_mul(v1.w, u1.w));
            }
            float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (!cast_only) {
                unsigned m = 0u;
#pragma unroll
                for (int t = 0; t < 8; ++t) m = max(m, __float_as_uint(o[t]) & 0x7FFFFFFFu);
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true));   // the block's other half
                const float bm1 = m != 0u ? __uint_as_float(m) : 1.0f;
                if (FMT == FMT_BM) {                                    // (all-zero block: fill 1, every element stays 0)
                    unsigned code;
                    const BlockParam bp = block_param<FMT_BM>(bm1, a, lut, code);
                    int q;
#pragma unroll
                    for (int t = 0; t < 8; ++t) o[t] = quant_elem<FMT_BM>(o[t], bp, a, lut, q);
                }
                const int k = __builtin_amdgcn_frexp_expf(bm1) - 1;
                const unsigned f = __float_as_uint(__builtin_amdgcn_frexp_mantf(bm1)) & 0x7FFFFFu;
                const int e = clampi(k + ((f != 0u && f >= lut.a[lut_index(k)]) ? 1 : 0), a.e_min, a.e_max);
                const int up = mbits_int - e;
                if (FMT == FMT_BM) {
                } else if (__any(up >= 28)) {                           // blocks below 2^-23: the general rule
                    BlockParam bp;
                    bp.p = e;
                    bp.eps = 0.f;
                    int q;
#pragma unroll
                    for (int t = 0; t < 8; ++t) o[t] = quant_elem<FMT_BFP>(o[t], bp, a, lut, q);
                } else {
                    // m = rne(clamp(fma(x, 2^up, copysign(1e-9 2^up, x)))) as in bfp_quant_align_rows_kernel; value = m 2^-up
                    constexpr float MAGIC = 12582912.0f;
                    const float sc = __builtin_ldexpf(1.0f, up), es = EPS9 * sc, inv = __builtin_ldexpf(1.0f, -up);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float x = o[t];
                        const float r = __builtin_amdgcn_fmed3f(__builtin_fmaf(x, sc, __builtin_copysignf(es, x)), -a.mant_max, a.mant_max);
                        const float q = __builtin_fmaf(r + MAGIC, inv, -MAGIC * inv);
                        o[t] = fabsf(x) <= ATOL ? x : q;
                    }
                }
            }
            if (valid) {
                if (y4) {
                    y4[2 * j] = make_float4(o[0], o[1], o[2], o[3]);
                    y4[2 * j + 1] = make_float4(o[4], o[5], o[6], o[7]);
                }
                const int kb = j * 16;                                 // byte offset of these 8 values in the bf16 row
                *reinterpret_cast<uint4*>(prow + (long long)(kb >> 6) * 1024 + ((kb >> 4) & 3) * 256) =
                    make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
            }
        }
    }
}

int launch_quant_bf16_tiled(const QuantArgs& a, uint16_t* yt, hipStream_t st, bool cast_only, int fmt) {
    long long grid = a.rows;
    if (grid > 65536) grid = 65536;
    if (grid < 1) grid = 1;
    if (fmt == FMT_BM) hipLaunchKernelGGL(bfp_quant_bf16_tiled_kernel<FMT_BM>, (unsigned)grid, 256, 0, st, a, yt, cast_only ? 1 : 0);
    else hipLaunchKernelGGL(bfp_quant_bf16_tiled_kernel<FMT_BFP>, (unsigned)grid, 256, 0, st, a, yt, cast_only ? 1 : 0);
    return (int)hipGetLastError();
}

int launch_quant(const QuantArgs& a, int fmt, bool needs_fixup, hipStream_t st) {
    switch (fmt) {
        case FMT_BFP: return launch_format<FMT_BFP>(a, needs_fixup, st);
        case FMT_BM: return launch_format<FMT_BM>(a, needs_fixup, st);
        default: return launch_format<FMT_BL>(a, needs_fixup, st);
    }
}

int launch_integer(const float* x, float* y, long long n, float scale, float lo, float hi, hipStream_t st) {
    hipLaunchKernelGGL(integer_kernel, grid_for(n, 256), 256, 0, st, x, y, n, scale, lo, hi);
    return (int)hipGetLastError();
}

}  // namespace mi355q
