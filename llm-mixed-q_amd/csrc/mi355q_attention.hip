// mi355q_attention.hip -- the reference's quantised attention core as ONE pass (bfp_attention_kernel: per 16 queries, the
// score strip resident in registers; bfp_attention_stream_kernel further down: any number of keys, scores formed twice):
//
//     scores = bmm_0(Qa(q), Qb(k^T))            quantized_functions/matmul.py:146-196 (block_fp), callers
//     scores = scores / scale                   modeling_llama.py:318-322 (OPT scales q before the product instead)
//     scores = max(scores + mask, finfo.min)    modeling_opt.py:262-276, modeling_llama.py:323-329
//     probs  = softmax(scores, -1)              fp32
//     out    = bmm_1(Qc(probs), Qd(v))          modeling_opt.py:312, modeling_llama.py:341
//
// Neither the scores nor the probabilities [heads, T, T] exist in memory: a workgroup keeps the score strip of its 16
// queries (16 x T fp32, T <= 2048) in MFMA accumulators -- 128 VGPRs per lane at T = 2048 -- from the first product to
// the second.  Block structure of the four quantisers ([1,16] blocks along each operand's LAST dim, like the reference):
//   q [.., M, D]  blocks of 16 along D  (the contraction of the first product): quantised in registers here
//   k^T [.., D, T] blocks of 16 consecutive KEYS at a fixed d: attn_pack_kv_kernel / attn_pack_k (k is taken untransposed)
//   probs          blocks of 16 consecutive keys of a query = one 16 x 16 score tile's row: quantised in registers
//   v [.., T, D]  blocks of 16 along D at a fixed key: attn_pack_kv_kernel / attn_pack_v
// MFMA operand roles (v_mfma_f32_16x16x32_bf16; A rows x k, B k x columns, lane (c = lane % 16, g = lane / 16) holds k
// slots 8 g .. 8 g + 7 of row / column c and gets rows 4 g .. 4 g + 3 of column c of the result):
//   scores tile t (16 keys):  A = K fragment (rows = keys 16 t + c, slots = d),  B = Q fragment (columns = queries)
//        -> lane holds scores[query c][keys 16 t + 4 g + 0..3]
//   output:  A = V fragment (rows = d, slots = keys),  B = P fragment (columns = queries, slots = keys)
//        -> lane holds out[query c][d = 16 dt + 4 g + 0..3]: 16-byte stores.
//   The P fragment of a lane is made of its own values of TWO score tiles a, b: slot j <-> key 16 (j < 4 ? a : b) + 4 g +
//   (j & 3); attn_pack_v stores V with the same slot order, so no value ever changes lanes between the products.
// Wave w owns the key tiles t = w, w + KW, w + 2 KW, ... (KW = 4 or 8 key-waves; interleaved: under a causal mask every wave
// loses the same share);
// tiles behind the horizon of the workgroup's last query are skipped altogether (probabilities exactly 0).  With KW = 4 a
// workgroup is two such groups (32 queries) that walk the same fragments in step: every other fragment request is an L1 hit.
// Row statistics (max, sum of exponentials) are combined over the 4 lane groups by lane swaps and over the KW waves through
// LDS; the partial outputs of the waves are summed through LDS in wave order (reproducible).
// Arithmetic: products of two block_fp values (width <= 9) are exact in fp32, accumulation is fp32 like the reference's
// GEMMs (order differs: the tolerance of the matmul tests); exp to ~1 ulp, quotient corrected once (mi355q_matmul.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_quant_dev.h"

namespace mi355q {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int AT_MAX_T = 2048, AT_MAX_D = 128;

// element of a block with shared exponent p (mi355q_matmul.hip: quant_elem_fused)
__device__ __forceinline__ float at_quant(float x, int up, int down, float mant_max) {
    const float m = fminf(__builtin_rintf(__builtin_ldexpf(fabsf(x) + EPS9, up)), mant_max);
    const float q = __builtin_copysignf(__builtin_ldexpf(m, down), x);
    return fabsf(x) <= ATOL ? x : q;
}
__device__ __forceinline__ float at_exp_neg(float x) {
    x = fmaxf(x, -104.0f);
    constexpr float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f, LN2 = 0.693147182464599609375f;
    const float t = x * L2E_HI;
    float r = __builtin_fmaf(x, L2E_HI, -t);
    r = __builtin_fmaf(x, L2E_LO, r);
    const float p = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(p, r * LN2, p);
}
__device__ __forceinline__ float at_div(float e, float l, float inv) {
    const float q = e * inv;
    return __builtin_fmaf(__builtin_fmaf(-q, l, e), inv, q);
}

// max / sum over the four lanes c16, c16 + 16, c16 + 32, c16 + 48 (one query's values of a score tile), on the VALU:
// v_permlane32_swap / v_permlane16_swap of a register with itself leave {own, partner} in the result pair for every lane
// (gfx950; no LDS crossbar round trip as with ds_bpermute, whose latency two waves per SIMD cannot hide)
__device__ __forceinline__ float at_max4(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float at_sum4(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float at_max2_16(float x) {       // lanes l, l ^ 16
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// shared exponent of a block whose largest magnitude is bmax >= 0 (block_fp.py:72-73): ceil(log2(bmax)) is the fp32
// exponent field, plus one unless bmax is a power of two -- except within 45 ulps above one (fp32 log2 rounds back onto the
// integer there: log2_tables.inc) and for subnormals, where some lane of the wave sends everybody to the table walk.
__device__ __forceinline__ int at_block_exponent(float bmax, const QuantArgs& a, const Lut& lut) {
    const unsigned bits = __float_as_uint(bmax), E = bits >> 23, f = bits & 0x7FFFFFu;
    if (__any((E == 0u && bits != 0u) || (f != 0u && f < MI355Q_LOG2_CEIL_THR_MAX))) {
        unsigned code;
        return block_param<FMT_BFP>(bmax != 0.f ? bmax : 1.0f, a, lut, code).p;
    }
    return clampi((int)E - 127 + (f != 0u ? 1 : 0), a.e_min, a.e_max);
}
// the same with the rare walk reading its threshold from MEMORY (an L2 hit; one block maximum in 2 x 10^5 takes it): a kernel that uses
// only this form need not stage the 277-entry table in LDS in front of its first instruction -- the one-pass attention kernel's
// workgroups live 13-24 us and spent 1 of them on that (tools/dbg/attn_stamps.py)
__device__ __forceinline__ int at_block_exponent_mem(float bmax, const QuantArgs& a) {
    const unsigned bits = __float_as_uint(bmax), E = bits >> 23, f = bits & 0x7FFFFFu;
    if (__any((E == 0u && bits != 0u) || (f != 0u && f < MI355Q_LOG2_CEIL_THR_MAX))) {
        int k; unsigned m;
        split_pos(bmax != 0.f ? bmax : 1.0f, k, m);
        return clampi(k + ((m != 0u && m >= mi355q_log2_ceil_thr[lut_index(k)]) ? 1 : 0), a.e_min, a.e_max);
    }
    return clampi((int)E - 127 + (f != 0u ? 1 : 0), a.e_min, a.e_max);
}
// block_fp element for x >= 0 (probabilities) given the block's scales 2^up, 2^-up (block_fp.py:69-94 with sign = +1)
// (round 6: (x + 1e-9) 2^up as ONE fused multiply-add with eps_up = 1e-9 2^up -- scaling by a power of two commutes with the
//  rounding of the sum, so fma(x, 2^up, 1e-9 2^up) is round(x + 1e-9) 2^up bit for bit; blocks whose scale overflows hold only
//  pass-through values.  One VALU operation of the ~36 per probability.)
__device__ __forceinline__ float at_quant_pos(float x, float sc_up, float eps_up, float sc_dn, float mant_max) {
    const float m = fminf(__builtin_rintf(__builtin_fmaf(x, sc_up, eps_up)), mant_max);
    return x <= ATOL ? x : m * sc_dn;
}

// ---- rotary position embedding on the way in (round 6) ------------------------------------------------------------------
// modeling_llama.py:289-299 turns q and k with quantised cos / sin tables between the projections and the first product
// (quantized_functions/rotary_positional_encoding.py:59-82; rope_kernel in mi355q_rope.hip does that as one launch: 8 B per element
// of q and k through memory, 30 us per Llama-7B layer at 2048 tokens).  With `cos` set the PACK launch below does it on the values it
// is about to quantise -- k in attn_pack_k, and q in attn_pack_q, which then leaves the quantised Q fragments of the first product
// for the attention kernels to load (instead of q itself to quantise) -- the same fp32 arithmetic (two products rounded on their own,
// then the sum), so the same bits, and the turned q / k never exist in memory.
//     x'[d] = x[d] cos[p][d] + rot[d] sin[p][d],   rot[d] = d < D / 2 ? -x[d + D / 2] : x[d - D / 2],   p = position_ids[batch][row]
// (Measured first, round 6: q turned inside the attention kernels' own Q load -- every one of a query group's four key-waves forms all
//  Q fragments, so tables and q were read four times over by a workgroup that is alone on its compute unit: 172 -> 204 us at
//  [32, 2048, 128], more than the launch it saved.)  heads: q / k are [batch x heads, rows, D]; position_ids [batch, rows].
struct RopeIn {
    const float* cos;            // [table_rows, D] quantised tables; null = no rotary embedding here
    const float* sin;
    const long long* pos;        // [batch, rows]
    long long table_rows;
    int heads;
};
__device__ __forceinline__ float at_rope(float x, float c, float rot, float s) {
    float p = x * c, q = rot * s;
    asm volatile("" : "+v"(p), "+v"(q));        // (each product rounded to fp32 on its own: no fma contraction, as mi355q_rope.hip)
    return p + q;
}

// ---- k [B, T, D] -> fragments of Qb(k^T): blocks of 16 consecutive keys at a fixed d --------------------------------
// piece (b, t, c) = 1 KiB: lane (key = lane % 16, g = lane / 16) holds d = 32 c + 8 g .. + 7 of key 16 t + (lane % 16).
// One workgroup = 256 / D key tiles; thread (tile, d) walks the 16 keys of its block (coalesced over d).
template <bool ROPE>
__device__ __forceinline__ void attn_pack_k(const QuantArgs& a, const Lut& lut, uint16_t* __restrict__ tile,
                                            const float* __restrict__ k, uint16_t* __restrict__ kf, long long T, int D,
                                            long long NT, long long bx, long long sb, long long st, const RopeIn& rope) {
    // tile: [sub-tile][c][lane][8] = D * 16 values per sub-tile
    const int tid = threadIdx.x, per = 256 / D, sub = tid / D, d = tid % D;
    const long long b = blockIdx.y, t = bx * per + sub;
    const int mbits = (int)__builtin_log2f(a.shift);
    if (sub < per && t < NT) {
        float v[16];
        float bmax = 0.f;
        if (!ROPE) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long key = t * 16 + e;
                v[e] = key < T ? k[b * sb + key * st + d] : 0.f;
            }
        } else {
            // the rotary embedding of k, see RopeIn.  A wave's 64 threads share their key tile (D >= 64 here), so the tile index goes
            // through readfirstlane: the 16 positions become scalar loads and every row / table address a scalar base + the lane's d --
            // all loads of a kind in one batch, none inside a per-key branch.  (32-bit table offsets: the launcher checks rows x D.)
            const int half = D >> 1, dp = d < half ? d + half : d - half;
            const int tu = __builtin_amdgcn_readfirstlane((int)t), Ti = (int)T;
            const long long* __restrict__ prow = rope.pos + (b / rope.heads) * T;
            const float* __restrict__ kb = k + b * sb;
            int pe[16];
            float xp[16], cs[16], sn[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = min(tu * 16 + e, Ti - 1);
                const long long p = prow[key];
                pe[e] = (int)(p < 0 ? 0 : (p >= rope.table_rows ? rope.table_rows - 1 : p)) * D;
                const float* __restrict__ row = kb + key * st;
                v[e] = row[d];
                xp[e] = row[dp];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                cs[e] = rope.cos[pe[e] + d];
                sn[e] = rope.sin[pe[e] + d];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                v[e] = tu * 16 + e < Ti ? at_rope(v[e], cs[e], d < half ? -xp[e] : xp[e], sn[e]) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) bmax = fmaxf(bmax, fabsf(v[e]));
        unsigned code;
        const int p = bmax != 0.f ? block_param<FMT_BFP>(bmax, a, lut, code).p : 0;
        const int c = d >> 5, g = (d >> 3) & 3, j = d & 7;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float q = bmax != 0.f ? at_quant(v[e], mbits - p, p - mbits, a.mant_max) : 0.f;
            tile[sub * D * 16 + (c * 64 + e + 16 * g) * 8 + j] = (uint16_t)(pack_bf16(q, 0.f) & 0xFFFFu);
        }
    }
    __syncthreads();
    // D * 16 * 2 bytes per sub-tile = D * 2 sixteen-byte chunks; 256 threads write 16 bytes each, twice
    const int chunks = per * D * 2;
    for (int ch = tid; ch < chunks; ch += 256) {
        const int s2 = ch / (D * 2), in = ch % (D * 2);
        const long long t2 = bx * per + s2;
        if (t2 < NT)
            *reinterpret_cast<uint4*>(kf + ((b * NT + t2) * (D >> 5)) * 512 + in * 8) =
                *reinterpret_cast<const uint4*>(&tile[s2 * D * 16 + in * 8]);
    }
}

// ---- v [B, T, D] -> fragments of Qd(v): blocks of 16 along D at a fixed key ------------------------------------------
// piece (b, pair, dt) = 1 KiB, pair = 4 s + w <-> key tiles a = 8 s + w, b = 8 s + 4 + w (wave w's s-th pair; kw = 4; kw = 8: 16 s + w, 16 s + 8 + w):
// lane (d = 16 dt + lane % 16, g = lane / 16) holds slot j <-> key 16 (j < 4 ? a : b) + 4 g + (j & 3).  Keys behind T: 0.
// One workgroup = 128 keys (8 tiles = 4 pairs) x D; thread (key, 16-d block) quantises one block.
__device__ __forceinline__ void attn_pack_v(const QuantArgs& a, const Lut& lut, uint16_t* __restrict__ stage,
                                            const float* __restrict__ v, uint16_t* __restrict__ vf, long long T, int D,
                                            long long NPAIR, long long bx, long long sb, long long st, int kw) {
    // stage: [dt][pair in group (w)][lane][8]
    const int tid = threadIdx.x, DT = D >> 4, PG = kw == 8 ? 8 : 4;         // pairs per workgroup: 32 PG keys
    const long long b = blockIdx.y, key0 = bx * 32 * PG;
    const int mbits = (int)__builtin_log2f(a.shift);
    for (int item = tid; item < 32 * PG * DT; item += 256) {
        const int kl = item / DT, dt = item % DT;         // key inside the group, 16-d block
        const long long key = key0 + kl;
        float x[16];
        float bmax = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (key < T) f = *reinterpret_cast<const float4*>(v + b * sb + key * st + dt * 16 + 4 * i);
            x[4 * i] = f.x; x[4 * i + 1] = f.y; x[4 * i + 2] = f.z; x[4 * i + 3] = f.w;
            bmax = fmaxf(bmax, fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w))));
        }
        unsigned code;
        const int p = bmax != 0.f ? block_param<FMT_BFP>(bmax, a, lut, code).p : 0;
        // key inside the 128-group: tile tl = kl / 16 (0..7) -> pair w of the group and half h of the pair -- tiles
        // (w, w + 4) for the resident kernel's four key-waves (kw = 4), consecutive tiles (2 w, 2 w + 1) for the streaming
        // kernel (kw = 1); g = (kl & 15) / 4
        const int tl = kl >> 4, w = kw == 1 ? tl >> 1 : tl & (kw - 1), h = kw == 1 ? tl & 1 : tl / kw, g = (kl & 15) >> 2,
                  j = 4 * h + (kl & 3);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float q = bmax != 0.f ? at_quant(x[c], mbits - p, p - mbits, a.mant_max) : 0.f;
            stage[((dt * PG + w) * 64 + c + 16 * g) * 8 + j] = (uint16_t)(pack_bf16(q, 0.f) & 0xFFFFu);
        }
    }
    __syncthreads();
    const long long s = bx;                                // pair group: pairs PG s .. PG s + PG - 1
    for (int ch = tid; ch < DT * PG * 64; ch += 256) {
        const int dt = ch / (PG * 64), w = (ch >> 6) % PG, ln = ch & 63;
        *reinterpret_cast<uint4*>(vf + (((b * NPAIR + PG * s + w) * DT + dt) * 64 + ln) * 8) =
            *reinterpret_cast<const uint4*>(&stage[((dt * PG + w) * 64 + ln) * 8]);
    }
}

// ---- q [B, M, D], turned by the rotary embedding -> the fragments of Qa(q): blocks of 16 along D at a fixed query -----------
// piece (b, query tile, c) = 1 KiB, the attention kernels' Q operand as they hold it: lane (query = lane % 16, g = lane / 16) has
// d = 32 c + 8 g .. + 7.  One thread = one [1,16] block (query, 16 d): its values, their partners d +- D / 2, the tables' 16 entries --
// sixteen consecutive threads are sixteen consecutive queries, so each of a thread's two 16-byte stores lies in a 256-byte run.
template <bool ROPE>
__device__ __forceinline__ void attn_pack_q(const QuantArgs& a, const Lut& lut, const float* __restrict__ q, uint16_t* __restrict__ qf,
                                            long long M, int D, long long NQT, long long bx, long long sb, long long sm, const RopeIn& rope,
                                            float qscale) {
    const int tid = threadIdx.x, nblk = D >> 4, QPW = 256 / nblk;          // queries per workgroup: 32 at head_dim 128, 64 at 64
    const int qi = tid % QPW, blk = tid / QPW;
    const long long b = blockIdx.y, query = bx * QPW + qi;
    if (query >= (M + 15) / 16 * 16) return;
    const int mbits = (int)__builtin_log2f(a.shift);
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.f;
    if (!ROPE && query < M) {
        const float4* x4 = reinterpret_cast<const float4*>(q + b * sb + query * sm + 16 * blk);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 xv = x4[i];
            x[4 * i] = xv.x; x[4 * i + 1] = xv.y; x[4 * i + 2] = xv.z; x[4 * i + 3] = xv.w;
        }
        // (OPT scales q by head_dim^-0.5 between q_proj and the first product, modeling_opt.py:231: one fp32 multiply, the same bits as
        //  the torch kernel that wrote q * scaling to memory)
        if (qscale != 0.f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] *= qscale;
        }
    }
    if (ROPE && query < M) {
        const int half = nblk >> 1, pblk = blk < half ? blk + half : blk - half;
        const float* __restrict__ row = q + b * sb + query * sm;
        long long p = rope.pos[(b / rope.heads) * M + query];
        p = p < 0 ? 0 : (p >= rope.table_rows ? rope.table_rows - 1 : p);
        const float4* x4 = reinterpret_cast<const float4*>(row + 16 * blk);
        const float4* y4 = reinterpret_cast<const float4*>(row + 16 * pblk);
        const float4* c4 = reinterpret_cast<const float4*>(rope.cos + p * D + 16 * blk);
        const float4* s4 = reinterpret_cast<const float4*>(rope.sin + p * D + 16 * blk);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 xv = x4[i], yv = y4[i], cv = c4[i], sv = s4[i];
            const float sg = blk < half ? -1.0f : 1.0f;                      // (rotate_half: -x[d + D/2] for the first half, x[d - D/2] behind)
            x[4 * i] = at_rope(xv.x, cv.x, sg * yv.x, sv.x); x[4 * i + 1] = at_rope(xv.y, cv.y, sg * yv.y, sv.y);
            x[4 * i + 2] = at_rope(xv.z, cv.z, sg * yv.z, sv.z); x[4 * i + 3] = at_rope(xv.w, cv.w, sg * yv.w, sv.w);
        }
    }
    float bmax = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) bmax = fmaxf(bmax, fabsf(x[i]));
    unsigned code;
    const int p = bmax != 0.f ? block_param<FMT_BFP>(bmax, a, lut, code).p : 0;
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const float q0 = bmax != 0.f ? at_quant(x[i], mbits - p, p - mbits, a.mant_max) : 0.f;
        const float q1 = bmax != 0.f ? at_quant(x[i + 1], mbits - p, p - mbits, a.mant_max) : 0.f;
        w[i >> 1] = pack_bf16(q0, q1);
    }
    // values 16 blk .. + 15 = chunk c = blk / 2, lane groups g = 2 (blk & 1) and g + 1
    uint16_t* dst = qf + ((b * NQT + (query >> 4)) * (D >> 5) + (blk >> 1)) * 512 + ((query & 15) + 32 * (blk & 1)) * 8;
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(dst + 16 * 8) = make_uint4(w[4], w[5], w[6], w[7]);
}

// the small operands in one launch: workgroups [0, kblocks) pack k, the next NPAIR / pairs-per-group pack v, the rest (if any) pack q --
// with the rotary embedding (ROPE, see RopeIn), and without it wherever a launch has enough queries for the fragments to pay: every one
// of a query group's key-waves forms ALL of the group's Q fragments in the attention kernels (four or eight times the loads and the
// quantiser arithmetic, on a workgroup that is alone on its compute unit): [32, 2048, 128] 173 -> 148 us with the fragments packed here
template <bool ROPE>
__global__ __launch_bounds__(256) void attn_pack_kv_kernel(const QuantArgs ak, const QuantArgs av, const float* __restrict__ k,
                                                           const float* __restrict__ v, uint16_t* __restrict__ kf,
                                                           uint16_t* __restrict__ vf, long long T, int D, long long NT,
                                                           long long NPAIR, int kblocks, long long ksb, long long kst,
                                                           long long vsb, long long vst, int kw, const RopeIn rope, const QuantArgs aq,
                                                           const float* __restrict__ q, uint16_t* __restrict__ qf, int vblocks,
                                                           long long qsb, long long qsm, long long M, float qscale) {
    __shared__ Lut lut;
    __shared__ __attribute__((aligned(16))) uint16_t buf[AT_MAX_D / 16 * 4 * 512];
    load_lut<FMT_BFP, true>(lut);
    if ((int)blockIdx.x < kblocks) attn_pack_k<ROPE>(ak, lut, buf, k, kf, T, D, NT, blockIdx.x, ksb, kst, rope);
    else if ((int)blockIdx.x < kblocks + vblocks) attn_pack_v(av, lut, buf, v, vf, T, D, NPAIR, (long long)blockIdx.x - kblocks, vsb, vst, kw);
    else attn_pack_q<ROPE>(aq, lut, q, qf, M, D, NT, (long long)blockIdx.x - kblocks - vblocks, qsb, qsm, rope, qscale);
}

// ---- the consumer's operand in the store epilogue (round 6) -----------------------------------------------------------
// Behind the attention core both models reshape to [tokens, heads x D] and call the out-projection (modeling_opt.py:318-328,
// modeling_llama.py:349-353), whose activation quantiser -- where that Linear runs on the per-block-exponent route --
// is mi355q_block_fp_quantize_bf16_tiled: 4 B read + 2 B written per value in a launch of its own.  Its [1,16] blocks along the
// hidden dimension are exactly a head's column tile dt of one query, i.e. the four lanes c16 + 16 g of the epilogue: with
// `out_tiled` set the kernels form the block maximum over those lanes, quantise (the Q fragments' arithmetic with the
// consumer's parameters) and store 8 bytes of bf16 a lane straight into the consumer's tiled operand (1-KiB pieces of 16 rows
// x 32 values, [8-value group][row][16 B]: mi355q_quant.hip) -- the fp32 attention output is never written.
struct AttnConsumer {
    uint16_t* out_tiled;      // null: fp32 `out` as before
    long long kp;             // pieces per 16-row tile of the operand: heads x D / 32
    float mant_max;
    int mbits, e_min, e_max;
};
__device__ __forceinline__ void at_store_consumer(const AttnConsumer& c, const f32x4& v, long long m, long long M, long long col) {
    float bmax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    bmax = at_max4(bmax);
    const unsigned bits = __float_as_uint(bmax), E = bits >> 23, f = bits & 0x7FFFFFu;
    int p;
    if (__any((E == 0u && bits != 0u) || (f != 0u && f < MI355Q_LOG2_CEIL_THR_MAX))) {       // (rare: the table decides, see at_block_exponent_mem)
        int k; unsigned mm;
        split_pos(bmax != 0.f ? bmax : 1.0f, k, mm);
        p = clampi(k + ((mm != 0u && mm >= mi355q_log2_ceil_thr[lut_index(k)]) ? 1 : 0), c.e_min, c.e_max);
    } else {
        p = clampi((int)E - 127 + (f != 0u ? 1 : 0), c.e_min, c.e_max);
    }
    const int up = c.mbits - p, dn = p - c.mbits;
    // (+ 0.0f: a value that rounds to zero comes out of at_quant with its sign, out of the streaming quantiser's magic-number rounding as
    //  +0.0 -- the same operand of a product, but the claim is byte for byte)
    const unsigned lo = pack_bf16(at_quant(v[0], up, dn, c.mant_max) + 0.0f, at_quant(v[1], up, dn, c.mant_max) + 0.0f);
    const unsigned hi = pack_bf16(at_quant(v[2], up, dn, c.mant_max) + 0.0f, at_quant(v[3], up, dn, c.mant_max) + 0.0f);
    if (m < M) {
        unsigned char* dst = reinterpret_cast<unsigned char*>(c.out_tiled) + ((m >> 4) * c.kp + (col >> 5)) * 1024 + ((col & 31) >> 3) * 256 +
                             (m & 15) * 16 + (col & 7) * 2;
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
    }
}

// ---- the attention pass ------------------------------------------------------------------------------------------------
// NTW = score tiles per wave (KW = 4: 8, 16, 32 <-> T <= 512, 1024, 2048; KW = 8: half of that), DC = D / 32.
static unsigned long long* g_attn_stamps = nullptr;     // diagnostic (-DATTN_STAMPS builds)
struct AttnArgs {
    const float* q;
    const uint16_t* kf;
    const uint16_t* vf;
    const float* mask;        // additive [M, T] or null
    float* out;
    long long M, T, NT, NPAIR;
    long long causal_off;     // >= 0: query i sees keys 0 .. i + causal_off; < 0: no causal rule
    float scale_div;          // 0: none
    int D;
    long long qsb, qsm;       // element strides of q's batch (head) and row
    long long osb, osm;       // ... of out's
    unsigned long long* stamps;   // diagnostic (-DATTN_STAMPS builds, tools/dbg/attn_stamps.py): [workgroup][8] realtime words
    int nxb, nb;              // query blocks (of a workgroup's queries) per head, heads: the work items of a launch
    const uint16_t* qfrag;    // the quantised Q fragments (attn_pack_q: rotary embedding applied); null: q is quantised here
    AttnConsumer cons;        // the out-projection's tiled bf16 operand instead of fp32 `out` (out_tiled == null: fp32)
};
#ifndef ATTN_EARLY_EXIT
#define ATTN_EARLY_EXIT 1
#endif
#ifdef ATTN_STAMPS
#define ATTN_STAMP(k) ast_[k] = __builtin_amdgcn_s_memrealtime()
#else
#define ATTN_STAMP(k)
#endif

// QG = 16-query groups per workgroup (4 waves each).  Two groups walk the same key tiles in step: the second request for
// a K / V fragment is served by the compute unit's L1 instead of the L2 (the kernel is L2-bandwidth bound at long T: every
// 16 queries stream their head's whole K and V fragments).
// KW = waves that share the keys of a query group (4, or 8 with half the strip per wave: 64 accumulator VGPRs, twice the
// waves per SIMD).
// QF: the Q fragments come packed (attn_pack_q: the rotary embedding applied on the way) instead of q itself -- a template
// argument, not a branch: the branch alone moved the allocator of the eight-key-wave variant from 127 to 130 VGPRs, one workgroup a
// compute unit instead of two.
// OT: the consumer's tiled operand as the output (at_store_consumer) -- a template argument for the same reason.
template <int NTW, int DC, int QG, bool HASMASK, int KW = 4, bool QF = false, bool OT = false>
__global__ __launch_bounds__(64 * KW * QG) void bfp_attention_kernel(const QuantArgs aq, const QuantArgs ap, const AttnArgs g) {
    constexpr int DT = DC * 2;
    constexpr float FMIN = -3.4028234663852886e38f;
    __shared__ float stat_[QG][KW][16];
    __shared__ f32x4 red_[QG][KW][DT][64];
#ifdef ATTN_STAMPS
    unsigned long long ast_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ATTN_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % KW, grp = wave_all / KW;
    float (&stat)[KW][16] = stat_[grp];
    f32x4 (&red)[KW][DT][64] = red_[grp];
    const int c16 = lane & 15, lg = lane >> 4;
    // Work items = (query block, head) in ONE grid dimension, sorted by visible keys: causal launches start the last query block of
    // EVERY head first and end on the first blocks (round 5; before, the order was heavy to light within a head, head after head:
    // the chip was never short of heavy blocks to end on -- [32, 1024, 128] 87 -> 68 us, [32, 2048, 128] 215 -> 204).
    // (The same list walked by a chip-filling persistent launch in a snake was measured too: 201 us -- and 40 more registers for
    //  the item loop, which cost the 8-key-wave variant its occupancy: not kept.)
    const int item = (int)blockIdx.x, xrank = item / g.nb;
    const long long b = item - xrank * g.nb, m0 = ((long long)(g.causal_off >= 0 ? g.nxb - 1 - xrank : xrank) * QG + grp) * 16;
    const long long qrow = min(m0 + c16, g.M - 1);

    // Q fragments: lane (query c16, g) holds d = 32 c + 8 g .. + 7; a [1,16] block = the lanes g, g ^ 1 of a chunk
    bf16x8 qf[DC];
    if constexpr (QF) {                                     // packed in front of this launch: attn_pack_q
        const uint16_t* __restrict__ qfb = g.qfrag + ((b * g.NT + (m0 >> 4)) * DC) * 512 + lane * 8;
#pragma unroll
        for (int c = 0; c < DC; ++c) qf[c] = *reinterpret_cast<const bf16x8*>(qfb + c * 512);
    } else {
        const int mb = (int)__builtin_log2f(aq.shift);
        const float* __restrict__ qp = g.q + b * g.qsb + qrow * g.qsm;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            const float4 lo = *reinterpret_cast<const float4*>(qp + 32 * c + 8 * lg);
            const float4 hi = *reinterpret_cast<const float4*>(qp + 32 * c + 8 * lg + 4);
            float bmax = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(lo.y)), fmaxf(fabsf(lo.z), fabsf(lo.w))),
                               fmaxf(fmaxf(fabsf(hi.x), fabsf(hi.y)), fmaxf(fabsf(hi.z), fabsf(hi.w))));
            bmax = at_max2_16(bmax);
            const int p = at_block_exponent_mem(bmax, aq);
            const int up = mb - p, dn = p - mb;
            uint4 pk;
            pk.x = pack_bf16(at_quant(lo.x, up, dn, aq.mant_max), at_quant(lo.y, up, dn, aq.mant_max));
            pk.y = pack_bf16(at_quant(lo.z, up, dn, aq.mant_max), at_quant(lo.w, up, dn, aq.mant_max));
            pk.z = pack_bf16(at_quant(hi.x, up, dn, aq.mant_max), at_quant(hi.y, up, dn, aq.mant_max));
            pk.w = pack_bf16(at_quant(hi.z, up, dn, aq.mant_max), at_quant(hi.w, up, dn, aq.mant_max));
            qf[c] = __builtin_bit_cast(bf16x8, pk);
        }
    }
    ATTN_STAMP(1);
    // tiles this workgroup needs: up to the horizon of its last query
    const long long kvis = g.causal_off >= 0 ? qrow + g.causal_off : g.T - 1;           // this lane's query
    long long need = g.NT;
    if (g.causal_off >= 0) need = min(g.NT, (min(m0 + 15, g.M - 1) + g.causal_off) / 16 + 1);
    const uint16_t* __restrict__ kfb = g.kf + b * g.NT * DC * 512;
    const float scale_inv = g.scale_div != 0.f ? 1.0f / g.scale_div : 0.f;
    const float* __restrict__ mrow = HASMASK ? g.mask + qrow * g.T : nullptr;

    // ---- scores of this wave's tiles, masked; row maximum.  K fragments arrive in groups of G tiles, the next group
    //      requested before this one is used, and UNCONDITIONALLY (a tile behind the horizon re-reads the last needed one:
    //      an L1 hit): a load inside a branch makes the compiler drain everything in flight at every tile -- one L2 round
    //      trip per tile was 57 % of the kernel.
    // (eight key-waves at head_dim 64: groups of two tiles like head_dim 128 -- 32 registers of prefetch instead of 64, round 6)
    constexpr int G = DC == 1 ? 8 : (DC == 2 ? ((KW == 8) ? 2 : 4) : 2), NG = NTW / G;
    f32x4 acc[NTW];
    float mx = -INFINITY;
    const long long tlast = need - 1;
    uint4 kb[2][G][DC];
    float4 mb[2][HASMASK ? G : 1];                         // (the additive mask's values ride with the K fragments)
    // (round 5: through a buffer descriptor that ends behind the last NEEDED tile -- a fragment behind the causal horizon is out of
    //  range: zeros, no cache access.  Before, such a request re-read the last needed tile.  Stamps (tools/dbg/attn_stamps.py) and three
    //  experiments put the two loops' floor elsewhere, though: the compute unit's vector-memory front end takes 16 clocks per 64-lane
    //  dwordx4 request -- hit, miss or out of range -- and 8 waves x 128 requests a loop are 16 k clocks of a 24-us workgroup at head_dim
    //  128; a deeper prefetch through private LDS-DMA rings changes nothing, fragments SHARED by the two query groups through LDS (one
    //  barrier a group) make the walk proportional to the causal horizon but the full-length workgroups slower: profiles/r05_attention.txt)
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kfb), 0, (int)(need * DC * 1024), 0x00020000);
    auto load_group = [&](int gi, uint4 (&dst)[G][DC], float4 (&mdst)[HASMASK ? G : 1]) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int tu = KW * (gi * G + j) + wave;
            const long long t = min((long long)tu, tlast);
#pragma unroll
            for (int c = 0; c < DC; ++c)
                dst[j][c] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(krs, (tu * DC + c) * 1024 + lane * 16, 0, 0));
            if (HASMASK) mdst[j] = *reinterpret_cast<const float4*>(mrow + t * 16 + 4 * lg);
        }
    };
    load_group(0, kb[0], mb[0]);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        if (ATTN_EARLY_EXIT && gi > 0 && KW * gi * G >= need) break;      // (uniform: every tile from here on lies behind the horizon)
        if (gi + 1 < NG) load_group(gi + 1, kb[(gi + 1) & 1], mb[(gi + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int i = gi * G + j;
            const long long t = KW * i + wave;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < DC; ++c)
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb[gi & 1][j][c]), qf[c], s, 0, 0, 0);
            if (t < need) {                                // (uniform over the wave)
                const long long key0 = t * 16 + 4 * lg;
                if (g.scale_div != 0.f) {                  // (the corrected quotient of at_div: 3 operations instead of ~10)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] = at_div(s[e], g.scale_div, scale_inv);
                }
                if (HASMASK) {
                    const float4 mk = mb[gi & 1][j];
                    s[0] = fmaxf(s[0] + mk.x, FMIN); s[1] = fmaxf(s[1] + mk.y, FMIN);
                    s[2] = fmaxf(s[2] + mk.z, FMIN); s[3] = fmaxf(s[3] + mk.w, FMIN);
                }
                if (t * 16 + 15 > m0 + g.causal_off && g.causal_off >= 0) {      // (uniform: a tile on the diagonal)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (key0 + e > kvis) s[e] = FMIN;
                }
                mx = fmaxf(mx, fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
            }
            acc[i] = s;
        }
    }
    ATTN_STAMP(2);
    mx = at_max4(mx);
    if (lg == 0) stat[wave][c16] = mx;
    __syncthreads();
    float row_max = stat[0][c16];
#pragma unroll
    for (int w = 1; w < KW; ++w) row_max = fmaxf(row_max, stat[w][c16]);
    __syncthreads();
    ATTN_STAMP(3);
    // ---- exponentials in place, row sum
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (KW * i + wave < need) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ex = at_exp_neg(acc[i][e] - row_max);
                acc[i][e] = ex;
                sm += ex;
            }
        }
    }
    ATTN_STAMP(4);
    sm = at_sum4(sm);
    if (lg == 0) stat[wave][c16] = sm;
    __syncthreads();
    float row_sum = (stat[0][c16] + stat[1][c16]) + (stat[2][c16] + stat[3][c16]);
    if (KW == 8) row_sum += (stat[4][c16] + stat[5][c16]) + (stat[6][c16] + stat[7][c16]);
    const float row_inv = 1.0f / row_sum;

    ATTN_STAMP(5);
    // ---- probabilities, quantised per tile row (one [1,16] block = the 4 lanes c16 + 16 g'), times V
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int mbp = (int)__builtin_log2f(ap.shift);
    const uint16_t* __restrict__ vfb = g.vf + b * DT * g.NPAIR * 512;
    // (V fragments one pair ahead and unconditional, like the K fragments; a pair behind the horizon re-reads the last one)
    // (pair KW s + w holds key tiles 2 KW s + w and 2 KW s + KW + w: the pairs with a visible key are 0 .. KW s' + min(KW - 1, r),
    //  tlast = 2 KW s' + r -- a contiguous range, so one descriptor bounds them)
    const long long s_full = tlast / (2 * KW), r_last = tlast - 2 * KW * s_full;
    const long long pairs_needed = min((long long)g.NPAIR, KW * s_full + min((long long)KW - 1, r_last) + 1);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vfb), 0, (int)(pairs_needed * DT * 1024), 0x00020000);
    uint4 vb[2][DT];
    auto load_pair = [&](int sp, uint4 (&dst)[DT]) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            dst[dt] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(vrs, ((KW * sp + wave) * DT + dt) * 1024 + lane * 16, 0, 0));
    };
    load_pair(0, vb[0]);
#pragma unroll
    for (int s = 0; s < NTW / 2; ++s) {
        if (ATTN_EARLY_EXIT && s > 0 && KW * 2 * s >= need) break;        // (uniform)
        if (s + 1 < NTW / 2) load_pair(s + 1, vb[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (KW * (2 * s) + wave < need) {                  // (uniform; tiles are needed in order)
            float pq[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * s + h;
                float pr[4];
                float bmax = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (a pair's second tile may lie behind the horizon: its strip entries are the exact zeros of a product with
                    //  out-of-range K fragments, untouched by the exponentials -- and 0 / sum = 0 without a select)
                    pr[e] = at_div(acc[i][e], row_sum, row_inv);
                    bmax = fmaxf(bmax, pr[e]);
                }
                bmax = at_max4(bmax);
                const int p = at_block_exponent_mem(bmax, ap);
                const float sc_up = __builtin_ldexpf(1.0f, mbp - p), sc_dn = __builtin_ldexpf(1.0f, p - mbp), eps_up = EPS9 * sc_up;
#pragma unroll
                for (int e = 0; e < 4; ++e) pq[4 * h + e] = at_quant_pos(pr[e], sc_up, eps_up, sc_dn, ap.mant_max);
            }
            uint4 pk;
            pk.x = pack_bf16(pq[0], pq[1]); pk.y = pack_bf16(pq[2], pq[3]);
            pk.z = pack_bf16(pq[4], pq[5]); pk.w = pack_bf16(pq[6], pq[7]);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vb[s & 1][dt]), pf, o[dt], 0, 0, 0);
        }
    }
    ATTN_STAMP(6);
    // ---- sum the four waves' partial outputs in wave order, store
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) red[wave][dt][lane] = o[dt];
    __syncthreads();
    const long long m = m0 + c16;
    for (int dt = wave; dt < DT; dt += KW) {
        f32x4 sum = red[0][dt][lane];
#pragma unroll
        for (int w = 1; w < KW; ++w) sum += red[w][dt][lane];
        if constexpr (OT) at_store_consumer(g.cons, sum, m, g.M, b * g.D + 16 * dt + 4 * lg);
        else if (m < g.M)
            *reinterpret_cast<float4*>(g.out + b * g.osb + m * g.osm + 16 * dt + 4 * lg) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    }
#ifdef ATTN_STAMPS
    ATTN_STAMP(7);
    if (g.stamps && lane == 0 && wave_all == 0) {
        unsigned long long* d = g.stamps + (long long)item * 8;
        for (int k = 0; k < 8; ++k) d[k] = ast_[k];
    }
#endif
}

// ---- the same pass for any number of keys: scores are not kept but formed twice ----------------------------------------
// A workgroup is 64 queries (4 waves x 16) that walk the key tiles together, 32 keys a step; the step's K (and, second
// time round, V) fragments come into LDS by LDS-DMA one step ahead, ONE barrier a step, and all four waves read them from
// there: each fragment leaves the L2 once per 64 queries (the resident kernel above: once per 32).  Pass 1 keeps, per
// lane, a running maximum and sum of exponentials of its scores (re-based when the maximum moves: rare after the first
// tiles) -- combined over the four lanes of a query at the end; pass 2 forms the scores again (the same MFMAs on the same
// operands: the same bits), turns them into probabilities with the final statistics, quantises, multiplies with V.  No
// cross-wave exchange at all, ~90 VGPRs (4-5 waves per SIMD against 2), any T.  The sum of exponentials is accumulated
// in a different order than in the resident kernel (and than torch's): 1e-7 relative, inside the functions' tolerance.
template <int DC, bool HASMASK, bool QF = false, bool OT = false>
__global__ __launch_bounds__(256) void bfp_attention_stream_kernel(const QuantArgs aq, const QuantArgs ap, const AttnArgs g) {
    constexpr int DT = DC * 2, KSTEP = 2 * DC * 1024, VSTEP = DT * 1024, STEP = KSTEP + VSTEP;      // bytes per 32 keys
    constexpr float FMIN = -3.4028234663852886e38f;
    using gptr_t = const __attribute__((address_space(1))) void*;
    using lptr_t = __attribute__((address_space(3))) void*;
    __shared__ Lut lut;
    __shared__ __attribute__((aligned(16))) unsigned char stage[2][STEP];
    load_lut<FMT_BFP, true>(lut);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, lg = lane >> 4;
    // (work items (query block, head) in one grid dimension, the heaviest query block of every head first: see the resident kernel)
    const int xrank = (int)blockIdx.x / g.nb;
    const long long b = (int)blockIdx.x - xrank * g.nb;
    const long long wg0 = (long long)(g.causal_off >= 0 ? g.nxb - 1 - xrank : xrank) * 64;
    const long long m0 = wg0 + 16 * wave;
    const long long qrow = min(m0 + c16, g.M - 1);
    bf16x8 qf[DC];
    if constexpr (QF) {                                     // packed in front of this launch: attn_pack_q
        const uint16_t* __restrict__ qfb = g.qfrag + ((b * g.NT + (m0 >> 4)) * DC) * 512 + lane * 8;
#pragma unroll
        for (int c = 0; c < DC; ++c) qf[c] = *reinterpret_cast<const bf16x8*>(qfb + c * 512);
    } else {
        const int mb = (int)__builtin_log2f(aq.shift);
        const float* __restrict__ qp = g.q + b * g.qsb + qrow * g.qsm;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            const float4 lo = *reinterpret_cast<const float4*>(qp + 32 * c + 8 * lg);
            const float4 hi = *reinterpret_cast<const float4*>(qp + 32 * c + 8 * lg + 4);
            float bmax = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(lo.y)), fmaxf(fabsf(lo.z), fabsf(lo.w))),
                               fmaxf(fmaxf(fabsf(hi.x), fabsf(hi.y)), fmaxf(fabsf(hi.z), fabsf(hi.w))));
            bmax = at_max2_16(bmax);
            unsigned code;
            const int p = block_param<FMT_BFP>(bmax != 0.f ? bmax : 1.0f, aq, lut, code).p;
            const int up = mb - p, dn = p - mb;
            uint4 pk;
            pk.x = pack_bf16(at_quant(lo.x, up, dn, aq.mant_max), at_quant(lo.y, up, dn, aq.mant_max));
            pk.y = pack_bf16(at_quant(lo.z, up, dn, aq.mant_max), at_quant(lo.w, up, dn, aq.mant_max));
            pk.z = pack_bf16(at_quant(hi.x, up, dn, aq.mant_max), at_quant(hi.y, up, dn, aq.mant_max));
            pk.w = pack_bf16(at_quant(hi.z, up, dn, aq.mant_max), at_quant(hi.w, up, dn, aq.mant_max));
            qf[c] = __builtin_bit_cast(bf16x8, pk);
        }
    }
    const long long kvis = g.causal_off >= 0 ? qrow + g.causal_off : g.T - 1;
    long long need = g.NT;                                  // tiles the WORKGROUP walks: up to its last query's horizon
    if (g.causal_off >= 0) need = min(g.NT, (min(wg0 + 63, g.M - 1) + g.causal_off) / 16 + 1);
    const int nsteps = (int)((need + 1) / 2);
    const unsigned char* __restrict__ kfb = reinterpret_cast<const unsigned char*>(g.kf) + b * g.NT * DC * 1024;
    const unsigned char* __restrict__ vfb = reinterpret_cast<const unsigned char*>(g.vf) + b * g.NPAIR * DT * 1024;
    const float* __restrict__ mrow = HASMASK ? g.mask + qrow * g.T : nullptr;
    const float scale_inv = g.scale_div != 0.f ? 1.0f / g.scale_div : 0.f;

    // LDS-DMA of one step: K pieces (2 DC KiB, contiguous) and, when with_v, V pieces (DT KiB, contiguous); piece p by wave p % 4
    // (an odd tile count: the last step's second tile reads DC KiB past this batch's K fragments -- the next batch's, or the
    //  V fragments that follow in the workspace; its scores are never used)
    auto dma = [&](int st, int buf, bool with_v) {
#pragma unroll
        for (int p = 0; p < 2 * DC; ++p)
            if ((p & 3) == wave)
                __builtin_amdgcn_global_load_lds((gptr_t)(kfb + (long long)st * KSTEP + p * 1024 + lane * 16),
                                                 (lptr_t)(&stage[buf][p * 1024]), 16, 0, 0);
        if (with_v) {
#pragma unroll
            for (int p = 0; p < DT; ++p)
                if ((p & 3) == wave)
                    __builtin_amdgcn_global_load_lds((gptr_t)(vfb + (long long)st * VSTEP + p * 1024 + lane * 16),
                                                     (lptr_t)(&stage[buf][KSTEP + p * 1024]), 16, 0, 0);
        }
    };
    // scores of the step's two tiles for this lane's query: s[h][e] <-> key 32 st + 16 h + 4 lg + e, masked
    auto scores = [&](int st, int buf, f32x4 (&sv)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const uint4 kv = *reinterpret_cast<const uint4*>(&stage[buf][(h * DC + c) * 1024 + lane * 16]);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kv), qf[c], s, 0, 0, 0);
            }
            const long long t = 2ll * st + h, key0 = t * 16 + 4 * lg;
            if (t < need) {
                if (g.scale_div != 0.f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] = at_div(s[e], g.scale_div, scale_inv);
                }
                if (HASMASK) {
                    const float4 mk = *reinterpret_cast<const float4*>(mrow + key0);
                    s[0] = fmaxf(s[0] + mk.x, FMIN); s[1] = fmaxf(s[1] + mk.y, FMIN);
                    s[2] = fmaxf(s[2] + mk.z, FMIN); s[3] = fmaxf(s[3] + mk.w, FMIN);
                }
                if (g.causal_off >= 0 && t * 16 + 15 > m0 + g.causal_off) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (key0 + e > kvis) s[e] = FMIN;
                }
            } else {
                s = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};       // (not a tile of this product)
            }
            sv[h] = s;
        }
    };

    // ---- pass 1: running maximum and sum of exponentials per lane
    float m_run = -INFINITY, l_run = 0.f;
    dma(0, 0, false);
    for (int st = 0; st < nsteps; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < nsteps) dma(st + 1, (st + 1) & 1, false);
        f32x4 sv[2];
        scores(st, st & 1, sv);
        const float tmax = fmaxf(fmaxf(fmaxf(sv[0][0], sv[0][1]), fmaxf(sv[0][2], sv[0][3])),
                                 fmaxf(fmaxf(sv[1][0], sv[1][1]), fmaxf(sv[1][2], sv[1][3])));
        if (__any(tmax > m_run)) {                          // re-base (exp(-inf) = 0 takes care of the first tile)
            const float m_new = fmaxf(m_run, tmax);
            l_run = m_new == -INFINITY ? 0.f : l_run * at_exp_neg(m_run - m_new);
            m_run = m_new;
        }
        float add = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) add += sv[h][e] == -INFINITY ? 0.f : at_exp_neg(sv[h][e] - m_run);
        l_run += add;
    }
    const float row_max = at_max4(m_run);
    const float row_sum = at_sum4(m_run == -INFINITY ? 0.f : l_run * at_exp_neg(m_run - row_max));
    const float row_inv = 1.0f / row_sum;
    __syncthreads();                                        // (every wave is out of the last step's buffer)

    // ---- pass 2: the scores again, probabilities, quantised, times V
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int mbp = (int)__builtin_log2f(ap.shift);
    dma(0, 0, true);
    for (int st = 0; st < nsteps; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < nsteps) dma(st + 1, (st + 1) & 1, true);
        f32x4 sv[2];
        scores(st, st & 1, sv);
        float pq[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float pr[4];
            float bmax = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pr[e] = sv[h][e] == -INFINITY ? 0.f : at_div(at_exp_neg(sv[h][e] - row_max), row_sum, row_inv);
                bmax = fmaxf(bmax, pr[e]);
            }
            bmax = at_max4(bmax);
            const int p = at_block_exponent(bmax, ap, lut);
            const float sc_up = __builtin_ldexpf(1.0f, mbp - p), sc_dn = __builtin_ldexpf(1.0f, p - mbp), eps_up = EPS9 * sc_up;
#pragma unroll
            for (int e = 0; e < 4; ++e) pq[4 * h + e] = at_quant_pos(pr[e], sc_up, eps_up, sc_dn, ap.mant_max);
        }
        uint4 pk;
        pk.x = pack_bf16(pq[0], pq[1]); pk.y = pack_bf16(pq[2], pq[3]);
        pk.z = pack_bf16(pq[4], pq[5]); pk.w = pack_bf16(pq[6], pq[7]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const uint4 vv = *reinterpret_cast<const uint4*>(&stage[st & 1][KSTEP + dt * 1024 + lane * 16]);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[dt], 0, 0, 0);
        }
    }
    const long long m = m0 + c16;
    if constexpr (OT) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) at_store_consumer(g.cons, o[dt], m, g.M, b * g.D + 16 * dt + 4 * lg);
    } else if (m < g.M) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            *reinterpret_cast<float4*>(g.out + b * g.osb + m * g.osm + 16 * dt + 4 * lg) = make_float4(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
    }
}

static int g_attention_kernel = 0;          // 0: by size, 1: resident scores with 4 key-waves (T <= 2048), 2: streaming,
                                            // 3: resident scores with 8 key-waves (head_dim <= 64)   (A/B runs, tests)
static int g_attention_qpack = getenv("MI355Q_ATTN_QPACK") ? atoi(getenv("MI355Q_ATTN_QPACK")) : 1;
int attention_set_qpack(int on) {
    const int prev = g_attention_qpack;
    if (on >= 0 && on <= 2) g_attention_qpack = on;
    return prev;
}
int attention_set_kernel(int which) {
    const int prev = g_attention_kernel;
    if (which >= 0 && which <= 3) g_attention_kernel = which;
    return prev;
}

size_t attention_workspace_bytes(long long B, long long T, long long D) {
    const long long NT = (T + 15) / 16, NPAIR = ((T + 255) / 256) * 8;            // (the widest pair grouping)
    // (K fragments, V fragments, and -- rotary embedding on load -- the Q fragments of as many queries)
    return (size_t)B * (size_t)(2 * NT * (D / 32) * 1024 + (D / 16) * NPAIR * 1024) + 256;
}

int launch_bfp_attention(const QuantArgs& aq, const QuantArgs& ak, const QuantArgs& ap, const QuantArgs& av, const float* q,
                         const float* k, const float* v, const float* mask, float* out, void* workspace, long long B,
                         long long M, long long T, long long D, long long causal_off, float scale_div, hipStream_t st,
                         const long long* strides, const float* rope_cos, const float* rope_sin, const long long* rope_pos,
                         long long rope_rows, int rope_heads, uint16_t* out_tiled, const QuantArgs* ao, float q_scale) {
    // (q_scale: q multiplied on its way into the Q fragments -- served by the pack launch only, so the fragments must be packable)
    if (q_scale != 0.f && (rope_cos || (D != 64 && D != 128) || M > T)) return MI355Q_E_UNSUPPORTED;
    // (the consumer's operand: rows = queries, columns = head x D in head order -- B is the head count of ONE batch element)
    if (out_tiled && (!ao || (B * D) % 32 != 0)) return MI355Q_E_BADARG;
    if (out_tiled && (mask || (D != 64 && D != 128))) return MI355Q_E_UNSUPPORTED;        // (the flavours that are built: OT above)
    const AttnConsumer cons{out_tiled, out_tiled ? B * D / 32 : 0, ao ? ao->mant_max : 0.f, ao ? (int)__builtin_log2f(ao->shift) : 0,
                            ao ? ao->e_min : 0, ao ? ao->e_max : 0};
    // (the rotary embedding on load: q and k rows are the same positions, whole [1,16] blocks in each half of the head -- see RopeIn)
    if (rope_cos && (M != T || (D != 64 && D != 128) || !rope_sin || !rope_pos || rope_rows < 1 || rope_heads < 1 || B % rope_heads ||
                     rope_rows * D >= (1ll << 31) || T >= (1ll << 27)))
        return MI355Q_E_UNSUPPORTED;
    const RopeIn rope{rope_cos, rope_sin, rope_pos, rope_rows, rope_heads};
    if (D > AT_MAX_D || D % 32 != 0 || T % 16 != 0 || (mask && T % 4 != 0)) return MI355Q_E_UNSUPPORTED;
    const bool stream = g_attention_kernel == 2 || ((g_attention_kernel == 0 || g_attention_kernel > 3) && T > AT_MAX_T);
    if (!stream && T > AT_MAX_T) return MI355Q_E_UNSUPPORTED;
    // Q fragments packed in front of the kernels (attn_pack_q): always with the rotary embedding; without it where it pays -- on the
    // resident kernels from 64 queries up (the area holds NT tiles a head: M <= T) at head_dim 128, and at head_dim 64 where the
    // eight-key-wave variant runs (T > 1024: every one of EIGHT waves formed all Q fragments there).  Pack + attention, us, fragments
    // packed / q quantised in the kernels (profiles/r06_attention_qpack.jsonl, r06_attention_kw8_qf.jsonl): [32, 2048, 128] 156 / 169,
    // [32, 1024, 128] 64 / 69, [32, 2048, 64] 106 / 117-121, [12, 2048, 64] 51 / 57, [32, 1536, 64] 73 / 80; NOT at head_dim 64 with
    // four key-waves ([12, 1024, 64] 37 / 35) nor in the streaming kernel, whose waves each own their queries ([32, 4096, 128] 542 / 526).
    // (The eight-key-wave flavour that loads fragments needed 130 VGPRs -- one workgroup a compute unit instead of two, 134 us -- until
    //  its K prefetch went from groups of four tiles to groups of two: 122.)
    // attention_set_qpack(0): never without the rotary embedding (A/B runs, tests); (2): wherever the fragments fit (tests).
    const bool kw8_auto = !stream && D <= 64 && g_attention_kernel == 0 && T > 1024;
    const bool qpack = rope_cos || q_scale != 0.f || (g_attention_qpack && M >= 64 && M <= T &&
                                    (g_attention_qpack == 2 ? (D == 64 || D == 128) : (!stream && (D == 128 || (D == 64 && kw8_auto)))));
    // eight key-waves per query group (half the score strip per wave: 127 VGPRs, twice the waves per SIMD) for head_dim <=
    // 64 and long rows: 65 vs 78 us at 12 x 2048 x 64, 135 vs 167 us at 32 x 2048 x 64; no difference at 1024 keys
    const bool kw8 = !stream && D <= 64 && (g_attention_kernel == 3 || kw8_auto);
    const int kw = stream ? 1 : (kw8 ? 8 : 4), pg = kw == 8 ? 8 : 4;
    const long long NT = T / 16, NPAIR = ((T + 32 * pg - 1) / (32 * pg)) * pg;
    uint16_t* kf = static_cast<uint16_t*>(workspace);
    uint16_t* vf = kf + (size_t)B * NT * (D / 32) * 512;
    const int per = 256 / (int)D;
    const int kblocks = (int)((NT + per - 1) / per);
    // strides: {q batch, q row, k batch, k row, v batch, v row, out batch, out row} in elements (innermost stride 1);
    // null = contiguous
    const long long qsb = strides ? strides[0] : M * D, qsm = strides ? strides[1] : D;
    const long long ksb = strides ? strides[2] : T * D, kst = strides ? strides[3] : D;
    const long long vsb = strides ? strides[4] : T * D, vst = strides ? strides[5] : D;
    const long long osb = strides ? strides[6] : M * D, osm = strides ? strides[7] : D;
    uint16_t* qfrag = vf + (size_t)B * (D / 16) * NPAIR * 512;
    const int vblocks = (int)(NPAIR / pg), qpw = 256 / (int)(D / 16), qblocks = qpack ? (int)(((M + 15) / 16 * 16 + qpw - 1) / qpw) : 0;
    if (rope_cos)
        hipLaunchKernelGGL(attn_pack_kv_kernel<true>, dim3((unsigned)(kblocks + vblocks + qblocks), (unsigned)B), 256, 0, st, ak, av, k, v, kf, vf,
                           T, (int)D, NT, NPAIR, kblocks, ksb, kst, vsb, vst, kw, rope, aq, q, qfrag, vblocks, qsb, qsm, M, q_scale);
    else
        hipLaunchKernelGGL(attn_pack_kv_kernel<false>, dim3((unsigned)(kblocks + vblocks + qblocks), (unsigned)B), 256, 0, st, ak, av, k, v, kf, vf,
                           T, (int)D, NT, NPAIR, kblocks, ksb, kst, vsb, vst, kw, rope, aq, q, qfrag, vblocks, qsb, qsm, M, q_scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    AttnArgs g{q, kf, vf, mask, out, M, T, NT, NPAIR, causal_off, scale_div, (int)D, qsb, qsm, osb, osm, g_attn_stamps, 0, 0, qpack ? qfrag : nullptr, cons};
    if (stream) {
        g.nxb = (int)((M + 63) / 64);
        g.nb = (int)B;
        const dim3 sgrid((unsigned)((long long)g.nxb * B));
#define MI355Q_ATTN_S1(DC_, M_, QF_, OT_) hipLaunchKernelGGL((bfp_attention_stream_kernel<DC_, M_, QF_, OT_>), sgrid, 256, 0, st, aq, ap, g)
#define MI355Q_ATTN_S(DC_)                                                                                     \
    if (mask) { if (g.qfrag) MI355Q_ATTN_S1(DC_, true, (DC_) % 2 == 0, false); else MI355Q_ATTN_S1(DC_, true, false, false); }          \
    else if (g.cons.out_tiled) { if (g.qfrag) MI355Q_ATTN_S1(DC_, false, (DC_) % 2 == 0, (DC_) % 2 == 0); else MI355Q_ATTN_S1(DC_, false, false, (DC_) % 2 == 0); } \
    else { if (g.qfrag) MI355Q_ATTN_S1(DC_, false, (DC_) % 2 == 0, false); else MI355Q_ATTN_S1(DC_, false, false, false); }
        switch (D / 32) {
            case 1: MI355Q_ATTN_S(1) break;
            case 2: MI355Q_ATTN_S(2) break;
            case 3: MI355Q_ATTN_S(3) break;
            default: MI355Q_ATTN_S(4) break;
        }
#undef MI355Q_ATTN_S
#undef MI355Q_ATTN_S1
        return (int)hipGetLastError();
    }
    // the resident kernel's launch: one workgroup per work item (query block, head), heaviest items first (see the kernel)
    g.nb = (int)B;
#define MI355Q_ATTN_GO(QPB_, ...)                                                                                         \
    {                                                                                                                     \
        g.nxb = (int)((M + (QPB_) - 1) / (QPB_));                                                                         \
        hipLaunchKernelGGL((bfp_attention_kernel<__VA_ARGS__>), dim3((unsigned)((long long)g.nxb * B)), 512, 0, st, aq, ap, g); \
    }
    if (kw8) {
#define MI355Q_ATTN_PICK(QPB_, NTW_, DC_, QG_, KW_)                                                                                       \
    if (mask) { if (g.qfrag) MI355Q_ATTN_GO(QPB_, NTW_, DC_, QG_, true, KW_, (DC_) % 2 == 0, false) else MI355Q_ATTN_GO(QPB_, NTW_, DC_, QG_, true, KW_, false, false) } \
    else if (g.cons.out_tiled) { if (g.qfrag) MI355Q_ATTN_GO(QPB_, NTW_, DC_, QG_, false, KW_, (DC_) % 2 == 0, (DC_) % 2 == 0) else MI355Q_ATTN_GO(QPB_, NTW_, DC_, QG_, false, KW_, false, (DC_) % 2 == 0) } \
    else { if (g.qfrag) MI355Q_ATTN_GO(QPB_, NTW_, DC_, QG_, false, KW_, (DC_) % 2 == 0, false) else MI355Q_ATTN_GO(QPB_, NTW_, DC_, QG_, false, KW_, false, false) }
#define MI355Q_ATTN8(NTW_, DC_) MI355Q_ATTN_PICK(16, NTW_, DC_, 1, 8)
        const int ntw8 = T <= 1024 ? 8 : 16;
        if (ntw8 == 8) { if (D == 32) { MI355Q_ATTN8(8, 1); } else { MI355Q_ATTN8(8, 2); } }
        else { if (D == 32) { MI355Q_ATTN8(16, 1); } else { MI355Q_ATTN8(16, 2); } }
#undef MI355Q_ATTN8
        return (int)hipGetLastError();
    }
    // two 16-query groups per workgroup (measured at T = 2048: 70 vs 101 us at 12 heads x 64, 235 vs 342 us at 32 x 128)
    const int ntw = T <= 512 ? 8 : (T <= 1024 ? 16 : 32);
#define MI355Q_ATTN(NTW_, DC_) MI355Q_ATTN_PICK(32, NTW_, DC_, 2, 4)
#define MI355Q_ATTN_D(NTW_)                                     \
    switch (D / 32) {                                          \
        case 1: MI355Q_ATTN(NTW_, 1); break;                   \
        case 2: MI355Q_ATTN(NTW_, 2); break;                   \
        case 3: MI355Q_ATTN(NTW_, 3); break;                   \
        default: MI355Q_ATTN(NTW_, 4); break;                  \
    }
    if (ntw == 8) { MI355Q_ATTN_D(8) }
    else if (ntw == 16) { MI355Q_ATTN_D(16) }
    else { MI355Q_ATTN_D(32) }
#undef MI355Q_ATTN_D
#undef MI355Q_ATTN
#undef MI355Q_ATTN_GO
#undef MI355Q_ATTN_PICK
    return (int)hipGetLastError();
}

}  // namespace mi355q

// diagnostic hook, not part of include/mi355q.h: the buffer ([workgroups][8] 64-bit words) a -DATTN_STAMPS build fills
extern "C" __attribute__((visibility("default"))) void mi355q_debug_attn_stamps(void* buf) { mi355q::g_attn_stamps = static_cast<unsigned long long*>(buf); }
