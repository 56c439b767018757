// mi355q_mx.hip -- W4A4 / W5A5 block_fp Linear on the MX scaled matrix instruction (gfx950), round 5.
//
//     y[m,n] = sum_k x_q[m,k] * w_q[n,k]  (+ bias[n]),   x_q, w_q block_fp values of <= 5 bits, [1,16] blocks along k
// Reference path: quantized_modules/linear.py:59-76 (F.linear on the fake-quantised operands) at the widths of
// experiments/emnlp/configs/quantization/bfp_4bit.toml and the section-4.4 search (configs/search/opt_1.3b_sst2.toml:24-37).
//
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies FP6 (e2m3) operands with one E8M0 scale per 32 values at twice the int8 rate per
// unit of K (profiles/r03_mx_rate.txt: 8.05 against 4.28 POPS from registers).  A block_fp mantissa of <= 4 bits is exact in
// e2m3 (4 significant bits), the block's shared exponent goes into the scale: NO row alignment, no exception lists -- every
// 32-group carries its own exponent.  What the format cannot hold is a 32-group whose two [1,16] blocks lie more than 3 (W4) /
// 2 (W5) exponents apart: the quantiser (mi355q_quant.hip, MX flavour) raises a flag word then, and this launch forms the
// product from the fp32 tensors instead -- quantising x in registers (the reference's arithmetic, mi355q_quant_dev.h) on its way
// into bf16 MFMAs: exact, slow, decided on the device and uniform over the grid; callers move such a layer off this route.
// Products of the codes are exact, the accumulation is fp32 (the reference's own: F.linear in fp32).
//
// Kernel: 256 x 256 tile, 8 waves of 128 x 64 as in mi355q_gemm_v9.hip, K-step 128; a stage = x codes (16 + 8 KiB), w codes
// (16 + 8 KiB), x and w scales (1 KiB each): 50 KiB, ring of three filled by buffer_load ... lds (inline assembly, counted
// vmcnt, raw barriers: one K-step of LDS-DMA stays in flight across the barrier).  Fragments: ds_read_b128 + ds_read_b64 per
// lane (its 32 values of one row), scales one dword per four fragments (op_sel picks the byte).  Fragment reads and MFMAs are
// compiler-scheduled inside a K-step here (v1): the hand-counted pipeline of the int8 kernel is the next step.
// Roofline: MX FP6 MFMA (2x the int8 peak), 2*M*N*K ops.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_gemm_v2.h"
#include "mi355q_quant_dev.h"

namespace mi355q {

typedef int mx_i32x8 __attribute__((ext_vector_type(8)));
typedef int mx_i32x4 __attribute__((ext_vector_type(4)));
typedef int mx_i32x2 __attribute__((ext_vector_type(2)));
typedef float mx_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mx_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int MX_NT = 512;
constexpr int MX_X16 = 0, MX_X8 = 16384, MX_W16 = 24576, MX_W8 = 40960, MX_SX = 49152, MX_SW = 50176, MX_STAGE = 51200, MX_NS = 3;
constexpr int MX_RING = MX_NS * MX_STAGE, MX_BIAS = MX_RING, MX_LDS = MX_BIAS + 1024 + sizeof(Lut);

#define MX_BLDS16(vo, rs, so, lds) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so), "s"(lds) : "memory")
template <int N> __device__ __forceinline__ void mx_waitv() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ mx_i32x4 mx_desc(const void* base, long long bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    return mx_i32x4{(int)(unsigned)b, (int)(unsigned)(b >> 32), (int)(bytes > 0x7fffffffll ? 0x7fffffffll : bytes), 0x00020000};
}

template <int SEL>
__device__ __forceinline__ mx_f32x4 mx_mma(const mx_i32x8& fw, const mx_i32x8& fx, const mx_f32x4& c, int sw, int sx) {
    // (cbsz = blgp = 2: FP6 e2m3 both; op_sel picks the scale byte: w fragment SEL & 3, x fragment SEL >> 2)
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw, fx, c, 2, 2, SEL & 3, sw, SEL >> 2, sx);
}

__global__ __launch_bounds__(MX_NT, 1) void mx_gemm_kernel(const MxGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mx_smem[];
    unsigned char* const ring = mx_smem;
    float* const bst = reinterpret_cast<float*>(mx_smem + MX_BIAS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, l16 = lane & 15, lq = lane >> 4;
    const int Mi = (int)a.M, Ni = (int)a.N;
    const int tiles_m = (Mi + 255) >> 8, tiles_n = (Ni + 255) >> 8, nwg = tiles_m * tiles_n;
    int pid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int GM = 4, in_group = GM * tiles_n, group_id = pid / in_group, first_m = group_id * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (pid % in_group) % gsz, tn = (pid % in_group) / gsz;
    const int m0 = tm * 256, n0 = tn * 256;
    const int kp = (int)(a.K >> 7), nsteps = kp;
    const int ring_lds = (int)(size_t)(lptr_t)ring;
    const int bad = a.bad[0] | a.bad[1];                       // (scalar loads: in flight while the first stages are requested)

    // bias slice (columns past N: zero)
#pragma unroll
    for (int q = tid; q < 256; q += MX_NT) bst[q] = (a.bias && n0 + q < Ni) ? a.bias[n0 + q] : 0.f;

    // ---- the operand stream.  Per K-step this wave stages: x plane-16 pieces wave, wave + 8; x plane-8 piece wave; the same of
    //      w; waves 0 / 1 also the x / w scales (1 KiB each: four 64-row blocks of 256 bytes, kp * 256 bytes apart in memory)
    const long long prow = (long long)kp * 1024;               // one piece row of a code plane
    const int v16a = wave * (int)prow + lane * 16, v16b = (wave + 8) * (int)prow + lane * 16, v8 = wave * (int)prow + lane * 16;
    const int vsc = (lane >> 4) * (kp * 256) + (lane & 15) * 16;
    const mx_i32x4 dx16 = mx_desc(a.x16 + (long long)(m0 >> 4) * prow, 16 * prow), dx8 = mx_desc(a.x8 + (long long)(m0 >> 5) * prow, 8 * prow);
    const mx_i32x4 dw16 = mx_desc(a.w16 + (long long)(n0 >> 4) * prow, 16 * prow), dw8 = mx_desc(a.w8 + (long long)(n0 >> 5) * prow, 8 * prow);
    const mx_i32x4 dxs = mx_desc(a.xs + (long long)(m0 >> 6) * kp * 256, 4ll * kp * 256), dws = mx_desc(a.ws + (long long)(n0 >> 6) * kp * 256, 4ll * kp * 256);
    // piece q (0..5: codes, 6: this wave's scale piece if it has one) of K-step `step` into the stage at byte offset `so`
    // (a step past the end is requested through descriptors of zero bytes: no traffic, the counted waits stay uniform)
    auto piece = [&](int q, int step, int so) {
        const bool more = step < nsteps;
        const mx_i32x4 z{0, 0, 0, 0x00020000};
        const int s1k = step * 1024, dst = ring_lds + so;
        if (q == 0) MX_BLDS16(v16a, more ? dx16 : z, s1k, dst + MX_X16 + wave * 1024);
        else if (q == 1) MX_BLDS16(v16b, more ? dx16 : z, s1k, dst + MX_X16 + (wave + 8) * 1024);
        else if (q == 2) MX_BLDS16(v8, more ? dx8 : z, s1k, dst + MX_X8 + wave * 1024);
        else if (q == 3) MX_BLDS16(v16a, more ? dw16 : z, s1k, dst + MX_W16 + wave * 1024);
        else if (q == 4) MX_BLDS16(v16b, more ? dw16 : z, s1k, dst + MX_W16 + (wave + 8) * 1024);
        else if (q == 5) MX_BLDS16(v8, more ? dw8 : z, s1k, dst + MX_W8 + wave * 1024);
        else if (wave == 0) MX_BLDS16(vsc, more ? dxs : z, step * 256, dst + MX_SX);
        else if (wave == 1) MX_BLDS16(vsc, more ? dws : z, step * 256, dst + MX_SW);
    };
    auto stage = [&](int step, int so) {
#pragma unroll
        for (int q = 0; q < 7; ++q) piece(q, step, so);
    };
    stage(0, 0);
    stage(1, MX_STAGE);

    mx_f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mx_f32x4{0.f, 0.f, 0.f, 0.f};

    // lane-constant parts of the fragment addresses (bytes inside a stage)
    const int ax16 = MX_X16 + wm * 8192 + lq * 256 + l16 * 16, ax8 = MX_X8 + wm * 4096 + lq * 128 + l16 * 8;
    const int aw16 = MX_W16 + wn * 4096 + lq * 256 + l16 * 16, aw8 = MX_W8 + wn * 2048 + lq * 128 + l16 * 8;
    const int asx = MX_SX + wm * 512 + lq * 64 + l16 * 4, asw = MX_SW + wn * 256 + lq * 64 + l16 * 4;
    auto frag = [&](const unsigned char* st, int o16, int o8) {
        const mx_i32x4 lo = *reinterpret_cast<const mx_i32x4*>(st + o16);
        const mx_i32x2 hi = *reinterpret_cast<const mx_i32x2*>(st + o8);
        return mx_i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], 0, 0};
    };

    if (!bad) {
        int cur = 0;
        for (int t = 0; t < nsteps; ++t) {
            // everything but this wave's pieces of step t + 1 has landed (waves 0 / 1 carry one piece more a step)
            if (wave < 2) mx_waitv<7>(); else mx_waitv<6>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // step t + 2 goes into the slot step t - 1 was read from (every wave is past its reads: they fed MFMAs already
            // issued), one piece per MFMA group
            const int nslot = (cur == 0 ? 2 : cur - 1) * MX_STAGE;
            const unsigned char* st = ring + cur * MX_STAGE;
            const int swv = *reinterpret_cast<const int*>(st + asw);
            const int sx0 = *reinterpret_cast<const int*>(st + asx), sx1 = *reinterpret_cast<const int*>(st + asx + 256);
            mx_i32x8 fw[4], fx[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) fw[j] = frag(st, aw16 + j * 1024, aw8 + (j >> 1) * 1024 + (j & 1) * 512);
            fx[0] = frag(st, ax16, ax8);
            fx[1] = frag(st, ax16 + 1024, ax8 + 512);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // x fragment i + 2 is read while fragment i's MFMAs issue (a window of three)
                if (i + 2 < 8) fx[(i + 2) % 3] = frag(st, ax16 + (i + 2) * 1024, ax8 + ((i + 2) >> 1) * 1024 + ((i + 2) & 1) * 512);
                if (i < 7) piece(i, t + 2, nslot);
                const int sx = i < 4 ? sx0 : sx1;
                const mx_i32x8& f = fx[i % 3];
#define MX_ROW(I)                                                                                                      \
                acc[i][0] = mx_mma<0 + 4 * (I)>(fw[0], f, acc[i][0], swv, sx); acc[i][1] = mx_mma<1 + 4 * (I)>(fw[1], f, acc[i][1], swv, sx); \
                acc[i][2] = mx_mma<2 + 4 * (I)>(fw[2], f, acc[i][2], swv, sx); acc[i][3] = mx_mma<3 + 4 * (I)>(fw[3], f, acc[i][3], swv, sx);
                if ((i & 3) == 0) { MX_ROW(0) } else if ((i & 3) == 1) { MX_ROW(1) } else if ((i & 3) == 2) { MX_ROW(2) } else { MX_ROW(3) }
#undef MX_ROW
                __builtin_amdgcn_sched_barrier(0);
            }
            cur = cur == 2 ? 0 : cur + 1;
        }
        mx_waitv<0>();
    } else {
        // ---- the exact route (some 32-group of x or w does not fit the format): the product from the fp32 tensors.  Lane
        //      (row l16, 8 values lq) of a 16 x 32 fragment: x quantised in registers -- a [1,16] block is two lanes' values --,
        //      w (already fake-quantised) cast; both exact in bf16, v_mfma_f32_16x16x32_bf16, fp32 accumulation.
        mx_waitv<0>();
        __syncthreads();
        Lut& lut = *reinterpret_cast<Lut*>(mx_smem + MX_BIAS + 1024);
        load_lut<FMT_BFP>(lut);
        const QuantArgs& qa = a.qx;
        for (long long k0 = 0; k0 < a.K; k0 += 32) {
            mx_bf16x8 fw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long n = min((long long)n0 + wn * 64 + j * 16 + l16, a.N - 1);
                const float4 p0 = *reinterpret_cast<const float4*>(a.wf + n * a.K + k0 + lq * 8), p1 = *reinterpret_cast<const float4*>(a.wf + n * a.K + k0 + lq * 8 + 4);
                fw[j] = mx_bf16x8{(__bf16)p0.x, (__bf16)p0.y, (__bf16)p0.z, (__bf16)p0.w, (__bf16)p1.x, (__bf16)p1.y, (__bf16)p1.z, (__bf16)p1.w};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long long m = min((long long)m0 + wm * 128 + i * 16 + l16, a.M - 1);
                const float4 p0 = *reinterpret_cast<const float4*>(a.xf + m * a.K + k0 + lq * 8), p1 = *reinterpret_cast<const float4*>(a.xf + m * a.K + k0 + lq * 8 + 4);
                float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                float bm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) bm = fmaxf(bm, fabsf(v[e]));
                bm = fmaxf(bm, __shfl_xor(bm, 16));             // the block's other eight values: lane lq ^ 1
                unsigned code;
                const BlockParam bp = block_param<FMT_BFP>(bm > 0.f ? bm : 1.0f, qa, lut, code);
                mx_bf16x8 fx;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int mant;
                    const float qv = quant_elem<FMT_BFP>(v[e], bp, qa, lut, mant);
                    fx[e] = (__bf16)(bm > 0.f ? qv : v[e] + 0.0f);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fx, acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: the lane holds, for tile (i, j), row wm * 128 + 16 i + l16 and the four columns wn * 64 + 16 j + 4 lq + 0..3
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.y) | (uintptr_t)(a.ldy * 4)) & 15) == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long row = (long long)m0 + wm * 128 + i * 16 + l16;
        if (row >= a.M) continue;
        float* yrow = a.y + row * a.ldy + n0 + wn * 64 + lq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cl = wn * 64 + j * 16 + lq * 4, col = n0 + cl;
            const mx_f32x4 bv = *reinterpret_cast<const mx_f32x4*>(&bst[cl]);
            mx_f32x4 val;
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = acc[i][j][r] + bv[r];
            if (vec_ok && col + 3 < Ni) {
                *reinterpret_cast<mx_f32x4*>(yrow + j * 16) = val;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < Ni) yrow[j * 16 + r] = val[r];
            }
        }
    }
}

int launch_mx_gemm(const MxGemmArgs& a, hipStream_t st) {
    static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&mx_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS) == hipSuccess;
    if (!ready) return (int)hipErrorInvalidValue;
    const unsigned grid = (unsigned)(((a.M + 255) / 256) * ((a.N + 255) / 256));
    hipLaunchKernelGGL(mx_gemm_kernel, grid, MX_NT, MX_LDS, st, a);
    return (int)hipGetLastError();
}

}  // namespace mi355q
