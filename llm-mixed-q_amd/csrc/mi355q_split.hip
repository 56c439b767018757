// mi355q_split.hip -- an fp32 operand as THREE bf16 parts, in the tile order of the bf16 tile GEMM: the layers the reference leaves
// unquantised (the language-model head: nn.Linear in fp32, models/llama_quantized/modeling_llama.py:772,866,
// models/opt_quantized/modeling_opt.py:942-944) as an fp32-equivalent product on the bf16 MFMA instead of a vendor fp32 GEMM.
//
//     a = h + m + l,   h = bf16(a), m = bf16(a - h), l = bf16(a - h - m)        (both differences exact in fp32: 24 significand bits
//                                                                               in three parts of 8)
//     x . w^T = sum over the SIX part pairs of weight >= 2^-18:  (m,m) (l,h) (h,l) (m,h) (h,m) (h,h)   -- the three dropped pairs
//               (m,l) (l,m) (l,l) weigh <= 2^-26 of a product, below the rounding of the fp32 accumulation itself
//
// The six pairs lie SIDE BY SIDE along the contraction: the left operand becomes [rows, 6 K] = [m | l | h | m | h | h], the right
// one [m | h | l | h | m | h], and ONE launch of mi355q_bf16_gemm_tiled over K' = 6 K adds them in fp32, smallest terms first.  Every
// part product is exact in fp32 (8 x 8 significand bits).  Measured at [2048, 4096] x [32000, 4096]^T against an fp64 product:
// mean error 3.4e-7 of the mean magnitude (the vendor library's fp32 GEMM: 1.0e-6) in 2.23 ms (3.84): profiles/r06_lm_head_split.json.
//
// Layout written: mi355q_quant.hip's bf16 tile order -- 1-KiB pieces of 16 rows x 32 values, [8-value group 0..3][row 0..15][16 bytes]
// inside; piece (row / 16, column / 32) at ((row / 16) * (6 K / 32) + column / 32) * 1024.  One wave per source piece position: lane
// (row = lane % 16, group = lane / 16) reads its 8 values (32 bytes) and writes 16 bytes into each of the six part pieces -- every
// store instruction of a wave covers one whole piece.  Rows behind `rows` up to the operand's 128-row padding are written as zeros.
// Bound: HBM, 4 B read + 12 B written per element; the weights' operand is built once per checkpoint.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_internal.h"
#include "mi355q_quant_dev.h"

namespace mi355q {

// part index (0 = h, 1 = m, 2 = l) of each of the six column segments: ROLE 0 = left operand, 1 = right operand
template <int ROLE>
__global__ __launch_bounds__(256) void fp32_split_tile_kernel(const float* __restrict__ x, unsigned char* __restrict__ yt, long long rows,
                                                              long long K) {
    constexpr int order[6] = {1, ROLE ? 0 : 2, ROLE ? 2 : 0, ROLE ? 0 : 1, ROLE ? 1 : 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long kp = K >> 5, kp6 = 6 * kp, rows_pad = (rows + 127) / 128 * 128;
    const long long pieces = (rows_pad >> 4) * kp;
    const int r = lane & 15, g = lane >> 4;
    for (long long p = (long long)blockIdx.x * 4 + wave; p < pieces; p += (long long)gridDim.x * 4) {
        const long long pr = p / kp, pc = p - pr * kp, row = pr * 16 + r;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (row < rows) {
            const float4* src = reinterpret_cast<const float4*>(x + row * K + pc * 32 + g * 8);
            const float4 a = src[0], b = src[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        unsigned part[3][4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float a0 = v[e], a1 = v[e + 1];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const unsigned pk = pack_bf16(a0, a1);
                part[s][e >> 1] = pk;
                a0 -= __uint_as_float(pk << 16);                  // (exact: the part shares a's leading bits)
                a1 -= __uint_as_float(pk & 0xFFFF0000u);
            }
        }
        unsigned char* dst = yt + (pr * kp6 + pc) * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < 6; ++s)
            *reinterpret_cast<uint4*>(dst + s * kp * 1024) = make_uint4(part[order[s]][0], part[order[s]][1], part[order[s]][2], part[order[s]][3]);
    }
}

int launch_fp32_split_tile(const float* x, uint16_t* yt, long long rows, long long K, int role, hipStream_t st) {
    const long long pieces = ((rows + 127) / 128 * 128 >> 4) * (K >> 5);
    long long grid = (pieces + 3) / 4;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    if (role) hipLaunchKernelGGL(fp32_split_tile_kernel<1>, (unsigned)grid, 256, 0, st, x, reinterpret_cast<unsigned char*>(yt), rows, K);
    else hipLaunchKernelGGL(fp32_split_tile_kernel<0>, (unsigned)grid, 256, 0, st, x, reinterpret_cast<unsigned char*>(yt), rows, K);
    return (int)hipGetLastError();
}

}  // namespace mi355q
