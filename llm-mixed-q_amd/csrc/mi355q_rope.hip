// mi355q_rope.hip -- rotary position embedding of the reference's quantised Llama attention
// (quantized_functions/rotary_positional_encoding.py:59-82 and the six sibling functions, callers
// models/llama_quantized/modeling_llama.py:289-299):
//     cos = Q(cos)[position_ids], sin = Q(sin)[position_ids]                  (tables quantised by the caller: small)
//     q_embed = q * cos + rotate_half(q) * sin,   rotate_half(x) = cat(-x[d/2:], x[:d/2])      (the same for k)
// as ONE launch for q and k instead of ten elementwise kernels and two gathers: every element is read once and written
// once, position lookup included.  Arithmetic as the reference's fp32 ops: two products rounded to fp32, then their sum
// (no fused multiply-add).  q / k may be strided views ([B, H, T, D] logical, innermost stride 1: the transposed view the
// models hand over); the outputs are contiguous [B, H, T, D].  Bound: HBM, 8 B per element.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_internal.h"

namespace mi355q {

// a * b + c * d with both products rounded to fp32 before the sum, like the reference's separate ops: hipcc contracts
// a * b + c into an fma by default (the __f*_rn intrinsics included); the empty asm makes each product opaque to that.
__device__ __forceinline__ float mul_add(float a, float b, float c, float d) {
    float p = a * b, q = c * d;
    asm volatile("" : "+v"(p), "+v"(q));
    return p + q;
}

__global__ __launch_bounds__(256) void rope_kernel(const RopeArgs a) {
    const int which = blockIdx.y;
    const long long H = a.heads[which], half4 = a.D >> 3;             // float4 pairs per row
    const long long total = a.B * H * a.T * half4;
    const float* __restrict__ x = a.x[which];
    float* __restrict__ y = a.y[which];
    // (index arithmetic in 32 bits where the tensor allows: six 64-bit divisions per float4 pair were a third of this kernel's
    //  30 us at [1, 32, 2048, 128] -- round 5)
    const bool small = total < (1ll << 31);
    const unsigned half4u = (unsigned)half4, Tu = (unsigned)a.T, Hu = (unsigned)H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long d4, row, t, h, b;
        if (small) {
            const unsigned iu = (unsigned)i, rowu = iu / half4u, bhu = rowu / Tu;
            d4 = iu - rowu * half4u; row = rowu; t = rowu - bhu * Tu; b = bhu / Hu; h = bhu - (unsigned)b * Hu;
        } else {
            d4 = i % half4; row = i / half4;                          // row = (b, h, t)
            t = row % a.T;
            const long long bh = row / a.T;
            h = bh % H; b = bh / H;
        }
        long long p = a.pos[b * a.T + t];
        p = p < 0 ? 0 : (p >= a.table_rows ? a.table_rows - 1 : p);
        const float* xr = x + b * a.sb[which] + h * a.sh[which] + t * a.st[which];
        const float4 x1 = *reinterpret_cast<const float4*>(xr + 4 * d4);
        const float4 x2 = *reinterpret_cast<const float4*>(xr + 4 * d4 + (a.D >> 1));
        const float4 c1 = *reinterpret_cast<const float4*>(a.cos + p * a.D + 4 * d4);
        const float4 c2 = *reinterpret_cast<const float4*>(a.cos + p * a.D + 4 * d4 + (a.D >> 1));
        const float4 s1 = *reinterpret_cast<const float4*>(a.sin + p * a.D + 4 * d4);
        const float4 s2 = *reinterpret_cast<const float4*>(a.sin + p * a.D + 4 * d4 + (a.D >> 1));
        float4 o1, o2;
        // first half: x1 * cos + (-x2) * sin ; second half: x2 * cos + x1 * sin   (each product rounded on its own)
        o1.x = mul_add(x1.x, c1.x, -x2.x, s1.x); o1.y = mul_add(x1.y, c1.y, -x2.y, s1.y);
        o1.z = mul_add(x1.z, c1.z, -x2.z, s1.z); o1.w = mul_add(x1.w, c1.w, -x2.w, s1.w);
        o2.x = mul_add(x2.x, c2.x, x1.x, s2.x); o2.y = mul_add(x2.y, c2.y, x1.y, s2.y);
        o2.z = mul_add(x2.z, c2.z, x1.z, s2.z); o2.w = mul_add(x2.w, c2.w, x1.w, s2.w);
        float* yr = y + row * a.D;
        *reinterpret_cast<float4*>(yr + 4 * d4) = o1;
        *reinterpret_cast<float4*>(yr + 4 * d4 + (a.D >> 1)) = o2;
    }
}

int launch_rope(const RopeArgs& a, hipStream_t st) {
    const long long most = a.B * (a.heads[0] > a.heads[1] ? a.heads[0] : a.heads[1]) * a.T * (a.D >> 3);
    long long grid = (most + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)grid, 2), 256, 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace mi355q
