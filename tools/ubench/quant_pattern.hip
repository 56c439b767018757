// Memory-pattern ceiling of the fused row quantiser: 4096 x 4096 fp32 in, tiled int8 mantissas + exponent bytes
// out, (almost) no arithmetic.  Variants: 0 = row per workgroup (the kernel's pattern), 1 = same without the code
// bytes, 2 = 16-B stores (4 lanes' dwords gathered), 3 = loads only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ long long tiled_offset(long long row, long long k, long long K) {
    const long long piece = (row >> 4) * (K >> 6) + (k >> 6);
    const int chunk = (int)((k >> 4) & 3), slot = chunk ^ ((0x78 >> (2 * (int)((row >> 2) & 3))) & 3);
    return piece * 1024 + (row & 15) * 64 + slot * 16 + (k & 15);
}
template <int V>
__global__ __launch_bounds__(256) void pat(const float* __restrict__ x, int8_t* __restrict__ mt, uint8_t* __restrict__ code,
                                           long long rows, long long K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nkb = (int)(K >> 4);
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float4* x4 = reinterpret_cast<const float4*>(x + row * K);
        float4 v[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) v[it] = x4[it * 256 + tid];
        unsigned acc = 0;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int kb = it * 64 + wave * 16 + (lane >> 2);
            const unsigned pk = ((unsigned)(int)v[it].x & 0xFF) | (((unsigned)(int)v[it].y & 0xFF) << 8) |
                                (((unsigned)(int)v[it].z & 0xFF) << 16) | ((unsigned)(int)v[it].w << 24);
            if (V == 3) { acc ^= pk; continue; }
            if (V == 2) {
                const unsigned p1 = __shfl(pk, (lane & ~3) + 1), p2 = __shfl(pk, (lane & ~3) + 2), p3 = __shfl(pk, (lane & ~3) + 3);
                if ((lane & 3) == 0) *reinterpret_cast<uint4*>(mt + tiled_offset(row, (long long)kb * 16, K)) = make_uint4(pk, p1, p2, p3);
            } else {
                *reinterpret_cast<unsigned*>(mt + tiled_offset(row, (long long)kb * 16 + (lane & 3) * 4, K)) = pk;
            }
            if (V != 1 && (lane & 3) == 0) code[row * nkb + kb] = (uint8_t)(pk & 0xFF);
        }
        if (V == 3 && acc == 0x12345678u) code[row] = 1;
    }
}
int main() {
    const long long rows = 4096, K = 4096;
    float* x; int8_t* mt; uint8_t* code;
    hipMalloc(&x, rows * K * 4); hipMalloc(&mt, rows * K); hipMalloc(&code, rows * K / 16);
    hipMemset(x, 0x3f, rows * K * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int v, int grid) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) {
                if (v == 0) hipLaunchKernelGGL(pat<0>, grid, 256, 0, 0, x, mt, code, rows, K);
                if (v == 1) hipLaunchKernelGGL(pat<1>, grid, 256, 0, 0, x, mt, code, rows, K);
                if (v == 2) hipLaunchKernelGGL(pat<2>, grid, 256, 0, 0, x, mt, code, rows, K);
                if (v == 3) hipLaunchKernelGGL(pat<3>, grid, 256, 0, 0, x, mt, code, rows, K);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("variant %d grid %5d: %7.2f us\n", v, grid, ms / 50 * 1e3);
        }
    };
    for (int v = 0; v < 4; ++v) { run(v, 4096); run(v, 2048); }
    return 0;
}
