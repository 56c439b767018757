// Microbenchmark: sustained issue rate of the gfx950 MFMA shapes used (or considered) by the GEMM.
// Each wave runs ITERS x 8 back-to-back MFMAs on 4 or 8 independent accumulators; operands in registers.
// Prints cycles per MFMA per SIMD (from s_memtime) and chip TOPS (from wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, unsigned seed, unsigned long long* cyc, int* sink) {
    const int lane = threadIdx.x & 63;
    i32x4 a = {(int)(lane * 2654435761u ^ seed), (int)(lane * 40503u + seed), (int)(seed >> 3) ^ lane, lane + 7};
    i32x4 b = {(int)(lane * 97u ^ seed), (int)(seed * 31u + lane), lane ^ 0x55aa55aa, (int)seed};
    long al = ((long)a[0] << 32) | (unsigned)a[1], bl = ((long)b[0] << 32) | (unsigned)b[1];
    i32x16 c32[4] = {};
    i32x4 c16[8] = {};
    f32x16 f32[4] = {};
    f32x4 f16[8] = {};
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(float)((lane + i) & 7); hb[i] = (__bf16)(float)((lane * 3 + i) & 7); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) c32[u & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c32[u & 3], 0, 0, 0);
            if (KIND == 1) c16[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c16[u], 0, 0, 0);
            if (KIND == 2) c32[u & 3] = __builtin_amdgcn_mfma_i32_32x32x16_i8(al, bl, c32[u & 3], 0, 0, 0);
            if (KIND == 3) c16[u] = __builtin_amdgcn_mfma_i32_16x16x32_i8(al, bl, c16[u], 0, 0, 0);
            if (KIND == 4) f32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, f32[u & 3], 0, 0, 0);
            if (KIND == 5) f16[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, f16[u], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int u = 0; u < 4; ++u) s += c32[u][0] + c32[u][7] + (int)f32[u][3];
    for (int u = 0; u < 8; ++u) s += c16[u][1] + (int)f16[u][2];
    if (s == 0x7fffffff) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    unsigned long long* cyc; int* sink;
    hipMalloc(&cyc, 8); hipMalloc(&sink, 4);
    const char* names[] = {"i32_32x32x32_i8", "i32_16x16x64_i8", "i32_32x32x16_i8 (legacy)", "i32_16x16x32_i8 (legacy)",
                           "f32_32x32x16_bf16", "f32_16x16x32_bf16"};
    const double macs[] = {32. * 32 * 32, 16. * 16 * 64, 32. * 32 * 16, 16. * 16 * 32, 32. * 32 * 16, 16. * 16 * 32};
    const int iters = 20000;
    for (int waves = 1; waves <= 2; ++waves)
        for (int kind = 0; kind < 6; ++kind) {
            const int threads = 256 * waves, blocks = 256;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k<0>, blocks, threads, 0, 0, iters, 12345u, cyc, sink); break;
                    case 1: hipLaunchKernelGGL(k<1>, blocks, threads, 0, 0, iters, 12345u, cyc, sink); break;
                    case 2: hipLaunchKernelGGL(k<2>, blocks, threads, 0, 0, iters, 12345u, cyc, sink); break;
                    case 3: hipLaunchKernelGGL(k<3>, blocks, threads, 0, 0, iters, 12345u, cyc, sink); break;
                    case 4: hipLaunchKernelGGL(k<4>, blocks, threads, 0, 0, iters, 12345u, cyc, sink); break;
                    default: hipLaunchKernelGGL(k<5>, blocks, threads, 0, 0, iters, 12345u, cyc, sink); break;
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double n_per_wave = (double)iters * 8;
            const double total_macs = n_per_wave * macs[kind] * blocks * 4 * waves;
            printf("%-26s waves/SIMD %d: %6.1f cyc/MFMA/wave (memtime@100MHz-ticks? raw %llu)  wall %.3f ms  %.0f TOPS  -> %.1f cyc/MFMA/SIMD @2.4GHz\n",
                   names[kind], waves, (double)c / n_per_wave, c, ms, 2 * total_macs / ms / 1e9,
                   ms * 1e-3 * 2.4e9 / (n_per_wave * waves));
        }
    return 0;
}
