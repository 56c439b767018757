// Microbenchmark: the tile GEMM's MFMA pattern from registers -- 32 accumulators (8 A fragments x 4 B fragments of
// v_mfma_i32_16x16x64_i8 per K-step), one or two waves per SIMD, no LDS, no barriers.  Cycles per MFMA per SIMD (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
using i32x4 = __attribute__((ext_vector_type(4))) int;

template <int NACC_I, int VARIANT>   // VARIANT 0: A_i x B_j;  1: one A, one B for everything;  2: accumulators passed through AGPR-style moves
__global__ __launch_bounds__(512) void k(int steps, unsigned seed, unsigned long long* cyc, int* sink) {
    const int lane = threadIdx.x & 63;
    i32x4 fa[8], fb[4];
    for (int i = 0; i < 8; ++i) fa[i] = i32x4{(int)(lane * 2654435761u ^ seed) + i, (int)(lane * 40503u + seed) ^ i, (int)(seed >> 3) ^ lane, lane + 7 * i};
    for (int j = 0; j < 4; ++j) fb[j] = i32x4{(int)(lane * 97u ^ seed) - j, (int)(seed * 31u + lane) + j, lane ^ 0x55aa55aa, (int)seed + j};
    i32x4 acc[NACC_I][4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < steps; ++t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i % NACC_I][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(VARIANT == 1 ? fa[0] : fa[i], VARIANT == 1 ? fb[0] : fb[j], acc[i % NACC_I][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < NACC_I; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] ^ acc[i][j][3];
    if (s == 0x7fffffff) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    unsigned long long* cyc; int* sink;
    hipMalloc(&cyc, 8); hipMalloc(&sink, 4);
    const int steps = 4000;
    for (int waves = 1; waves <= 2; ++waves)
        for (int v = 0; v < 4; ++v) {
            const int threads = 256 * waves, blocks = 256;
            auto launch = [&]() {
                if (v == 0) hipLaunchKernelGGL((k<8, 0>), blocks, threads, 0, 0, steps, 12345u, cyc, sink);
                if (v == 1) hipLaunchKernelGGL((k<8, 1>), blocks, threads, 0, 0, steps, 12345u, cyc, sink);
                if (v == 2) hipLaunchKernelGGL((k<2, 0>), blocks, threads, 0, 0, steps, 12345u, cyc, sink);
                if (v == 3) hipLaunchKernelGGL((k<2, 1>), blocks, threads, 0, 0, steps, 12345u, cyc, sink);
            };
            launch(); hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double n = (double)steps * 32;
            const char* names[] = {"32 acc, A_i x B_j", "32 acc, one A one B", "8 acc, A_i x B_j", "8 acc, one A one B"};
            printf("%-22s waves/SIMD %d: %6.2f cyc per MFMA per wave, %6.2f per SIMD; wall %.3f ms -> %.0f TOPS\n", names[v], waves,
                   (double)c / n, (double)c / n / waves, ms, 2.0 * n * 16 * 16 * 64 * blocks * 4 * waves / ms / 1e9);
        }
    return 0;
}
