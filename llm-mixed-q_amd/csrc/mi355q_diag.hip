// mi355q_diag.hip -- diagnostic entry points (NOT part of include/mi355q.h; bench.py's `roofline.peak_measured`).
//
// mi355q_debug_mfma_i8_peak: what the int8 matrix pipes of THIS device deliver when nothing else is in the way --
// back-to-back v_mfma_i32_16x16x64_i8 from registers in the tile GEMM's register blocking (256 workgroups x 8 waves, wave tile
// 128 x 64: 32 MFMAs per K-step, two waves per SIMD), on RANDOM int8 operands (the chip holds a lower clock on random data than
// on constants: cdna guide, "DVFS give-back" item 1), no LDS, no global traffic.  SURVEY 8d asks for the achievable figure next to
// the nominal 5 POPS; round 3 measured 4.28 POPS with near-constant operands (tools/ubench/mx_rate.hip, profiles/r03_mx_rate.txt).
// Also returns the clock the loop ran at: delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz), median over workgroups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace {
using i32x4 = __attribute__((ext_vector_type(4))) int;

__device__ __forceinline__ unsigned diag_hash(unsigned v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}

__global__ __launch_bounds__(512, 1) void mfma_i8_peak_kernel(int steps, int* __restrict__ sink, unsigned long long* __restrict__ clk) {
    const int lane = threadIdx.x & 63;
    i32x4 fa[8], fb[4], acc[8][4];
    // random bytes in [-31 << 2, 31 << 2]-ish: full-range toggling like W6 mantissas shifted onto a row exponent
    for (int i = 0; i < 8; ++i)
        for (int q = 0; q < 4; ++q) fa[i][q] = (int)diag_hash(blockIdx.x * 7919u + threadIdx.x * 131u + i * 17u + q);
    for (int j = 0; j < 4; ++j)
        for (int q = 0; q < 4; ++q) fb[j][q] = (int)diag_hash(blockIdx.x * 104729u + threadIdx.x * 257u + j * 29u + q + 99u);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
        // (the operands change every step -- one rotate per fragment register set, nothing the MFMA stream waits for)
        fa[0][0] = __builtin_amdgcn_alignbit(fa[0][0], fa[0][0], 7);
        fb[0][1] = __builtin_amdgcn_alignbit(fb[0][1], fb[0][1], 5);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int r = 0;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 4; ++q) r ^= acc[i][j][q];
    if (r == 0x12345678) sink[0] = r;                      // (keeps the accumulators live)
    if (lane == 0 && (threadIdx.x >> 6) == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
}  // namespace

// steps: K-steps of 64 per workgroup and launch (>= 1000 for a stable figure); soak_ms: un-timed launches for this long first (the
// clock a short burst runs at is not the clock the chip holds).  *tops: 2 * 256 * 256 * 64 * steps * 256 workgroups / time;
// *clock_ghz: the in-kernel clock, median over the workgroups of the last launch.  Synchronises the stream.
extern "C" __attribute__((visibility("default"))) int mi355q_debug_mfma_i8_peak(int steps, double soak_ms, double* tops,
                                                                                double* clock_ghz, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* sink = nullptr;
    unsigned long long* clk = nullptr;
    const int wgs = 256;
    if (hipMalloc(&sink, 64) != hipSuccess || hipMalloc(&clk, wgs * 16) != hipSuccess) return (int)hipGetLastError();
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float ms = 0.f, done = 0.f;
    // soak
    while (done < soak_ms) {
        (void)hipEventRecord(a, st);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mfma_i8_peak_kernel, wgs, 512, 0, st, steps, sink, clk);
        (void)hipEventRecord(b, st);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        done += ms > 0.f ? ms : 1.f;
    }
    const int reps = 8;
    (void)hipEventRecord(a, st);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_i8_peak_kernel, wgs, 512, 0, st, steps, sink, clk);
    (void)hipEventRecord(b, st);
    (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(wgs * 2);
    (void)hipMemcpy(h.data(), clk, wgs * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int w = 0; w < wgs; ++w)
        if (h[2 * w + 1]) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    if (clock_ghz) *clock_ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    if (tops) *tops = 2.0 * 256 * 256 * 64 * (double)steps * wgs * reps / ((double)ms * 1e-3) / 1e12;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(sink);
    (void)hipFree(clk);
    return (int)hipGetLastError();
}
