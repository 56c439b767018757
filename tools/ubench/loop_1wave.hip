// Microbenchmark: the K-step with ONE wave per SIMD (256-thread workgroup, wave tile 128 x 128: 64 MFMAs, 16 fragment
// reads and 8 LDS-DMA pieces per wave and step, one barrier) against loop_parts' two waves per SIMD (128 x 64 wave tiles).
#include <hip/hip_runtime.h>
#include <cstdio>
using i32x4 = __attribute__((ext_vector_type(4))) int;
using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

template <int READS, int DMA, int BARRIER>
__global__ __launch_bounds__(256) void k(int steps, const int8_t* __restrict__ src, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * 32768 / 4; i += 256) reinterpret_cast<int*>(smem)[i] = (i * 2654435761u) >> 5;
    __syncthreads();
    i32x4 fa[4], fb[8], fbn[8];
    for (int i = 0; i < 4; ++i) fa[i] = i32x4{lane * 3 + i, lane ^ 77, i, lane};
    for (int i = 0; i < 8; ++i) { fb[i] = i32x4{lane * 5 - i, lane ^ 33, i + 1, lane + 9}; fbn[i] = fb[i]; }
    i32x4 acc[8][8] = {};
    const int va = (wave >> 1) * 8192 + (lane >> 4) * 256 + (lane & 15) * 16, vb = 16384 + (wave & 1) * 8192 + (lane >> 4) * 256 + (lane & 15) * 16;
    const int8_t* gsrc = src + (long long)blockIdx.x * 65536 + wave * 8192 + lane * 16;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int slot = 0;
    for (int t = 0; t < steps; t += 2) {
#define STEP(FB, FBN)                                                                                                      \
        {                                                                                                                  \
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
        if (BARRIER) __builtin_amdgcn_s_barrier();                                                                         \
        const int ac = va + slot * 32768, bc = vb + slot * 32768;                                                          \
        const int dslot = slot == 0 ? 2 : slot - 1;                                                                        \
        GROUP(0, FB, FBN, 3) GROUP(1, FB, FBN, 4) GROUP(2, FB, FBN, 5) GROUP(3, FB, FBN, 5) GROUP(4, FB, FBN, 5) GROUP(5, FB, FBN, 5) GROUP(6, FB, FBN, 5) GROUP(7, FB, FBN, 5) \
        slot = slot == 2 ? 0 : slot + 1;                                                                                   \
        }
#define GROUP(i, FB, FBN, w)                                                                                               \
        if (READS) {                                                                                                       \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(i + 2) & 3]) : "v"(ac), "n"(((i + 2) & 7) * 1024));   \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FBN[i]) : "v"(bc), "n"(i * 1024));                         \
        }                                                                                                                  \
        if (DMA) __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + ((t & 7) * 8 + i) * 1024), (lptr_t)(smem + dslot * 32768 + (wave * 8 + i) * 1024), 16, 0, 0); \
        if (READS) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(w));                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i & 3], FB[j], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);
        STEP(fb, fbn)
        STEP(fbn, fb)
#undef GROUP
#undef STEP
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] ^ acc[i][j][2];
    if (s == 0x7fffffff) out[7] = s;
    if (tid == 0 && blockIdx.x == 3) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

int main() {
    unsigned long long* out; int8_t* src;
    hipMalloc(&out, 64); hipMalloc(&src, 256ll * 65536 + 131072);
    hipMemset(src, 1, 256ll * 65536 + 131072);
    const int steps = 4000;
    const char* names[] = {"MFMA only", "+ 16 ds_read_b128", "+ 8 LDS-DMA pieces", "+ reads + DMA", "+ reads + DMA + barrier"};
    for (int v = 0; v < 5; ++v) {
        auto launch = [&]() {
            if (v == 0) hipLaunchKernelGGL((k<0, 0, 0>), 256, 256, 0, 0, steps, src, out);
            if (v == 1) hipLaunchKernelGGL((k<1, 0, 0>), 256, 256, 0, 0, steps, src, out);
            if (v == 2) hipLaunchKernelGGL((k<0, 1, 0>), 256, 256, 0, 0, steps, src, out);
            if (v == 3) hipLaunchKernelGGL((k<1, 1, 0>), 256, 256, 0, 0, steps, src, out);
            if (v == 4) hipLaunchKernelGGL((k<1, 1, 1>), 256, 256, 0, 0, steps, src, out);
        };
        launch(); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("1 wave/SIMD %-26s %7.1f cycles per K-step at %4.0f MHz = %6.1f ns (in-kernel); wall %.1f ns per step -> %.0f TOPS\n", names[v],
               (double)h[0] / steps, (double)h[0] / h[1] * 100.0, (double)h[1] * 10.0 / steps, ms * 1e6 / steps, 2.0 * steps * 256 * 256 * 64 * 256 / ms / 1e9);
    }
    return 0;
}
