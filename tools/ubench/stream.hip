// HBM stream limits of one MI355X as a quantiser sees them: read-only, write-only and copy of N MiB with the access shapes the
// quantiser kernels could use.  hipcc --offload-arch=gfx950 -O3 -o stream stream.hip && ./stream [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool CONTIG>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ x, float* __restrict__ out, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    float acc = 0.f;
    if (CONTIG) {   // a workgroup walks U * 256 consecutive float4 per trip
        for (long long i = (long long)blockIdx.x * blockDim.x * U + threadIdx.x; i < n4; i += stride * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = i + u * 256 < n4 ? x[i + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = i + u * stride < n4 ? x[i + u * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <int NT>
__global__ __launch_bounds__(256) void wr(float4* __restrict__ y, long long n4, float val) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float4 v = make_float4(val, val, val, val);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        if (NT) __builtin_nontemporal_store(v.x, &y[i].x), __builtin_nontemporal_store(v.y, &y[i].y), __builtin_nontemporal_store(v.z, &y[i].z), __builtin_nontemporal_store(v.w, &y[i].w);
        else y[i] = v;
    }
}
template <int U>
__global__ __launch_bounds__(256) void cp(const float4* __restrict__ x, float4* __restrict__ y, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x * U + threadIdx.x; i < n4; i += stride * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = i + u * 256 < n4 ? x[i + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n4) y[i + u * 256] = v[u];
    }
}

// each workgroup owns PIECES consecutive 4-KiB pieces (grid = n4 / 256 / PIECES): no grid-stride loop
template <int PIECES>
__global__ __launch_bounds__(256) void cp_own(const float4* __restrict__ x, float4* __restrict__ y, long long n4) {
    const long long base = (long long)blockIdx.x * 256 * PIECES + threadIdx.x;
    float4 v[PIECES];
#pragma unroll
    for (int u = 0; u < PIECES; ++u) v[u] = base + u * 256 < n4 ? x[base + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < PIECES; ++u) if (base + u * 256 < n4) y[base + u * 256] = v[u];
}
template <int PIECES>
__global__ __launch_bounds__(256) void wr_own(float4* __restrict__ y, long long n4, float val) {
    const long long base = (long long)blockIdx.x * 256 * PIECES + threadIdx.x;
    const float4 v = make_float4(val, val, val, val);
#pragma unroll
    for (int u = 0; u < PIECES; ++u) if (base + u * 256 < n4) y[base + u * 256] = v;
}
// grid-stride over CHUNKS of consecutive pieces: workgroup b takes chunk b, b + grid, ... (chunk = PIECES x 4 KiB)
template <int PIECES>
__global__ __launch_bounds__(256) void cp_chunk(const float4* __restrict__ x, float4* __restrict__ y, long long n4) {
    const long long nchunks = (n4 + 256 * PIECES - 1) / (256 * PIECES);
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long long base = c * 256 * PIECES + threadIdx.x;
#pragma unroll
        for (int u = 0; u < PIECES; ++u) if (base + u * 256 < n4) y[base + u * 256] = x[base + u * 256];
    }
}

template <typename F> static double timeit(F f, int n = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / n * 1e3;
}
int main(int argc, char** argv) {
    const long long mib = argc > 1 ? atoll(argv[1]) : 512;
    const long long bytes = mib << 20, n4 = bytes / 16;
    float4 *x, *y; float* o;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(x, 0, bytes)); CK(hipMemset(y, 0, bytes));
    const double gb = bytes / 1e9;
    for (int grid : {2048, 4096, 8192, 1 << 16}) {
        printf("grid %d (%lld MiB)\n", grid, mib);
#define R(U, C) { double us = timeit([&] { hipLaunchKernelGGL((rd<U, C>), grid, 256, 0, 0, x, o, n4); }); printf("  read  U=%d %s: %7.1f us %6.0f GB/s\n", U, C ? "contig " : "strided", us, gb / us * 1e6); }
        R(1, false) R(2, false) R(4, false) R(2, true) R(4, true) R(8, true)
        { double us = timeit([&] { hipLaunchKernelGGL((wr<0>), grid, 256, 0, 0, y, n4, 1.f); }); printf("  write          : %7.1f us %6.0f GB/s\n", us, gb / us * 1e6); }
        { double us = timeit([&] { hipLaunchKernelGGL((wr<1>), grid, 256, 0, 0, y, n4, 1.f); }); printf("  write nt       : %7.1f us %6.0f GB/s\n", us, gb / us * 1e6); }
#define C_(U) { double us = timeit([&] { hipLaunchKernelGGL((cp<U>), grid, 256, 0, 0, x, y, n4); }); printf("  copy  U=%d      : %7.1f us %6.0f GB/s (read + write)\n", U, us, 2 * gb / us * 1e6); }
        C_(1) C_(2) C_(4)
    }
#define O_(P) { const int g = (int)((n4 + 256 * P - 1) / (256 * P)); double us = timeit([&] { hipLaunchKernelGGL((cp_own<P>), g, 256, 0, 0, x, y, n4); }); printf("copy  own %2d pieces (grid %6d): %7.1f us %6.0f GB/s\n", P, g, us, 2 * gb / us * 1e6); \
                double uw = timeit([&] { hipLaunchKernelGGL((wr_own<P>), g, 256, 0, 0, y, n4, 1.f); }); printf("write own %2d pieces (grid %6d): %7.1f us %6.0f GB/s\n", P, g, uw, gb / uw * 1e6); }
    O_(1) O_(2) O_(4) O_(8) O_(16)
#define K_(P, G) { double us = timeit([&] { hipLaunchKernelGGL((cp_chunk<P>), G, 256, 0, 0, x, y, n4); }); printf("copy  chunk-stride %2d pieces grid %5d: %7.1f us %6.0f GB/s\n", P, G, us, 2 * gb / us * 1e6); }
    K_(4, 2048) K_(16, 2048) K_(4, 8192) K_(16, 8192)
    { double us = timeit([&] { CK(hipMemsetAsync(y, 0, bytes, 0)); }); printf("hipMemsetAsync: %7.1f us %6.0f GB/s\n", us, gb / us * 1e6); }
    { double us = timeit([&] { CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)); }); printf("hipMemcpyAsync D2D: %7.1f us %6.0f GB/s (read + write)\n", us, 2 * gb / us * 1e6); }
    return 0;
}
