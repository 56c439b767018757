// Inner-loop ceiling for int32-only accumulation (whole-K chains): wave tile 128x64 or 64x64, LDS-fed.
#include <hip/hip_runtime.h>
#include <cstdio>
using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
__device__ __forceinline__ int off64(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }
constexpr int STAGE = 32768;   // A 256x64 + B 256x64

// 8 waves: 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 tiles of 32x32.  MODE bit0 barrier/step, bit2 prefetch
template <int MODE>
__global__ __launch_bounds__(512, 2) void loop32_128x64(int steps, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * STAGE / 4; i += 512) reinterpret_cast<int*>(smem)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    const int wm = wave >> 2, wn = wave & 3, lr = lane & 31, lh = lane >> 5;
    const int arow = wm * 128 + lr, brow = wn * 64 + lr;
    i32x16 ci[4][2] = {};
    for (int t = 0; t < steps; ++t) {
        if (MODE & 1) __builtin_amdgcn_s_barrier();
        const unsigned char* sa = smem + (t & 3) * STAGE;
        const unsigned char* sb = sa + 16384;
        i32x4 fa[2][4], fb[2][2];
        if (MODE & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = *reinterpret_cast<const i32x4*>(sa + off64(arow + i * 32, lh));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][j] = *reinterpret_cast<const i32x4*>(sb + off64(brow + j * 32, lh));
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (MODE & 4) {
                if (p == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[1][i] = *reinterpret_cast<const i32x4*>(sa + off64(arow + i * 32, 2 + lh));
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[1][j] = *reinterpret_cast<const i32x4*>(sb + off64(brow + j * 32, 2 + lh));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[p][i] = *reinterpret_cast<const i32x4*>(sa + off64(arow + i * 32, 2 * p + lh));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[p][j] = *reinterpret_cast<const i32x4*>(sb + off64(brow + j * 32, 2 * p + lh));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[p][i], fb[p][j], ci[i][j], 0, 0, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += ci[i][j][r];
    if (s == 12345) out[0] = (float)s;
}

// 16x16x64: wave tile 128 x 64 = 8 x 4 tiles
template <int MODE>
__global__ __launch_bounds__(512, 2) void loop16_128x64(int steps, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * STAGE / 4; i += 512) reinterpret_cast<int*>(smem)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    const int wm = wave >> 2, wn = wave & 3, l16 = lane & 15, lq = lane >> 4;
    i32x4 ci[8][4] = {};
    for (int t = 0; t < steps; ++t) {
        if (MODE & 1) __builtin_amdgcn_s_barrier();
        const unsigned char* sa = smem + (t & 3) * STAGE;
        const unsigned char* sb = sa + 16384;
        i32x4 fa[8], fb[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const i32x4*>(sa + off64(wm * 128 + i * 16 + l16, lq));
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const i32x4*>(sb + off64(wn * 64 + j * 16 + l16, lq));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += ci[i][j][r];
    if (s == 12345) out[0] = (float)s;
}

template <typename F>
static void run(const char* name, F launch) {
    const int steps = 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(steps); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(steps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double)steps * 256 * 256 * 64 * 256;
    printf("%-44s %7.3f ms  %6.0f TOPS\n", name, ms, 2 * macs / ms / 1e9);
}

int main() {
    float* out; (void)hipMalloc(&out, 4);
    printf("256x256 tile, 8 waves x (128x64), 1 block/CU; mode bits: 1 = barrier per K-step, 4 = fragment prefetch\n");
    run("32x32x32 mode 0", [&](int s) { hipLaunchKernelGGL(loop32_128x64<0>, 256, 512, 0, 0, s, out); });
    run("32x32x32 mode 1", [&](int s) { hipLaunchKernelGGL(loop32_128x64<1>, 256, 512, 0, 0, s, out); });
    run("32x32x32 mode 4", [&](int s) { hipLaunchKernelGGL(loop32_128x64<4>, 256, 512, 0, 0, s, out); });
    run("32x32x32 mode 5", [&](int s) { hipLaunchKernelGGL(loop32_128x64<5>, 256, 512, 0, 0, s, out); });
    run("16x16x64 mode 0", [&](int s) { hipLaunchKernelGGL(loop16_128x64<0>, 256, 512, 0, 0, s, out); });
    run("16x16x64 mode 1", [&](int s) { hipLaunchKernelGGL(loop16_128x64<1>, 256, 512, 0, 0, s, out); });
    return 0;
}
