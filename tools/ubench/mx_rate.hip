// Issue rate of the MX matrix instruction against the int8 one in the tile GEMM's register blocking (wave tile 128 x 64: per
// K-step 8 A fragments x 4 B fragments = 32 MFMAs, two waves per SIMD, 256 workgroups of 512 threads), operands from
// registers only (MFMA) or re-read from LDS every K-step as the real loop does (MFMA + LDS).  No global traffic: this is
// the ceiling the K loop of an FP6 (e2m3, E8M0 per 32) W4A4 kernel would start from.
//   int8:  v_mfma_i32_16x16x64_i8            K-step = 64,  fragment 16 B per lane
//   fp6:   v_mfma_scale_f32_16x16x128_f8f6f4 K-step = 128, fragment 24 B per lane (+ one scale VGPR per operand)
//   fp4:   the same instruction with FP4 operands, fragment 16 B per lane
// hipcc --offload-arch=gfx950 -O3 -o mx_rate mx_rate.hip && ./mx_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KIND, bool LDS>      // KIND 0 int8, 1 fp6, 2 fp4
__global__ __launch_bounds__(512, 1) void loop(int steps, float* out, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(sm)[i] = 0x01010101u * (i & 3);
    __syncthreads();
    i32x8 fa[8], fb[4];
    for (int i = 0; i < 8; ++i) for (int q = 0; q < 8; ++q) fa[i][q] = 0x00410041 + lane + i;
    for (int j = 0; j < 4; ++j) for (int q = 0; q < 8; ++q) fb[j][q] = 0x00410041 + lane * 3 + j;
    f32x4 accf[8][4];
    i32x4 acci[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) { accf[i][j] = f32x4{0, 0, 0, 0}; acci[i][j] = i32x4{0, 0, 0, 0}; }
    const unsigned char* base = sm + (wave & 1) * 24 * 1024 + lane * 32;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        if (LDS) {
            // fragment bytes per lane: int8 / fp4 16, fp6 24 -- read as b128 (+ b64)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(base + i * 2048 + (s & 1) * 16);
                fa[i][0] ^= lo[0]; fa[i][1] ^= lo[1]; fa[i][2] ^= lo[2]; fa[i][3] ^= lo[3];
                if (KIND == 1) { const long long hi = *reinterpret_cast<const long long*>(base + i * 2048 + 16); fa[i][4] ^= (int)hi; fa[i][5] ^= (int)(hi >> 32); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(base + 16384 + j * 2048 + (s & 1) * 16);
                fb[j][0] ^= lo[0]; fb[j][1] ^= lo[1]; fb[j][2] ^= lo[2]; fb[j][3] ^= lo[3];
                if (KIND == 1) { const long long hi = *reinterpret_cast<const long long*>(base + 16384 + j * 2048 + 16); fb[j][4] ^= (int)hi; fb[j][5] ^= (int)(hi >> 32); }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (KIND == 0) {
                    const i32x4 a4 = {fa[i][0], fa[i][1], fa[i][2], fa[i][3]}, b4 = {fb[j][0], fb[j][1], fb[j][2], fb[j][3]};
                    acci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, acci[i][j], 0, 0, 0);
                } else if (KIND == 1) {
                    accf[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[i], fb[j], accf[i][j], 2, 2, 0, 127, 0, 127);
                } else {
                    accf[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[i], fb[j], accf[i][j], 4, 4, 0, 127, 0, 127);
                }
            }
        if (LDS) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) r += accf[i][j][q] + (float)acci[i][j][q];
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND, bool LDS> static void run(const char* name, int kstep) {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
    const int steps = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop<KIND, LDS>), 256, 512, 0, 0, steps, out, clk);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((loop<KIND, LDS>), 256, 512, 0, 0, steps, out, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    // per workgroup and K-step: 256 x 256 x kstep MACs
    const double ops = 2.0 * 256 * 256 * kstep * (double)steps * 256;
    printf("%-22s %7.1f us per K-step-of-%3d per workgroup, %6.0f TOPS, s_memtime %5.0f ticks per step (100 MHz)\n", name,
           ms * 1e3 / steps, kstep, ops / (ms * 1e-3) / 1e12, (double)c / steps);
}
int main() {
    run<0, false>("int8 MFMA only", 64);
    run<0, true>("int8 MFMA + LDS", 64);
    run<1, false>("fp6 MX MFMA only", 128);
    run<1, true>("fp6 MX MFMA + LDS", 128);
    run<2, false>("fp4 MX MFMA only", 128);
    run<2, true>("fp4 MX MFMA + LDS", 128);
    return 0;
}
