// Inner-loop variants for the 64x64 wave tile WITH the per-group fp32 rescale (no global traffic):
//   shape 32x32x32 or 16x16x64; rescale scalar (cvt, mul, fma) or "magic + packed" (C-in = 0x4B400000,
//   u = pk_fma(f, sx, -M*sx); acc = pk_fma(u, sw, acc)); conflict-free LDS swizzle per shape.
#include <hip/hip_runtime.h>
#include <cstdio>
using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr float MAGIC_F = 12582912.0f;     // 1.5 * 2^23
constexpr int MAGIC_I = 0x4B400000;

__device__ __forceinline__ int off32(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }
// 16x16x64 fragment reads: lane = (row l16, chunk lq): groups of 16 lanes cover rows {0-3,12-15} x c, {4-11} x c+1 ...
// swizzle slot = c ^ (r & 3) keeps 4 rows x 4 chunks of a 256-B bank row distinct for every 16-lane group
__device__ __forceinline__ int off16(int r, int c) { return r * 64 + ((c ^ (r & 3)) << 4); }


using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE, int RESC>
__global__ __launch_bounds__(256, 2) void loop(int steps, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 16384 + 4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (4 * 16384 + 4096) / 4; i += 256) reinterpret_cast<int*>(smem)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1;
    const float* sc = reinterpret_cast<const float*>(smem + 4 * 16384);   // [0..127] sx, [128..255] -M*sx, [256..383] sw
    float s = 0;
    const int groups = steps >> 2;
    if (SHAPE == 32) {
        const int lr = lane & 31, lh = lane >> 5;
        f32x2 acc[2][2][8] = {};
        i32x16 ci[2][2];
        i32x16 magicv;
        for (int r = 0; r < 16; ++r) magicv[r] = RESC == 2 ? MAGIC_I : 0;
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                __builtin_amdgcn_s_barrier();
                const unsigned char* sa = smem + st * 16384;
                const unsigned char* sb = sa + 8192;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    i32x4 fa[2], fb[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        fa[i] = *reinterpret_cast<const i32x4*>(sa + off32(wm * 64 + i * 32 + lr, 2 * p + lh));
                        fb[i] = *reinterpret_cast<const i32x4*>(sb + off32(wn * 64 + i * 32 + lr, 2 * p + lh));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if (st == 0 && p == 0) ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], magicv, 0, 0, 0);
                            else ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
                        }
                }
            }
            if (RESC) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float sw = sc[256 + wn * 64 + j * 32 + lr];
                        const f32x2 sw2 = {sw, sw};
                        const f32x16 cf = __builtin_bit_cast(f32x16, ci[i][j]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(&sc[wm * 64 + i * 32 + 8 * q + 4 * lh]);
                            if (RESC == 1) {
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    acc[i][j][2 * q + h][0] += (float)ci[i][j][4 * q + 2 * h] * v[2 * h] * sw;
                                    acc[i][j][2 * q + h][1] += (float)ci[i][j][4 * q + 2 * h + 1] * v[2 * h + 1] * sw;
                                }
                            } else {
                                const f32x4 n = *reinterpret_cast<const f32x4*>(&sc[128 + wm * 64 + i * 32 + 8 * q + 4 * lh]);
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const f32x2 f = {cf[4 * q + 2 * h], cf[4 * q + 2 * h + 1]};
                                    const f32x2 sx2 = {v[2 * h], v[2 * h + 1]};
                                    const f32x2 nm2 = {n[2 * h], n[2 * h + 1]};
                                    const f32x2 u = __builtin_elementwise_fma(f, sx2, nm2);
                                    acc[i][j][2 * q + h] = __builtin_elementwise_fma(u, sw2, acc[i][j][2 * q + h]);
                                }
                            }
                        }
                    }
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 8; ++r) s += acc[i][j][r][0] + acc[i][j][r][1] + (float)ci[i][j][r];
    } else {
        const int l16 = lane & 15, lq = lane >> 4;
        f32x2 acc[4][4][2] = {};
        i32x4 ci[4][4];
        i32x4 magicv;
        for (int r = 0; r < 4; ++r) magicv[r] = RESC == 2 ? MAGIC_I : 0;
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                __builtin_amdgcn_s_barrier();
                const unsigned char* sa = smem + st * 16384;
                const unsigned char* sb = sa + 8192;
                i32x4 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = *reinterpret_cast<const i32x4*>(sa + off16(wm * 64 + i * 16 + l16, lq));
                    fb[i] = *reinterpret_cast<const i32x4*>(sb + off16(wn * 64 + i * 16 + l16, lq));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (st == 0) ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], magicv, 0, 0, 0);
                        else ci[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], ci[i][j], 0, 0, 0);
                    }
            }
            if (RESC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&sc[wm * 64 + i * 16 + lq * 4]);
                    const f32x4 n = *reinterpret_cast<const f32x4*>(&sc[128 + wm * 64 + i * 16 + lq * 4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sw = sc[256 + wn * 64 + j * 16 + l16];
                        const f32x2 sw2 = {sw, sw};
                        const f32x4 cf = __builtin_bit_cast(f32x4, ci[i][j]);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (RESC == 1) {
                                acc[i][j][h][0] += (float)ci[i][j][2 * h] * v[2 * h] * sw;
                                acc[i][j][h][1] += (float)ci[i][j][2 * h + 1] * v[2 * h + 1] * sw;
                            } else {
                                const f32x2 f = {cf[2 * h], cf[2 * h + 1]};
                                const f32x2 sx2 = {v[2 * h], v[2 * h + 1]};
                                const f32x2 nm2 = {n[2 * h], n[2 * h + 1]};
                                const f32x2 u = __builtin_elementwise_fma(f, sx2, nm2);
                                acc[i][j][h] = __builtin_elementwise_fma(u, sw2, acc[i][j][h]);
                            }
                        }
                    }
                }
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 2; ++r) s += acc[i][j][r][0] + acc[i][j][r][1] + (float)ci[i][j][r];
    }
    if (s == 1.2345f) out[0] = s;
}

template <typename F>
static void run(const char* name, F launch) {
    const int steps = 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(steps); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0); launch(steps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double macs = (double)steps * 64 * 64 * 64 * 4 * 512;
    printf("%-40s %7.3f ms  %6.0f TOPS\n", name, best, 2 * macs / best / 1e9);
}

int main() {
    float* out; (void)hipMalloc(&out, 4);
    printf("64x64 wave tile, 4 waves, 2 blocks/CU, barrier per K-step of 64, group = 4 steps\n");
#define R(S, M) run("shape " #S " rescale " #M, [&](int s) { hipLaunchKernelGGL((loop<S, M>), 512, 256, 0, 0, s, out); })
    R(32, 0); R(32, 1); R(32, 2); R(16, 0); R(16, 1); R(16, 2);
    return 0;
}
