// What does it cost a wave to ISSUE its share of a K-step's operand pieces (1 KiB = 64 lanes x 16 B each), by delivery path,
// when the wave is alone on its SIMD (the small-tile GEMM: 4-wave workgroups, mi355q_gemm_v10.hip) or shares it with one
// partner (WGS = 2 workgroups per compute unit)?  Per K-step and wave: 8 ds_read_b128 fragment reads, 16 int8 MFMAs
// (a 64 x 64 wave tile), one barrier, and P pieces fetched from an L2-resident source:
//   mode 0  nothing fetched (the MFMA + fragment-read floor)
//   mode 1  LDS-DMA: s_mov m0 + buffer_load_dwordx4 ... offen lds          (what the tile GEMMs do)
//   mode 2  buffer_load_dwordx4 into registers, ds_write_b128 two steps later
//   mode 3  global_load_dwordx4 (saddr form) into registers, ds_write_b128 two steps later
// Prints clocks per K-step (s_memtime of workgroup 0) and the whole-launch time.
//   hipcc --offload-arch=gfx950 -O3 -o vmem_issue vmem_issue.hip && ./vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using i32x4 = __attribute__((ext_vector_type(4))) int;

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
template <int N> __device__ __forceinline__ void waitv() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void waitl() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#define SB() __builtin_amdgcn_sched_barrier(0)
template <int OFF> __device__ __forceinline__ void dsr(i32x4& d, int addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
__device__ __forceinline__ void dsw(int addr, const i32x4& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void ldma(int voff, i32x4 rs, unsigned so, int ld) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(so), "s"(ld) : "memory");
}
__device__ __forceinline__ void bload(i32x4& d, int voff, i32x4 rs, unsigned so) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(rs), "s"(so) : "memory");
}
__device__ __forceinline__ void gload(i32x4& d, int voff, const char* p) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(p) : "memory");
}

template <int MODE, int P>
__global__ __launch_bounds__(256) void loop(const char* __restrict__ src, unsigned span, int steps, float* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];      // 3 stages x 16 KiB + whatever the host asks for
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = 0x01010101u * (i & 3);
    __syncthreads();
    const i32x4 rs = {(int)(unsigned)(unsigned long long)src, (int)(unsigned)((unsigned long long)src >> 32), (int)span, 0x00020000};
    const int voff = lane * 16;
    const int lbase = (int)(unsigned long long)(sm) + 0;                      // LDS byte address of the ring (shared-space pointer's low word)
    const int va = lbase + (wave >> 1) * 4096 + lane * 16, vb = lbase + 8192 + (wave & 1) * 4096 + lane * 16;
    i32x4 acc[4][4], fa[4], fb[4], r[3][P > 0 ? P : 1];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) for (int q = 0; q < (P > 0 ? P : 1); ++q) r[i][q] = i32x4{0, 0, 0, 0};
    unsigned piece = (blockIdx.x * 37u + wave * P) * 1024u;
    auto fetch = [&](auto si, auto qi) {                                    // piece q of the step that lands in slot s
        constexpr int s = decltype(si)::value, q = decltype(qi)::value;
        const unsigned so = (piece + q * 1024u) & (span - 1);
        const int ld = lbase + s * 16384 + (wave * P + q) % 16 * 1024;
        if constexpr (MODE == 1) {
            ldma(voff, rs, so, ld);
        } else if constexpr (MODE == 2) {
            bload(r[s][q], voff, rs, so);
        } else if constexpr (MODE == 3) {
            gload(r[s][q], voff, src + so);
        }
    };
    auto put = [&](auto si, auto qi) {
        constexpr int s = decltype(si)::value, q = decltype(qi)::value;
        if constexpr (MODE >= 2) {
            const int ld = lbase + s * 16384 + (wave * P + q) % 16 * 1024 + voff;
            dsw(ld, r[s][q]);
        }
    };
    auto body = [&](auto ci) {
        constexpr int C = decltype(ci)::value;                              // the LDS stage being computed
        // LDS-DMA: step C fetches into stage (C + 2) % 3, read two steps on.  Registers: step C fetches into register slot C; the
        // slot fetched two steps ago ((C + 1) % 3) is written to LDS stage (C + 1) % 3 now and read in the NEXT step
        constexpr int F = MODE == 1 ? (C + 2) % 3 : C, W = (C + 1) % 3;
        waitv<P>();                                                         // everything but the previous step's pieces has arrived
        if constexpr (MODE >= 2) sfor<0, P>([&](auto qi) { put(std::integral_constant<int, W>{}, qi); });
        __builtin_amdgcn_s_barrier();
        const int ac = va + C * 16384, bc = vb + C * 16384;
        sfor<0, 4>([&](auto ji) { constexpr int j = decltype(ji)::value; dsr<j * 1024>(fb[j], bc); });
        sfor<0, 4>([&](auto ii) { constexpr int i = decltype(ii)::value; dsr<i * 1024>(fa[i], ac); });
        sfor<0, 4>([&](auto gi) {
            constexpr int g = decltype(gi)::value;
            waitl<3 - g>();
            SB();
            sfor<0, 4>([&](auto ji) {
                constexpr int j = decltype(ji)::value;
                acc[g][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[g], fb[j], acc[g][j], 0, 0, 0);
                if constexpr (j == 1) sfor<0, P>([&](auto qi) { if constexpr (decltype(qi)::value % 4 == g) fetch(std::integral_constant<int, F>{}, qi); });
            });
            SB();
        });
        piece += 4u * P * 1024u;
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s + 2 < steps; s += 3) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
        body(std::integral_constant<int, 2>{});
    }
    waitv<0>();
    waitl<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int x = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) x += acc[i][j][q];
    for (int i = 0; i < 3; ++i) for (int q = 0; q < (P > 0 ? P : 1); ++q) x += r[i][q][0];
    if (x == 0x12345677) out[0] = (float)x;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

// The same step with the NEXT step's fragments read during this step's MFMAs (two register sets, four LDS stages): what a
// wave that is alone on its SIMD needs, since no partner's MFMAs cover its LDS latency.  DMA 0 | 1 only.
// PLACE: 0 = the next step's reads spread over the step (A behind MFMA 0, B behind MFMA 2 of every group); 1 = all eight behind the
// first eight MFMAs (groups 0 and 1), LDS-DMA one per group; 2 = reads in groups 0-1, LDS-DMA pieces in groups 2-3 only
template <int DMA, int P, int PLACE>
__global__ __launch_bounds__(256) void loop_pf(const char* __restrict__ src, unsigned span, int steps, float* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];      // 4 stages x 16 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = 0x01010101u * (i & 3);
    __syncthreads();
    const i32x4 rs = {(int)(unsigned)(unsigned long long)src, (int)(unsigned)((unsigned long long)src >> 32), (int)span, 0x00020000};
    const int voff = lane * 16;
    const int lbase = (int)(unsigned long long)(sm);
    const int va = lbase + (wave >> 1) * 4096 + lane * 16, vb = lbase + 8192 + (wave & 1) * 4096 + lane * 16;
    i32x4 acc[4][4], fa[2][4], fb[2][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
    unsigned piece = (blockIdx.x * 37u + wave * P) * 1024u;
    sfor<0, 4>([&](auto ji) { constexpr int j = decltype(ji)::value; dsr<j * 1024>(fb[0][j], vb); dsr<j * 1024>(fa[0][j], va); });
    auto body = [&](auto ci) {
        constexpr int C = decltype(ci)::value;                              // stage being computed (0..3); registers C & 1
        constexpr int N = (C + 1) % 4, F = (C + 3) % 4, B = C & 1;
        waitv<P>();                                                         // stage N has landed (fetched two steps ago)
        waitl<0>();                                                         // this step's fragments (read during the last step)
        __builtin_amdgcn_s_barrier();
        const int an = va + N * 16384, bn = vb + N * 16384;
        sfor<0, 4>([&](auto gi) {
            constexpr int g = decltype(gi)::value;
            SB();
            sfor<0, 4>([&](auto ji) {
                constexpr int j = decltype(ji)::value;
                acc[g][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[B][g], fb[B][j], acc[g][j], 0, 0, 0);
                if constexpr (PLACE == 0) {
                    if constexpr (j == 0) dsr<g * 1024>(fa[B ^ 1][g], an);
                    if constexpr (j == 2) dsr<g * 1024>(fb[B ^ 1][g], bn);
                } else {
                    if constexpr (g == 0) dsr<j * 1024>(fb[B ^ 1][j], bn);
                    if constexpr (g == 1) dsr<j * 1024>(fa[B ^ 1][j], an);
                }
                if constexpr (DMA == 1 && PLACE < 2 && j == 1) sfor<0, P>([&](auto qi) {
                    constexpr int q = decltype(qi)::value;
                    if constexpr (q % 4 == g) ldma(voff, rs, (piece + q * 1024u) & (span - 1), lbase + F * 16384 + (wave * P + q) % 16 * 1024);
                });
                if constexpr (DMA == 1 && PLACE == 2 && g >= 2) sfor<0, P>([&](auto qi) {
                    constexpr int q = decltype(qi)::value;
                    if constexpr (q % 8 == (g - 2) * 4 + j) ldma(voff, rs, (piece + q * 1024u) & (span - 1), lbase + F * 16384 + (wave * P + q) % 16 * 1024);
                });
            });
            SB();
        });
        piece += 4u * P * 1024u;
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s + 3 < steps; s += 4) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
        body(std::integral_constant<int, 2>{});
        body(std::integral_constant<int, 3>{});
    }
    waitv<0>();
    waitl<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int x = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) x += acc[i][j][q];
    if (x == 0x12345677) out[0] = (float)x;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

// The same again with the barrier every KS K-steps (a stage = KS steps) and NWV waves a workgroup (8: two a SIMD inside ONE workgroup, in
// lockstep at its barriers -- what an eight-wave tile gets instead of two free-running workgroups).  Placement 1, LDS-DMA, P pieces a wave-step.
// SHARE: 0 = every workgroup its own stream; 1 = the GEMM's pattern: the first half of a wave's pieces from a stream shared by the 16
// workgroups of a tile row (blockIdx / 16), the second half from one shared by a tile column (blockIdx % 16)
template <int P, int KS, int NWV, int SHARE = 0>
__global__ __launch_bounds__(64 * NWV) void loop_ks(const char* __restrict__ src, unsigned span, int steps, float* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];      // 3 stages x KS x 16 KiB (the fragment addresses wrap inside 16 KiB)
    constexpr int STG = KS * 16384;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 3 * STG / 4; i += 64 * NWV) reinterpret_cast<unsigned*>(sm)[i] = 0x01010101u * (i & 3);
    __syncthreads();
    const i32x4 rs = {(int)(unsigned)(unsigned long long)src, (int)(unsigned)((unsigned long long)src >> 32), (int)span, 0x00020000};
    const int voff = lane * 16;
    const int lbase = (int)(unsigned long long)(sm);
    const int va = lbase + ((wave >> 1) & 1) * 4096 + lane * 16, vb = lbase + 8192 + (wave & 1) * 4096 + lane * 16;
    i32x4 acc[4][4], fa[2][4], fb[2][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
    unsigned piece = (blockIdx.x * 37u + wave * P) * 1024u;
    const unsigned pa = ((blockIdx.x / 16u) * 4099u + wave * P) * 1024u, pb = ((blockIdx.x % 16u) * 8209u + 1000003u + wave * P) * 1024u;
    sfor<0, 4>([&](auto ji) { constexpr int j = decltype(ji)::value; dsr<j * 1024>(fb[0][j], vb); dsr<j * 1024>(fa[0][j], va); });
    // sub-step u of stage C; fragments of the next sub-step (same stage, or sub-step 0 of stage C + 1) are read during it
    auto sub = [&](auto ci, auto ui) {
        constexpr int C = decltype(ci)::value, U = decltype(ui)::value, B = (C * KS + U) & 1;
        constexpr int NC = U + 1 < KS ? C : (C + 1) % 3, NU = U + 1 < KS ? U + 1 : 0, F = (C + 2) % 3;
        if constexpr (U == 0) {
            waitv<P * KS>();                                                // stage C + 1 has landed (fetched during stage C - 1)
            waitl<0>();
            __builtin_amdgcn_s_barrier();
        } else {
            waitl<0>();
        }
        const int an = va + NC * STG + NU * 16384, bn = vb + NC * STG + NU * 16384;
        sfor<0, 4>([&](auto gi) {
            constexpr int g = decltype(gi)::value;
            SB();
            sfor<0, 4>([&](auto ji) {
                constexpr int j = decltype(ji)::value;
                acc[g][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[B][g], fb[B][j], acc[g][j], 0, 0, 0);
                if constexpr (g == 0) dsr<j * 1024>(fb[B ^ 1][j], bn);
                if constexpr (g == 1) dsr<j * 1024>(fa[B ^ 1][j], an);
                if constexpr (j == 1) sfor<0, P>([&](auto qi) {
                    constexpr int q = decltype(qi)::value;
                    const unsigned from = SHARE == 0 ? piece : (q < (P + 1) / 2 ? pa : pb) + (piece - (blockIdx.x * 37u + wave * P) * 1024u);
                    if constexpr (q % 4 == g) ldma(voff, rs, (from + q * 1024u) & (span - 1), lbase + F * STG + U * 16384 + (wave * P + q) % 16 * 1024);
                });
            });
            SB();
        });
        piece += (unsigned)NWV * P * 1024u;
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s + 3 * KS <= steps; s += 3 * KS) {
        sfor<0, 3>([&](auto ci) { sfor<0, KS>([&](auto ui) { sub(ci, ui); }); });
    }
    waitv<0>();
    waitl<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int x = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) x += acc[i][j][q];
    if (x == 0x12345677) out[0] = (float)x;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int P, int KS, int NWV, int SHARE = 0> static void run_ks(const char* src, unsigned span) {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
    const int steps = 768;
    const int lds = 3 * KS * 16384 > 96 * 1024 ? 3 * KS * 16384 : 96 * 1024;          // (one workgroup a compute unit)
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_ks<P, KS, NWV, SHARE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop_ks<P, KS, NWV, SHARE>), 256, 64 * NWV, lds, 0, src, span, steps, out, clk);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((loop_ks<P, KS, NWV, SHARE>), 256, 64 * NWV, lds, 0, src, span, steps, out, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    printf("%s span %u MiB: barrier every %d K-steps, %d waves a workgroup, %d pieces a wave-step:  %6.1f cycles per K-step of a wave   launch %7.1f us = %6.1f ns per K-step (MFMA bound: %d clocks per SIMD-step)\n",
           SHARE ? "shared streams," : "private streams,", span >> 20, KS, NWV, P, (double)c / steps, ms * 1e3, ms * 1e6 / steps, 256 * NWV / 4);
    CK(hipFree(out)); CK(hipFree(clk));
}

template <int DMA, int P, int PLACE> static void run_pf(const char* src, unsigned span, int wgs) {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
    const int steps = 768;
    const int lds = wgs == 1 ? 96 * 1024 : 64 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_pf<DMA, P, PLACE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop_pf<DMA, P, PLACE>), 256 * wgs, 256, lds, 0, src, span, steps, out, clk);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((loop_pf<DMA, P, PLACE>), 256 * wgs, 256, lds, 0, src, span, steps, out, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    printf("prefetched fragments (placement %d), LDS-DMA %d  pieces/wave-step %d  wgs/cu %d:  %6.1f cycles per K-step   launch %7.1f us = %6.1f ns per K-step\n", PLACE, DMA, P, wgs,
           (double)c / steps, ms * 1e3, ms * 1e6 / steps);
    CK(hipFree(out)); CK(hipFree(clk));
}

template <int MODE, int P> static void run(const char* src, unsigned span, int wgs) {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
    const int steps = 768;
    // one workgroup per compute unit: ask for more LDS than two could share; two per compute unit: 48 KiB each
    const int lds = wgs == 1 ? 96 * 1024 : 48 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<MODE, P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop<MODE, P>), 256 * wgs, 256, lds, 0, src, span, steps, out, clk);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((loop<MODE, P>), 256 * wgs, 256, lds, 0, src, span, steps, out, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    printf("mode %d  pieces/wave-step %d  wgs/cu %d:  %6.1f cycles per K-step   launch %7.1f us = %6.1f ns per K-step   %5.2f TB/s fetched\n", MODE, P, wgs,
           (double)c / steps, ms * 1e3, ms * 1e6 / steps, (double)P * 4 * 1024 * steps * 256 * wgs / (ms * 1e-3) / 1e12);
    CK(hipFree(out)); CK(hipFree(clk));
}

int main() {
    const unsigned span = 4u << 20;                                          // 4 MiB: stays in every XCD's L2
    char* src; CK(hipMalloc(&src, span + 4096)); CK(hipMemset(src, 1, span + 4096));
    for (int wgs = 1; wgs <= 2; ++wgs) {
        run<0, 0>(src, span, wgs);
        run<1, 2>(src, span, wgs); run<2, 2>(src, span, wgs); run<3, 2>(src, span, wgs);
        run<1, 4>(src, span, wgs); run<2, 4>(src, span, wgs); run<3, 4>(src, span, wgs);
        run<1, 6>(src, span, wgs); run<2, 6>(src, span, wgs); run<3, 6>(src, span, wgs);
    }
    for (int wgs = 1; wgs <= 2; ++wgs) {
        run_pf<0, 0, 0>(src, span, wgs); run_pf<1, 4, 0>(src, span, wgs); run_pf<1, 6, 0>(src, span, wgs);
        run_pf<0, 0, 1>(src, span, wgs); run_pf<1, 4, 1>(src, span, wgs); run_pf<1, 6, 1>(src, span, wgs);
        run_pf<1, 4, 2>(src, span, wgs); run_pf<1, 6, 2>(src, span, wgs);
    }
    run_ks<4, 1, 4>(src, span); run_ks<4, 2, 4>(src, span);
    run_ks<3, 1, 8>(src, span); run_ks<3, 2, 8>(src, span); run_ks<2, 2, 8>(src, span);
    // the GEMM's sharing pattern, and a source larger than the L2s (64 MiB: from the memory-side cache)
    run_ks<3, 1, 8, 1>(src, span); run_ks<4, 1, 8, 1>(src, span);
    char* big; const unsigned bspan = 64u << 20; CK(hipMalloc(&big, bspan + 4096)); CK(hipMemset(big, 1, bspan + 4096));
    run_ks<3, 1, 8, 0>(big, bspan); run_ks<3, 1, 8, 1>(big, bspan); run_ks<4, 1, 8, 1>(big, bspan); run_ks<4, 1, 4, 1>(big, bspan);
    return 0;
}
