// step_driver.cpp -- the benchmark step (bench.py: x[4096,4096] fp32 -> fused quantise + row-align -> row-scale int8
// GEMM vs pre-packed W -> y fp32) through the C ABI only, no Python: a short-lived program for rocprofv3 counter
// (--pmc) passes, which need few kernels and no interpreter start-up.  Also the smallest complete C example of the
// boundary (include/mi355q.h).      build: make -C tools/cdriver        run: tools/cdriver/step_driver [steps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "mi355q.h"

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define Q_OK(e) do { int rc_ = (e); if (rc_) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, mi355q_error_string(rc_)); return 3; } } while (0)

template <class T> static T* dalloc(size_t n, bool zero = false) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) + 16) != hipSuccess) std::abort();
    if (zero && hipMemset(p, 0, n * sizeof(T)) != hipSuccess) std::abort();
    return static_cast<T*>(p);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 3;
    if (argc > 2 && std::string(argv[2]) == "matmul") {      // attention-shaped fused quantise + matmul instead of the Linear step
        const int64_t B = 12, M = 2048, K = 2048, N = 64;
        std::mt19937 gen(1);
        std::uniform_real_distribution<float> ud(0.f, 1.f);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> hx(B * M * K), hy(B * K * N);
        for (auto& v : hx) v = ud(gen);
        for (auto& v : hy) v = nd(gen);
        float *x = dalloc<float>(hx.size()), *y = dalloc<float>(hy.size()), *out = dalloc<float>(B * M * N);
        void* ws = dalloc<unsigned char>(mi355q_bfp_matmul_workspace_bytes(B, K, N));
        HIP_OK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
        for (int s = 0; s < steps; ++s) Q_OK(mi355q_bfp_matmul(x, y, out, ws, B, M, K, N, 6, 8, 127, 6, 8, 127, nullptr));
        HIP_OK(hipDeviceSynchronize());
        float h[4];
        HIP_OK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        std::printf("step_driver matmul: %d calls, out[0..3] = %g %g %g %g\n", steps, h[0], h[1], h[2], h[3]);
        return 0;
    }
    if (argc > 2 && std::string(argv[2]) == "quantizers") {  // BASELINE config 5: the streaming fake-quantisers at Llama-7B shapes
        const int64_t shapes[2][2] = {{2048, 4096}, {2048, 11008}};
        std::mt19937 gen(7);
        std::normal_distribution<float> nd(0.f, 4.f);
        void* ws = dalloc<unsigned char>(MI355Q_WORKSPACE_BYTES, true);
        for (const auto& shp : shapes) {
            const int64_t rows = shp[0], cols = shp[1];
            std::vector<float> hx(rows * cols);
            for (auto& v : hx) v = nd(gen);
            float *x = dalloc<float>(hx.size()), *y = dalloc<float>(hx.size());
            HIP_OK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
            for (int s = 0; s < steps; ++s) {
                Q_OK(mi355q_block_fp_quantize(x, y, nullptr, nullptr, 1, rows, cols, 1, 16, 6, 8, 127, MI355Q_ZERO_BLOCK_FAST, ws, nullptr));
                Q_OK(mi355q_block_minifloat_quantize(x, y, nullptr, 1, rows, cols, 1, 16, 8, 4, 8, MI355Q_ZERO_BLOCK_FAST, ws, nullptr));
                Q_OK(mi355q_block_log_quantize(x, y, nullptr, 1, rows, cols, 1, 16, 8, 8, 0, ws, nullptr));
            }
            HIP_OK(hipDeviceSynchronize());
            float h[4];
            HIP_OK(hipMemcpy(h, y, 16, hipMemcpyDeviceToHost));
            std::printf("step_driver quantizers [%lld, %lld]: %d rounds of block_fp / block_minifloat / block_log, y[0..3] = %g %g %g %g\n",
                        (long long)rows, (long long)cols, steps, h[0], h[1], h[2], h[3]);
            HIP_OK(hipFree(x));
            HIP_OK(hipFree(y));
        }
        return 0;
    }
    if (argc > 2 && std::string(argv[2]) == "attention") {   // the one-pass attention core: [heads] [head_dim], T = 2048, causal
        const int64_t B = argc > 3 ? std::atoi(argv[3]) : 12, T = 2048, D = argc > 4 ? std::atoi(argv[4]) : 64;
        std::mt19937 gen(1);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> hq(B * T * D), hk(B * T * D), hv(B * T * D);
        for (auto& v : hq) v = nd(gen);
        for (auto& v : hk) v = nd(gen);
        for (auto& v : hv) v = nd(gen);
        float *q = dalloc<float>(hq.size()), *k = dalloc<float>(hk.size()), *v = dalloc<float>(hv.size()), *out = dalloc<float>(hq.size());
        void* ws = dalloc<unsigned char>(mi355q_bfp_attention_workspace_bytes(B, T, D));
        HIP_OK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(k, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(v, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
        const int32_t par[6] = {6, 8, 127, 6, 8, 127};
        for (int s = 0; s < steps; ++s)
            Q_OK(mi355q_bfp_attention(q, k, v, nullptr, 1, D == 128 ? std::sqrt(128.f) : 0.f, out, ws, B, T, T, D, par, par, nullptr));
        HIP_OK(hipDeviceSynchronize());
        float h[4];
        HIP_OK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        std::printf("step_driver attention [%lld, %lld, %lld]: %d calls, out[0..3] = %g %g %g %g\n", (long long)B, (long long)T, (long long)D, steps, h[0], h[1], h[2], h[3]);
        return 0;
    }
    const int64_t M = 4096, N = 4096, K = 4096;
    std::mt19937 gen(0);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hx(M * K), hw(N * K), hb(N);
    for (int64_t r = 0; r < M; ++r) { const float s = std::exp(nd(gen)); for (int64_t k = 0; k < K; ++k) hx[r * K + k] = nd(gen) * s; }
    // (the distributions of bench.py's make_inputs: rows of x scaled by exp(N(0,1)), weights and bias N(0, 0.02^2))
    for (auto& v : hw) v = nd(gen) * 0.02f;
    for (auto& v : hb) v = nd(gen) * 0.02f;
    float *x = dalloc<float>(M * K), *w = dalloc<float>(N * K), *bias = dalloc<float>(N), *y = dalloc<float>(M * N);
    HIP_OK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    void* ws = dalloc<unsigned char>(MI355Q_WORKSPACE_BYTES, true);
    hipStream_t st = nullptr;

    // weights once: quantise in place + pack, row-align, bias quantise (linear.py:66-70)
    int8_t* wmant = dalloc<int8_t>(N * K);
    uint8_t* wexp = dalloc<uint8_t>(N * K / 16);
    Q_OK(mi355q_block_fp_quantize(w, w, wmant, wexp, 1, N, K, 1, 16, 6, 8, 127, MI355Q_ZERO_BLOCK_FAST, ws, st));
    Q_OK(mi355q_block_fp_quantize(bias, bias, nullptr, nullptr, 1, 1, N, 1, 16, 6, 8, 127, MI355Q_ZERO_BLOCK_EXACT, ws, st));
    int8_t* wt = dalloc<int8_t>(mi355q_bfp_tiled_bytes(N, K), true);
    uint8_t *we = dalloc<uint8_t>(N * K / 16), *wflag = dalloc<uint8_t>(N);
    float* wscale = dalloc<float>(mi355q_bfp_rows_pad(N), true);
    int32_t* wlist = dalloc<int32_t>(mi355q_bfp_row_list_bytes(N, 0) / 4, true);
    Q_OK(mi355q_bfp_align_rows(wmant, wexp, wt, we, wflag, wscale, wlist, 127 + 5, N, K, 0, st));

    // activations every step: fused quantise + pack + row-align + tile, then the GEMM
    int8_t* xt = dalloc<int8_t>(mi355q_bfp_tiled_bytes(M, K), true);
    uint8_t *xe = dalloc<uint8_t>(M * K / 16), *xflag = dalloc<uint8_t>(M);
    float* xscale = dalloc<float>(mi355q_bfp_rows_pad(M), true);
    int32_t* xlist[2] = {dalloc<int32_t>(mi355q_bfp_row_list_bytes(M, 0) / 4, true), dalloc<int32_t>(mi355q_bfp_row_list_bytes(M, 0) / 4, true)};
    for (int s = 0; s < steps; ++s) {
        Q_OK(mi355q_block_fp_quantize_aligned_rows(x, xt, xe, xflag, xscale, xlist[s & 1], xlist[(s + 1) & 1], M, K, 6, 8, 127, 0, st));
        mi355q_bfp_operand xo{xt, xe, xflag, xscale, xlist[s & 1], 0, 5, 127, 1};
        mi355q_bfp_operand wo{wt, we, wflag, wscale, wlist, 0, 5, 127, 1};
        Q_OK(mi355q_bfp_gemm_aligned(&xo, &wo, bias, y, M, N, K, N, st));
    }
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> hy(8);
    HIP_OK(hipMemcpy(hy.data(), y, 32, hipMemcpyDeviceToHost));
    std::printf("step_driver: %d steps, y[0..3] = %g %g %g %g\n", steps, hy[0], hy[1], hy[2], hy[3]);
    return 0;
}
