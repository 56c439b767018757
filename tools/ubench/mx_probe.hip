// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with FP6 (e2m3) operands and E8M0 block scales: which element of the
// contraction sits where in the 192-bit operand, which lane's scale applies to which 32-group.  One wave; host check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void k(const uint32_t* a, const uint32_t* b, const uint32_t* sa, const uint32_t* sb, float* out) {
    const int lane = threadIdx.x;
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = (int)a[lane * 8 + i]; vb[i] = (int)b[lane * 8 + i]; }
    f32x4 c = {0, 0, 0, 0};
    // cbsz = 2 (A: fp6 e2m3), blgp = 2 (B: fp6 e2m3); opsel 0: scale byte 0 of the scale VGPRs
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 2, 2, 0, (int)sa[lane], 0, (int)sb[lane]);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}

static float fp6_e2m3(unsigned v) {          // sign, 2-bit exponent (bias 1), 3-bit mantissa
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    const float mag = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * std::ldexp(1.0f, e - 1);
    return s ? -mag : mag;
}

int main() {
    // logical operands: A[16 rows][128 k], B[16 cols][128 k] as 6-bit codes; scales per (row, 32-group)
    std::vector<unsigned> A(16 * 128), B(16 * 128), SA(16 * 4), SB(16 * 4);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (auto& v : A) v = rnd() & 63;
    for (auto& v : B) v = rnd() & 63;
    for (auto& v : SA) v = 124 + (rnd() % 7);      // E8M0: 2^(v - 127)
    for (auto& v : SB) v = 125 + (rnd() % 5);
    // hypothesis: lane l holds row (l & 15), k-group g = l >> 4 (32 values k = 32 g .. 32 g + 31), value t at bits [6 t, 6 t + 6)
    // of the first 6 dwords; the lane's scale byte 0 = scale of (row, group g)
    std::vector<uint32_t> ha(64 * 8, 0), hb(64 * 8, 0), hsa(64), hsb(64);
    for (int l = 0; l < 64; ++l) {
        const int r = l & 15, g = l >> 4;
        for (int t = 0; t < 32; ++t) {
            const unsigned ca = A[r * 128 + g * 32 + t], cb = B[r * 128 + g * 32 + t];
            const int bit = 6 * t;
            ha[l * 8 + bit / 32] |= ca << (bit % 32);
            if (bit % 32 > 26) ha[l * 8 + bit / 32 + 1] |= ca >> (32 - bit % 32);
            hb[l * 8 + bit / 32] |= cb << (bit % 32);
            if (bit % 32 > 26) hb[l * 8 + bit / 32 + 1] |= cb >> (32 - bit % 32);
        }
        hsa[l] = SA[r * 4 + g];
        hsb[l] = SB[r * 4 + g];
    }
    uint32_t *da, *db, *dsa, *dsb; float* dout;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 1024);
    hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, da, db, dsa, dsb, dout);
    std::vector<float> out(256);
    hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
    // reference: C[m][n] = sum_g 2^(sa[m][g]-127) 2^(sb[n][g]-127) sum_t a[m][32g+t] b[n][32g+t];  output layout of the
    // 16x16 f32 MFMA: lane l, register i -> row 4 (l >> 4) + i, column l & 15
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int m = 4 * (l >> 4) + i, n = l & 15;
            double ref = 0;
            for (int g = 0; g < 4; ++g) {
                double s = 0;
                for (int t = 0; t < 32; ++t) s += (double)fp6_e2m3(A[m * 128 + g * 32 + t]) * fp6_e2m3(B[n * 128 + g * 32 + t]);
                ref += s * std::ldexp(1.0, (int)SA[m * 4 + g] - 127) * std::ldexp(1.0, (int)SB[n * 4 + g] - 127);
            }
            maxerr = std::fmax(maxerr, std::fabs(ref - out[l * 4 + i]));
            maxref = std::fmax(maxref, std::fabs(ref));
        }
    printf("mx probe: max |err| %.4g of max |ref| %.4g  (out[0..3] = %g %g %g %g)\n", maxerr, maxref, out[0], out[1], out[2], out[3]);
    return 0;
}
