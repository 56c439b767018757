// mi355q_pack.hip -- true w-bit storage of block_fp weights (the reference's memory-density accounting,
// quantized_layer_profiler.py:18-27: width bits per value + exponent_width bits per block).
//
// At rest a weight operand is  {packed mantissas: rows x K x width bits, a dense little-endian bit string per row of
// width-bit two's-complement values;  one code byte per 16-block}  = width + 0.5 bits per value (6.5 at W6, 4.5 at W4).
// The GEMM kernels stream 1-KiB tiled pieces of int8 (row-scale flavour) or bf16 (per-block-exponent flavour); the
// expansion into a scratch operand is a pure streaming kernel -- no reductions, no decisions: for the row-scale flavour
// the code byte is the block's left shift onto its row's exponent (0xFF: an exception block, zero in the operand and kept
// in the row's exception list), for the bf16 flavour it is the block's biased exponent.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_internal.h"

namespace mi355q {

// one thread per 16-block: 16 int8 -> 16 * width bits
__global__ __launch_bounds__(256) void bfp_pack_bits_kernel(const int8_t* __restrict__ mant, uint16_t* __restrict__ out,
                                                            long long nblocks, int width) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const int4 raw = reinterpret_cast<const int4*>(mant)[b];
    const int w[4] = {raw.x, raw.y, raw.z, raw.w};
    const unsigned mask = (1u << width) - 1u;
    unsigned long long lo = 0ull, hi = 0ull;            // 128-bit string, value i at bit i * width
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned v = ((unsigned)(w[i >> 2] >> (8 * (i & 3)))) & mask;
        const int pos = i * width;
        if (pos < 64) {
            lo |= (unsigned long long)v << pos;
            if (pos + width > 64) hi |= (unsigned long long)v >> (64 - pos);
        } else {
            hi |= (unsigned long long)v << (pos - 64);
        }
    }
    uint16_t* o = out + b * width;                       // 16 * width bits = width halfwords
    for (int h = 0; h < width; ++h) o[h] = (uint16_t)((h < 4 ? lo >> (16 * h) : hi >> (16 * (h - 4))) & 0xFFFFu);
}

__device__ __forceinline__ int unpack_value(unsigned long long lo, unsigned long long hi, int i, int width) {
    const int pos = i * width;
    unsigned long long v = pos < 64 ? lo >> pos : hi >> (pos - 64);
    if (pos < 64 && pos + width > 64) v |= hi << (64 - pos);
    const int s = 32 - width;
    return ((int)((unsigned)v << s)) >> s;               // sign-extend the low `width` bits
}

// MODE 0: int8 row-scale operand (tiled, shifted; code 0xFF -> zeros).  MODE 1: tiled bf16 values m * 2^(code - off).
// One thread = 16 output bytes; 64 consecutive threads = one 1-KiB piece, written contiguously.
// WIDTH_: the width as a compile-time constant (2 .. 8: every shift below an immediate, the block's bit string fetched
// with 32-bit loads where it is 4-byte aligned), 0: taken from the argument.
template <int MODE, int WIDTH_>
__global__ __launch_bounds__(256) void bfp_expand_kernel(const uint16_t* __restrict__ packed, const uint8_t* __restrict__ codes,
                                                         unsigned char* __restrict__ out, long long rows, long long K, int width_arg,
                                                         int off, long long npieces, const uint8_t* __restrict__ rowexp,
                                                         uint8_t* __restrict__ exp_out) {
    const int width = WIDTH_ ? WIDTH_ : width_arg;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long piece = g >> 6;
    if (piece >= npieces) return;
    const int t = (int)(g & 63), c = t >> 4, r = t & 15;
    const long long kp = MODE == 0 ? K >> 6 : K >> 5;    // pieces per 16 rows (64 int8 or 32 bf16 per row and piece)
    const long long row = (piece / kp) * 16 + r;
    const long long kb = MODE == 0 ? (piece % kp) * 4 + c : (piece % kp) * 2 + (c >> 1);   // 16-block along K
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (row < rows) {
        const long long blk = row * (K >> 4) + kb;
        const uint16_t* p = packed + blk * width;
        unsigned long long lo = 0ull, hi = 0ull;
        if (WIDTH_ && WIDTH_ % 2 == 0) {                  // (blk * WIDTH_ halfwords: a multiple of 4 bytes)
            const unsigned* p32 = reinterpret_cast<const unsigned*>(p);
#pragma unroll
            for (int d = 0; d < WIDTH_ / 2; ++d) {
                const unsigned long long dw = p32[d];
                if (d < 2) lo |= dw << (32 * d); else hi |= dw << (32 * (d - 2));
            }
        } else {
#pragma unroll
            for (int h = 0; h < (WIDTH_ ? WIDTH_ : 8); ++h) {
                if (h < width) {
                    const unsigned long long hw = p[h];
                    if (h < 4) lo |= hw << (16 * h); else hi |= hw << (16 * (h - 4));
                }
            }
        }
        const int code = codes[blk];
        // (the aligned operand's per-block exponents: its row's -- the four blocks of this row and piece as one dword by the first
        //  of their four lanes; K % 64 == 0, so blk is a multiple of four there)
        if (MODE == 0 && exp_out && c == 0) *reinterpret_cast<unsigned*>(exp_out + blk) = 0x01010101u * rowexp[row];
        if (MODE == 0) {
            if (code != 0xFF) {
                unsigned wds[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned acc = 0u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc |= ((unsigned)(unpack_value(lo, hi, 4 * q + i, width) << code) & 0xFFu) << (8 * i);
                    wds[q] = acc;
                }
                o = make_uint4(wds[0], wds[1], wds[2], wds[3]);
            }
        } else {
            const int half = c & 1;                       // values 8 half .. 8 half + 7 of the block
            unsigned wds[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float f0 = __builtin_ldexpf((float)unpack_value(lo, hi, 8 * half + 2 * q, width), code - off);
                const float f1 = __builtin_ldexpf((float)unpack_value(lo, hi, 8 * half + 2 * q + 1, width), code - off);
                wds[q] = (__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xFFFF0000u);      // exact in bf16: truncation
            }
            o = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
    }
    reinterpret_cast<uint4*>(out)[g] = o;
}

int launch_bfp_pack_bits(const int8_t* mant, uint16_t* out, long long rows, long long K, int width, hipStream_t st) {
    const long long nb = rows * (K >> 4);
    hipLaunchKernelGGL(bfp_pack_bits_kernel, (unsigned)((nb + 255) / 256), 256, 0, st, mant, out, nb, width);
    return (int)hipGetLastError();
}

int launch_bfp_expand(int mode, const uint16_t* packed, const uint8_t* codes, void* out, long long rows, long long K, int width,
                      int off, hipStream_t st, const uint8_t* rowexp, uint8_t* exp_out) {
    const long long rows_pad = (rows + 127) / 128 * 128;
    const long long npieces = (rows_pad >> 4) * (mode == 0 ? K >> 6 : K >> 5);
    const unsigned grid = (unsigned)((npieces * 64 + 255) / 256);
#define MI355Q_EXPAND(M, W) hipLaunchKernelGGL((bfp_expand_kernel<M, W>), grid, 256, 0, st, packed, codes, static_cast<unsigned char*>(out), rows, K, width, off, npieces, rowexp, exp_out)
#define MI355Q_EXPAND_W(M)                                                                                      \
    switch (width) {                                                                                            \
        case 4: MI355Q_EXPAND(M, 4); break;                                                                     \
        case 5: MI355Q_EXPAND(M, 5); break;                                                                     \
        case 6: MI355Q_EXPAND(M, 6); break;                                                                     \
        case 8: MI355Q_EXPAND(M, 8); break;                                                                     \
        default: MI355Q_EXPAND(M, 0); break;                                                                    \
    }
    if (mode == 0) { MI355Q_EXPAND_W(0) } else { MI355Q_EXPAND_W(1) }
#undef MI355Q_EXPAND_W
#undef MI355Q_EXPAND
    return (int)hipGetLastError();
}

}  // namespace mi355q
