// mi355q_gemm_v9g.hip -- the GATED epilogue of the 256 x 256 tile kernel (round 6, mi355q_bfp_gemm_aligned_gated): mi355q_gemm_v9.hip
// compiled as a translation unit of its own with V9_GATED_TU defined (see the note at the top of that file) -- x against the
// interleaved gate / up weights of a gated MLP, silu(gate) * up and the consumer's block_fp quantiser in the store epilogue, the
// consumer's tiled bf16 operand as the only output.
#define V9_GATED_TU 1
#include "mi355q_gemm_v9.hip"
