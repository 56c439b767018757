// mi355q_gemm_v9r.hip -- the row-scale int8 product of the 256 x 256 tile kernel WITH the caller's residual add in its one-pass store
// epilogue (round 6, mi355q_bfp_gemm_aligned_res): mi355q_gemm_v9.hip compiled as a translation unit of its own with V9_RESID_TU defined
// (see the note at the top of that file) -- y = (x . w^T + bias) + residual, the same bits as the separate add.
#define V9_RESID_TU 1
#include "mi355q_gemm_v9.hip"
