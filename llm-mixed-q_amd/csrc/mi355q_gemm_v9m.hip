// mi355q_gemm_v9m.hip -- the MIXED contraction of the 256 x 256 tile kernel (round 6, mi355q_bfp_gemm_mixed): mi355q_gemm_v9.hip
// compiled as a translation unit of its own with V9_MIXED_TU defined (see the note at the top of that file) -- class 0 of the
// columns on the int8 MFMA, its int32 sums turned into fp32 in place, class 1 on the bf16 MFMA in the same registers.
#define V9_MIXED_TU 1
#include "mi355q_gemm_v9.hip"
