// mi355q_align.h -- layout of the exception lists of an aligned operand (mi355q_align_row.h).  (The alignment of 256-value
// groups that used to live here -- the int32-chain kernel's operand format -- was removed in round 5 with that kernel.)
// Exception list layout (int32): [0] count / overflow word, [1..7] spare, then entries of 8 words
// {row (-1: void), block, exponent code, 0, 4 dwords of mantissas}.
#ifndef MI355Q_ALIGN_H
#define MI355Q_ALIGN_H
#include <hip/hip_runtime.h>

namespace mi355q {

constexpr int EXC_HEADER = 8, EXC_ENTRY = 8;   // int32 words

}  // namespace mi355q
#endif
