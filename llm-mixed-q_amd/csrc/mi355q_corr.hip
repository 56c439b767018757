// mi355q_corr.hip -- the static half of the producer-formed exception corrections (mi355q_corr.h): a weight operand's
// CORRECTION PLAN.  One 256-thread workgroup per 256-row bucket of W's exception list sorts the bucket's entries by
// (row, block) -- the order their products are summed in, a property of the data, not of the order in which rows reserved
// their list slots: reproducible --, numbers the distinct rows (column slots of the product's tile) and writes the tile's
// column map.  Runs once per packed weight (mi355q_bfp_corr_plan); nothing of it is on the timed path.
// Reference: quantized_modules/linear.py:59-76 (the exception blocks are part of W_q like every other block).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi355q.h"
#include "mi355q_corr.h"

namespace mi355q {

__global__ __launch_bounds__(256) void corr_plan_kernel(const int* __restrict__ wlist, long long N, int* __restrict__ plan) {
    __shared__ int key[ROW_BCAP];          // (row - n0) << 10 | block, or a large value for a void entry
    __shared__ int nfirst, bad;
    const int tid = threadIdx.x, b = blockIdx.x;
    const long long n0 = (long long)b * ROW_BUCKET_ROWS;
    const int* bucket = wlist + EXC_HEADER + (long long)b * row_bucket_words(ROW_BCAP);
    const int reserved = bucket[0], cw = min(reserved, ROW_BCAP);
    int* colmap = plan + plan_colmap_off() + n0;
    int* slots = plan + plan_slots_off(N) + (long long)b * CORR_WV * 4;
    int* sorted = plan + plan_entries_off(N) + (long long)b * ROW_BCAP * EXC_ENTRY;
    if (tid == 0) { nfirst = 0; bad = (reserved > ROW_BCAP || wlist[0] != 0) ? 1 : 0; }
    colmap[tid] = -1;
    if (tid < CORR_WV * 4) slots[tid] = (tid & 3) == 0 ? -1 : 0;
    int e[EXC_ENTRY];
    const bool mine = tid < cw;
    if (mine) {
#pragma unroll
        for (int q = 0; q < EXC_ENTRY; ++q) e[q] = bucket[EXC_HEADER + EXC_ENTRY * tid + q];
    }
    const bool valid = mine && e[0] >= n0 && e[0] < n0 + ROW_BUCKET_ROWS && e[0] < N && e[1] >= 0 && e[1] < 1024;
    if (tid < ROW_BCAP) key[tid] = valid ? (int)((e[0] - n0) << 10 | e[1]) : 0x7fffffff;
    __syncthreads();
    if (mine && !valid) bad = 1;           // (a void entry: its row kept its own exponents -- the row-scale product does not apply)
    __shared__ int isfirst[ROW_BCAP];
    int rank = 0, same_before = 0, same = 0, firsts_before = 0;
    const int k = valid ? key[tid] : 0x7fffffff;
    if (valid) {
        for (int j = 0; j < cw; ++j) {
            const int kj = key[j];
            if (kj == 0x7fffffff) continue;
            const bool before = kj < k || (kj == k && j < tid);
            rank += before ? 1 : 0;
            const bool same_row = (kj >> 10) == (k >> 10);
            same += same_row ? 1 : 0;
            same_before += same_row && before ? 1 : 0;
        }
    }
    if (tid < ROW_BCAP) isfirst[tid] = valid && same_before == 0 ? 1 : 0;
    __syncthreads();
    if (valid) {
        for (int j = 0; j < cw; ++j)       // rows in front of this one
            firsts_before += isfirst[j] && (key[j] >> 10) < (k >> 10) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < EXC_ENTRY; ++q) sorted[rank * EXC_ENTRY + q] = e[q];
        if (same_before == 0) {                                 // the first entry of its row: the row's column slot
            atomicAdd(&nfirst, 1);
            if (firsts_before < CORR_WV) {
                colmap[e[0] - n0] = firsts_before;
                slots[firsts_before * 4 + 0] = e[0];
                slots[firsts_before * 4 + 1] = rank;
                slots[firsts_before * 4 + 2] = same;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        plan[plan_ncols_off(N) + b] = min(nfirst, CORR_WV);
        if (nfirst > CORR_WV || bad) atomicOr(&plan[0], 1);
        atomicAdd(&plan[1], min(nfirst, CORR_WV));
    }
}

__global__ void corr_plan_header_kernel(int* __restrict__ plan, long long N) {
    const int t = threadIdx.x;
    if (t < PLAN_HDR) plan[t] = t == 2 ? (int)plan_nb(N) : t == 3 ? (int)N : 0;
}

// the dense records (mi355q_corr.h): slot (b, c) in use -> record (column slots in use of the buckets in front of b) + c
__global__ __launch_bounds__(256) void corr_plan_dense_kernel(int* __restrict__ plan, long long N) {
    const int nb = (int)plan_nb(N);
    const int* ncols = plan + plan_ncols_off(N);
    const int* slots = plan + plan_slots_off(N);
    const int* entries = plan + plan_entries_off(N);
    int* dense = plan + plan_dense_off(N);
    for (int idx = threadIdx.x; idx < nb * CORR_WV; idx += 256) {
        const int b = idx / CORR_WV, cs = idx % CORR_WV;
        const int n = slots[idx * 4], first = slots[idx * 4 + 1], cnt = slots[idx * 4 + 2];
        if (n < 0) continue;
        int pos = cs;
        for (int q = 0; q < b; ++q) pos += ncols[q];
        const int* e = entries + ((long long)b * ROW_BCAP + first) * EXC_ENTRY;
        int* d = dense + (long long)pos * 8;
        d[0] = idx; d[1] = e[1]; d[2] = e[2]; d[3] = ((cnt - 1) << 16) | first;
        d[4] = e[4]; d[5] = e[5]; d[6] = e[6]; d[7] = e[7];
    }
}

int launch_corr_plan(const int* wlist, long long N, int* plan, hipStream_t st) {
    hipLaunchKernelGGL(corr_plan_header_kernel, 1, 64, 0, st, plan, N);
    hipLaunchKernelGGL(corr_plan_kernel, (unsigned)plan_nb(N), 256, 0, st, wlist, N, plan);
    hipLaunchKernelGGL(corr_plan_dense_kernel, 1, 256, 0, st, plan, N);
    return (int)hipGetLastError();
}

}  // namespace mi355q
