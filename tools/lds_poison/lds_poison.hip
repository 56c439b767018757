// Debug aid: fill the LDS of every compute unit with a pattern, so that a kernel launched afterwards that reads LDS it
// never wrote shows up as a result that depends on the pattern (LDS keeps the previous kernel's contents).
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC lds_poison.hip -o liblds_poison.so
#include <hip/hip_runtime.h>
extern "C" __global__ void lds_fill(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds[];
    const int n = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    if (sink && lds[(threadIdx.x * 977) % n] != pattern) *sink = 1;       // (keeps the stores alive)
}
extern "C" int lds_poison(unsigned pattern, void* stream) {
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_fill), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        once = true;
    }
    hipLaunchKernelGGL(lds_fill, 1024, 1024, 160 * 1024, static_cast<hipStream_t>(stream), pattern, nullptr);
    return (int)hipGetLastError();
}

// What a kernel that reads LDS without writing it sees: word `7 + 64 * workgroup` of each of `n` workgroups' LDS.
extern "C" __global__ void lds_peek_kernel(unsigned* out) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) out[blockIdx.x] = lds[7 + 64 * (blockIdx.x % 512)];
}
extern "C" int lds_peek(unsigned* out, int n, void* stream) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_peek_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    hipLaunchKernelGGL(lds_peek_kernel, n, 64, 160 * 1024, static_cast<hipStream_t>(stream), out);
    return (int)hipGetLastError();
}
