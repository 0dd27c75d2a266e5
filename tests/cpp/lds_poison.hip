// Test helper: fills the LDS of every CU with a bit pattern (NaN, Inf, zero) so that a following kernel finds those
// leftovers wherever it has not written itself.  tests/test_gpu_lds_leftovers.py runs the engine after different patterns
// and demands identical results: no kernel may depend on LDS it did not initialise (round 3: a pad scalar between two LDS
// regions was read as "garbage times exact zero", which is NaN when the garbage is).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void lds_fill_kernel(uint32_t pattern, int dwords, uint32_t* sink) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < dwords; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    // read something back so that the stores cannot be dropped
    if (threadIdx.x == 0 && lds[(blockIdx.x * 131) % dwords] != pattern) sink[0] = 1u;
}

extern "C" int lds_poison(uint32_t pattern) {
    const int bytes = 160 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return 1;
    uint32_t* sink = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&sink), 4) != hipSuccess) return 2;
    (void)hipMemset(sink, 0, 4);
    // one workgroup owns the whole LDS of a CU; several per CU in turn so that every CU is visited
    hipLaunchKernelGGL(lds_fill_kernel, dim3(256 * 4), dim3(256), bytes, 0, pattern, bytes / 4, sink);
    const hipError_t err = hipDeviceSynchronize();
    (void)hipFree(sink);
    return err == hipSuccess ? 0 : 3;
}
