// LDS allocation granule of the device (run on the GPU box): occupancy API for a one-wave workgroup with small register needs,
// and the measured number of co-resident workgroups per CU (each workgroup spins until a flag count is reached).
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_granule.hip -o tools/build/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float dyn[];
__global__ __launch_bounds__(64) void k(float* out) { dyn[threadIdx.x] = threadIdx.x; __syncthreads(); if (out) out[threadIdx.x] = dyn[63 - threadIdx.x]; }
// residency probe: every workgroup records (CU id, arrival order on that CU) and waits ~200 us so that all co-resident
// workgroups overlap; the maximum simultaneous count per CU is read from a per-CU counter
__global__ __launch_bounds__(64) void resident(int* cu_now, int* cu_max, int spin) {
    dyn[threadIdx.x] = 0.f;
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int cu = ((xcc & 15) << 6) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15);   // xcc, se, cu
    if (threadIdx.x == 0) {
        const int now = atomicAdd(&cu_now[cu], 1) + 1;
        atomicMax(&cu_max[cu], now);
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
        atomicSub(&cu_now[cu], 1);
    }
}
int main() {
    for (int lds : {6784, 7488, 7680, 7681, 12288, 12800, 12801, 13120, 13312, 13653, 13824, 14080, 14081, 14912, 15360}) {
        int n = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, lds);
        printf("occupancy API: dynamic LDS %6d B -> %d workgroups per CU\n", lds, n);
    }
    int *now, *mx; (void)hipMalloc(&now, 4096 * 4); (void)hipMalloc(&mx, 4096 * 4);
    for (int lds : {12800, 13120, 13312, 14080, 14081}) {
        (void)hipMemset(now, 0, 4096 * 4); (void)hipMemset(mx, 0, 4096 * 4);
        resident<<<256 * 16, 64, lds>>>(now, mx, 20000);   // 200 us at 100 MHz
        (void)hipDeviceSynchronize();
        int h[4096]; (void)hipMemcpy(h, mx, sizeof(h), hipMemcpyDeviceToHost);
        int best = 0, cus = 0; for (int i = 0; i < 4096; ++i) { if (h[i] > best) best = h[i]; cus += h[i] > 0; }
        printf("measured: dynamic LDS %6d B -> max %d co-resident workgroups on a CU (%d CUs seen)\n", lds, best, cus);
    }
    return 0;
}
