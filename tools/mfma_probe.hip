// Probe (runs ON THE GPU BOX): operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, to decide whether two
// of them can replace the 12-instruction fp64 DPP butterfly for a sum over the 16 lanes of a DPP row.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_probe tools/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    const double one = 1.0, id = double(lane % 16) + 100.0 * double(lane / 16);
    double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(id, one, 0.0, 0, 0, 0);    // A = lane id, B = 1
    double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(one, id, 0.0, 0, 0, 0);    // A = 1, B = lane id
    // candidate all-reduce: x = lane id; partial sums, then a second product
    double p = __builtin_amdgcn_mfma_f64_4x4x4f64(id, one, 0.0, 0, 0, 0);
    double t1 = __builtin_amdgcn_mfma_f64_4x4x4f64(one, p, 0.0, 0, 0, 0);     // partials as B
    double t2 = __builtin_amdgcn_mfma_f64_4x4x4f64(p, one, 0.0, 0, 0, 0);     // partials as A
    out[lane] = d1; out[64 + lane] = d2; out[128 + lane] = t1; out[192 + lane] = t2;
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"A=id,B=1", "A=1,B=id", "B=partials", "A=partials"};
    for (int k = 0; k < 4; ++k) {
        printf("%s (row sums of ids 0..15 = 120 in row 0, +1600 per row)\n", names[k]);
        for (int l = 0; l < 64; ++l) printf("%7.0f%s", h[k * 64 + l], (l % 16 == 15) ? "\n" : " ");
    }
    return 0;
}
