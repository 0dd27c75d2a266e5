// VALU issue-rate probe for gfx950 (run on the GPU box): one wavefront, independent accumulators.
// Prints cycles per instruction for v_fmac_f32, v_fmac_f64, their row_newbcast DPP forms and v_pk_fma_f32.
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/build/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE> __global__ void probe(long long* out, float* sink, int waves) {
    float a[8]; double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; d[i] = threadIdx.x + i; }
    float x = 1.0001f, y = 0.5f; double dx = 1.0001, dy = 0.5;
    long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 512; ++it) {
        if constexpr (MODE == 0) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 1) {
#define X(i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(dx), "v"(dy));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 2) {
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x), "v"(y));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 3) {
#define X(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[i]) : "v"(dx), "v"(dy));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 4) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(dx), "v"(dy));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 5) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(x));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 6) {
#define X(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(x));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 7) {
#define X(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(dx));
            REP8(X) REP8(X)
#undef X
        }
    }
    long long c1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + float(d[i]);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    long long* d_out; float* d_sink;
    hipMalloc(&d_out, 8 * 64); hipMalloc(&d_sink, 4 * 64 * 1024);
    const char* names[] = {"v_fmac_f32", "v_fmac_f64", "v_fmac_f32_dpp", "v_fmac_f64_dpp", "v_pk_fma_f32", "v_mov_b32_dpp", "v_rcp_f32", "v_rcp_f64"};
    for (int threads : {64, 256, 512}) {   // 1, 4, 8 waves in one workgroup (1 / 1 / 2 per SIMD)
        for (int m = 0; m < 8; ++m) {
            long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                switch (m) {
                    case 0: probe<0><<<1, threads>>>(d_out, d_sink, 0); break;
                    case 1: probe<1><<<1, threads>>>(d_out, d_sink, 0); break;
                    case 2: probe<2><<<1, threads>>>(d_out, d_sink, 0); break;
                    case 3: probe<3><<<1, threads>>>(d_out, d_sink, 0); break;
                    case 4: probe<4><<<1, threads>>>(d_out, d_sink, 0); break;
                    case 5: probe<5><<<1, threads>>>(d_out, d_sink, 0); break;
                    case 6: probe<6><<<1, threads>>>(d_out, d_sink, 0); break;
                    case 7: probe<7><<<1, threads>>>(d_out, d_sink, 0); break;
                }
                hipDeviceSynchronize();
                hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
            }
            printf("threads=%3d %-16s %.2f clk/instr (per wave)\n", threads, names[m], double(h) / (512.0 * 16.0));
        }
    }
    return 0;
}
