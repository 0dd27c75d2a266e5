// VALU THROUGHPUT probe for gfx950 (run on the GPU box): every SIMD holds `W` wavefronts that issue independent
// instructions of one kind; wall time (HIP events) over the whole grid gives instructions per SIMD per second, printed
// relative to v_fmac_f32 and as cycles per wave-instruction at the clock GRBM-free estimate (time of a v_fmac_f64 = 4 cycles
// by the architecture's fp64 peak).  tools/valu_probe.hip measures the per-wave issue interval instead.
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_tput.hip -o tools/build/valu_tput
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int ITERS = 4096;

template <int MODE> __global__ __launch_bounds__(64) void probe(float* sink) {
    float a[8]; double d[8];
    __shared__ float lds[256];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 64] = 1.f; lds[threadIdx.x + 128] = 2.f; lds[threadIdx.x + 192] = 3.f;
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; d[i] = threadIdx.x + i; }
    float x = 1.0001f, y = 0.5f; double dx = 1.0001, dy = 0.5;
    unsigned addr = (threadIdx.x & 48) * 4;   // one address per 16-lane row: broadcast read
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
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
        } else if constexpr (MODE == 8) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 9) {
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(x), "v"(y));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 10) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(x));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 11) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d[i]) : "v"(dx), "v"(dy));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 12) {
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x), "v"(y));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 13) {
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x), "v"(y));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 14) {
#define X(i) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(addr));
            REP8(X) REP8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (MODE == 15) {
#define X(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(addr));
            REP8(X) REP8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (MODE == 16) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 q[4];
#define X(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(addr));
            REP8(X) REP8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
            a[0] += q[0].x + q[1].x + q[2].x + q[3].x;
        } else if constexpr (MODE == 17) {
#define X(i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(dx), "v"(dy));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 18) {
#define X(i) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(dx), "v"(dy));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (MODE == 19) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(dx));
            REP8(X) REP8(X)
#undef X
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + float(d[i]);
    if (s == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
}

template <int M> float run(float* sink, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<M><<<blocks, 64>>>(sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<M><<<blocks, 64>>>(sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main() {
    float* d_sink; hipMalloc(&d_sink, 4 * 64 * 16384);
    const char* names[] = {"v_fmac_f32", "v_fmac_f64", "v_fmac_f32_dpp newbcast", "v_fmac_f64_dpp newbcast", "v_pk_fma_f32",
                           "v_mov_b32_dpp newbcast", "v_rcp_f32", "v_rcp_f64", "v_fma_f32", "v_cndmask_b32", "v_mov_b32", "v_pk_mul_f32",
                           "v_fmac_f32_dpp quad_perm", "v_fmac_f32_dpp row_shr", "ds_read_b32 bcast", "ds_read_b64 bcast",
                           "ds_read_b128 bcast", "v_mul_f64", "v_add_f64", "v_mov_b64"};
    for (int W : {1, 2, 3, 4, 8}) {
        const int blocks = 1024 * W;          // 1024 SIMDs x W one-wave workgroups
        float ms[20];
        ms[0] = run<0>(d_sink, blocks); ms[1] = run<1>(d_sink, blocks); ms[2] = run<2>(d_sink, blocks); ms[3] = run<3>(d_sink, blocks);
        ms[4] = run<4>(d_sink, blocks); ms[5] = run<5>(d_sink, blocks); ms[6] = run<6>(d_sink, blocks); ms[7] = run<7>(d_sink, blocks);
        ms[8] = run<8>(d_sink, blocks); ms[9] = run<9>(d_sink, blocks); ms[10] = run<10>(d_sink, blocks); ms[11] = run<11>(d_sink, blocks);
        ms[12] = run<12>(d_sink, blocks); ms[13] = run<13>(d_sink, blocks); ms[14] = run<14>(d_sink, blocks); ms[15] = run<15>(d_sink, blocks);
        ms[16] = run<16>(d_sink, blocks); ms[17] = run<17>(d_sink, blocks); ms[18] = run<18>(d_sink, blocks); ms[19] = run<19>(d_sink, blocks);
        const double per = double(ITERS) * 16.0 * W;   // wave-instructions per SIMD
        for (int m = 0; m < 20; ++m)
            printf("W=%d %-26s %8.3f ms  %6.3f ns per wave-instruction per SIMD  (x%.2f of v_fmac_f64)\n", W, names[m], ms[m],
                   ms[m] * 1e6 / per, ms[m] / ms[1]);
    }
    return 0;
}
