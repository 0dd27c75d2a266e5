// LDS bank-conflict probe for gfx950 (run on the GPU box): one wavefront, lane i reads word
// (i * stride_dwords + j) for b32 / b64 / b128 accesses; prints cycles per access instruction.
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_probe.hip -o tools/build/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W> __global__ void probe(int stride, int group, int gstride, long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x;
    // lanes form groups of `group`; inside a group lane j reads offset j*stride, groups are gstride dwords apart
    const int off = (((lane / group) * gstride + (lane % group) * stride) & (16384 - 4)) / W * W;
    const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>(lds)) + off * 4;
    unsigned acc = 0;
    long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (W == 1) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)"); }
            if constexpr (W == 2) { uint2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)"); }
            if constexpr (W == 4) { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)"); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    long long c1 = clock64();
    if (lane == 0) out[0] = c1 - c0;
    sink[lane] = acc;
}

int main() {
    long long* d_out; unsigned* d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 256);
    auto run = [&](int W, int stride, int group, int gstride) {
        long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (W == 1) probe<1><<<1, 64>>>(stride, group, gstride, d_out, d_sink);
            if (W == 2) probe<2><<<1, 64>>>(stride, group, gstride, d_out, d_sink);
            if (W == 4) probe<4><<<1, 64>>>(stride, group, gstride, d_out, d_sink);
            hipDeviceSynchronize();
            hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        }
        printf("W=%d dwords stride=%3d group=%2d gstride=%4d : %.2f clk/instr\n", W, stride, group, gstride, double(h) / (256.0 * 16.0));
    };
    const int strides[] = {0, 1, 2, 3, 4, 8, 12, 16, 24, 32, 48, 64, 128};
    for (int W : {1, 2, 4})
        for (int s : strides) run(W, s, 64, 0);
    // 4 groups of 16 lanes: broadcast inside a group, groups gstride apart (the filter-slice pattern)
    for (int W : {1, 2, 4})
        for (int gs : {0, 4, 8, 16, 32, 64, 96, 100, 104, 128, 1648, 1652, 3296, 3300}) run(W, 0, 16, gs);
    // per-lane rows inside a group of 16 (stride) with groups 1652 apart
    for (int W : {1, 2, 4})
        for (int s : {4, 8, 12, 24, 26, 48, 50}) run(W, s, 16, 1652);
    return 0;
}
