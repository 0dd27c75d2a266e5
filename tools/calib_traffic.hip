// calib_traffic.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access pattern of
// the UKF engine (MI355X_MICROARCH.md, HBM section: FETCH_SIZE halves wide coalesced reads, "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Pattern replayed: one 16-lane row per record, lane l touches record[l + 16 t] (t = 0..), records of
// REC scalars back to back -- exactly how ukf_kernel16 stages mean + packed covariance and writes them back.
// usage: calib_traffic <records> <f64|f32>   (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

template <class T, int REC> __global__ void __launch_bounds__(64) read_rows(const T* in, T* sink, long n) {
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const long f = long(blockIdx.x) * 4 + g;
    T acc = T(0);
    if (f < n)
        for (int e = l; e < REC; e += 16) acc += in[f * REC + e];
    if (acc == T(123456789)) sink[0] = acc;   // never true: keeps the loads alive
}
template <class T, int REC> __global__ void __launch_bounds__(64) write_rows(T* out, long n) {
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const long f = long(blockIdx.x) * 4 + g;
    if (f < n)
        for (int e = l; e < REC; e += 16) out[f * REC + e] = T(e);
}

template <class T> int run(long n) {
    constexpr int REC = 91;   // 13 mean + 78 packed covariance scalars per Pose filter
    T *buf, *sink;
    if (hipMalloc(&buf, size_t(n) * REC * sizeof(T)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(buf, 0, size_t(n) * REC * sizeof(T));
    const unsigned grid = unsigned((n + 3) / 4);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL((read_rows<T, REC>), dim3(grid), dim3(64), 0, 0, buf, sink, n);
        hipLaunchKernelGGL((write_rows<T, REC>), dim3(grid), dim3(64), 0, 0, buf, n);
    }
    hipDeviceSynchronize();
    std::printf("{\"records\": %ld, \"bytes_per_launch\": %zu, \"scalar_bytes\": %zu}\n", n, size_t(n) * REC * sizeof(T), sizeof(T));
    return 0;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? std::atol(argv[1]) : 1048576;
    const bool f32 = argc > 2 && std::strcmp(argv[2], "f32") == 0;
    return f32 ? run<float>(n) : run<double>(n);
}
