// cabi_bench.cpp -- the headline measurement from a C++ host, nothing but include/ukf_batch.h and the HIP runtime:
// N PoseWithVelocity filters resident in HBM, K fused predict + update cycles through ukfb_cycle_dev with the samples already on
// the device, timed with the engine's HIP events.  north_star's host language is C++; bench.py (Python, ctypes) is plumbing around
// the same entry points -- this program exists to show that the number does not depend on it.
//   usage: cabi_bench [filters=1048576] [cycles=500] [f64|f32] [shards=0]   (shards > 0: a ukfb_group_* of that many shards on device 0)
// Inputs are simple and deterministic (not bench.py's synthetic set): the figure is a rate, parity is the GPU tests' business.
#include <hip/hip_runtime.h>
#include <ukf_batch.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(call)                                                                   \
    do {                                                                              \
        if ((call) != UKFB_OK) {                                                      \
            std::fprintf(stderr, "%s failed: %s\n", #call, ukfb_last_error());        \
            return 1;                                                                 \
        }                                                                             \
    } while (0)
#define HIPCHECK(call)                                                                \
    do {                                                                              \
        const hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));    \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

template <class T> static int upload(void** dev, const std::vector<double>& host) {
    std::vector<T> tmp(host.begin(), host.end());
    HIPCHECK(hipMalloc(dev, tmp.size() * sizeof(T)));
    HIPCHECK(hipMemcpy(*dev, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 1048576;
    const int cycles = argc > 2 ? std::atoi(argv[2]) : 500;
    const int prec = (argc > 3 && !std::strcmp(argv[3], "f32")) ? UKFB_F32 : UKFB_F64;
    const int shards = argc > 4 ? std::atoi(argv[4]) : 0;
    const int warm = 50;

    // initial states: positions on a grid, small rotations about z, 1 m/s forward; a diagonal covariance; an IMU acceleration
    std::vector<double> mu(size_t(n) * 13), acc(size_t(n) * 3), z(size_t(n) * 3), Q(size_t(n) * 9, 0.0);
    std::vector<double> cov1(144, 0.0);
    for (int k = 0; k < 12; ++k) cov1[k * 13] = (k < 3) ? 0.04 : (k < 6 ? 0.01 : 0.0025);
    for (int64_t i = 0; i < n; ++i) {
        double* m = &mu[size_t(i) * 13];
        const double yaw = 0.001 * double(i % 1000);
        m[0] = 0.01 * double(i % 977); m[1] = 0.02 * double(i % 389); m[2] = 1.0;
        m[3] = 0.0; m[4] = 0.0; m[5] = std::sin(0.5 * yaw); m[6] = std::cos(0.5 * yaw);
        m[7] = 1.0; m[8] = 0.0; m[9] = 0.0; m[10] = 0.0; m[11] = 0.0; m[12] = 0.05;
        acc[size_t(i) * 3 + 0] = 0.1; acc[size_t(i) * 3 + 1] = -0.05; acc[size_t(i) * 3 + 2] = 0.0;
        for (int k = 0; k < 3; ++k) z[size_t(i) * 3 + k] = m[k] + 0.01 * double((i + k) % 7 - 3);
        for (int k = 0; k < 3; ++k) Q[size_t(i) * 9 + k * 4] = 0.0025;
    }
    const double acc_cov[9] = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01};
    double R[144] = {0};   // PoseUKF.cpp:103-107
    for (int k = 0; k < 3; ++k) { R[k * 13] = 0.01; R[(3 + k) * 13] = 0.001; R[(6 + k) * 13] = 0.00001; R[(9 + k) * 13] = 0.00001; }

    void *acc_d = nullptr, *z_d = nullptr, *Q_d = nullptr;
    if (prec == UKFB_F64 ? (upload<double>(&acc_d, acc) || upload<double>(&z_d, z) || upload<double>(&Q_d, Q))
                         : (upload<float>(&acc_d, acc) || upload<float>(&z_d, z) || upload<float>(&Q_d, Q)))
        return 1;
    const size_t ts = prec == UKFB_F64 ? 8 : 4;

    float ms = 0.f;
    uint32_t status = 0;
    if (shards <= 0) {
        ukfb_engine* e = nullptr;
        CHECK(ukfb_create(&e, UKFB_MODEL_POSE, prec, n, 0, nullptr));
        CHECK(ukfb_set_process_noise(e, R));
        const int64_t CH = 65536;   // (the 144-scalar covariance rows: one chunk of host memory at a time)
        std::vector<double> cov(size_t(CH) * 144);
        for (int64_t i = 0; i < CH; ++i) std::memcpy(&cov[size_t(i) * 144], cov1.data(), 144 * sizeof(double));
        for (int64_t lo = 0; lo < n; lo += CH) {
            const int64_t c = (n - lo < CH) ? (n - lo) : CH;
            CHECK(ukfb_initialize(e, lo, c, &mu[size_t(lo) * 13], cov.data()));
        }
        CHECK(ukfb_pose_set_acceleration(e, 0, 0, nullptr, acc_cov));
        CHECK(ukfb_pose_bind_acceleration_dev(e, acc_d));
        for (int k = 0; k < warm; ++k) CHECK(ukfb_cycle_dev(e, 0.01, UKFB_MEAS_POS3, nullptr, z_d, Q_d));
        CHECK(ukfb_sync(e));
        CHECK(ukfb_timer_begin(e));
        for (int k = 0; k < cycles; ++k) CHECK(ukfb_cycle_dev(e, 0.01, UKFB_MEAS_POS3, nullptr, z_d, Q_d));
        CHECK(ukfb_timer_end(e, &ms));
        CHECK(ukfb_get_status_summary(e, &status));
        char name[128] = {0};
        ukfb_last_launch_info(e, name, sizeof(name), nullptr, nullptr, nullptr);
        std::printf("{\"host\": \"C++ over include/ukf_batch.h\", \"filters\": %lld, \"cycles\": %d, \"precision\": \"%s\", \"kernel\": \"%s\", ",
                    (long long)n, cycles, prec == UKFB_F64 ? "f64" : "f32", name);
        CHECK(ukfb_destroy(e));
    } else {
        std::vector<int> devs(size_t(shards), 0);
        ukfb_group* g = nullptr;
        CHECK(ukfb_group_create(&g, UKFB_MODEL_POSE, prec, n, devs.data(), shards));
        CHECK(ukfb_group_set_process_noise(g, R));
        const int64_t CH = 65536;
        std::vector<double> cov(size_t(CH) * 144);
        for (int64_t i = 0; i < CH; ++i) std::memcpy(&cov[size_t(i) * 144], cov1.data(), 144 * sizeof(double));
        for (int64_t lo = 0; lo < n; lo += CH) {
            const int64_t c = (n - lo < CH) ? (n - lo) : CH;
            CHECK(ukfb_group_initialize(g, lo, c, &mu[size_t(lo) * 13], cov.data()));
        }
        CHECK(ukfb_group_pose_set_acceleration(g, 0, 0, nullptr, acc_cov));
        std::vector<const void*> a_p(size_t(shards), nullptr), z_p(size_t(shards), nullptr), Q_p(size_t(shards), nullptr);
        for (int r = 0; r < shards; ++r) {
            int64_t first = 0, count = 0;
            CHECK(ukfb_group_shard(g, r, nullptr, nullptr, &first, &count));
            a_p[size_t(r)] = static_cast<const char*>(acc_d) + size_t(first) * 3 * ts;
            z_p[size_t(r)] = static_cast<const char*>(z_d) + size_t(first) * 3 * ts;
            Q_p[size_t(r)] = static_cast<const char*>(Q_d) + size_t(first) * 9 * ts;
        }
        CHECK(ukfb_group_pose_bind_acceleration_dev(g, a_p.data()));
        for (int k = 0; k < warm; ++k) CHECK(ukfb_group_cycle_dev(g, 0.01, UKFB_MEAS_POS3, z_p.data(), Q_p.data()));
        CHECK(ukfb_group_sync(g));
        CHECK(ukfb_group_timer_begin(g));
        for (int k = 0; k < cycles; ++k) CHECK(ukfb_group_cycle_dev(g, 0.01, UKFB_MEAS_POS3, z_p.data(), Q_p.data()));
        CHECK(ukfb_group_timer_end(g, &ms, nullptr));
        CHECK(ukfb_group_get_status_summary(g, &status));
        std::printf("{\"host\": \"C++ over include/ukf_batch.h, ukfb_group_* with %d shards on device 0\", \"filters\": %lld, \"cycles\": %d, \"precision\": \"%s\", ",
                    shards, (long long)n, cycles, prec == UKFB_F64 ? "f64" : "f32");
        CHECK(ukfb_group_destroy(g));
    }
    std::printf("\"ms_per_cycle\": %.5f, \"filter_cycles_per_s\": %.4e, \"status_or\": %u}\n", double(ms) / cycles,
                double(n) * cycles / (double(ms) * 1e-3), status);
    (void)hipFree(acc_d); (void)hipFree(z_d); (void)hipFree(Q_d);
    return status == 0 ? 0 : 2;
}
