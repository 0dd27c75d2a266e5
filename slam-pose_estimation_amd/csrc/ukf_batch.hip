// ukf_batch.hip -- implementation of the C-ABI declared in include/ukf_batch.h.
//
// Host-side plumbing only: device allocation, AoS(double, full covariance) <-> device (engine
// precision, packed lower triangle) conversion, latched inputs, launch requests.  All arithmetic
// of the hot path is in ukf_kernel.hpp.  There is deliberately NO CPU fallback: without a HIP
// device ukfb_create fails with UKFB_ERR_NO_DEVICE.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <thread>
#include <cstdlib>

#include <hipcub/hipcub.hpp>

#include "ukf_engine.hpp"

namespace {
// The covariance tiles of the OrientationState kernels are 3 x 3 on a 13 x 13 matrix: a lane whose tile hangs over the edge (rows 13,
// 14) loads its noise entries with immediate offsets from the tile origin like every other lane and drops them -- up to 2 * 13 + 2
// scalars past the last table.  Every noise allocation carries that many spare scalars (never written, never used).
constexpr size_t NOISE_TABLE_PAD = 32;

thread_local std::string g_last_error;
thread_local bool g_wait_timed_out = false;   // the last bounded wait of this thread gave up (it wrote the error text itself)

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            if (!(_e == hipErrorNotReady && g_wait_timed_out)) ukfb::set_error(#expr, _e); \
            g_wait_timed_out = false;                      \
            return UKFB_ERR_HIP;                           \
        }                                                  \
    } while (0)

// the engine's device for the rest of the enclosing scope, the caller's device again afterwards (ukfb::DeviceScope)
#define ON_DEVICE(dev)                                     \
    ukfb::DeviceScope device_scope_(dev);                  \
    HIP_TRY(device_scope_.err)

// Waits for the engine's stream with the bounded polling wait (defined below).  A wait that gives up POISONS the engine:
// work of unknown state is still queued, so every later call fails fast with UKFB_ERR_HIP, and ukfb_destroy neither waits
// again nor frees device memory that a kernel in flight may still touch (the process is expected to exit non-zero).
int engine_wait(ukfb_engine* e);
#define ENGINE_SYNC(e)                                     \
    do {                                                   \
        const int _rc = engine_wait(e);                    \
        if (_rc) return _rc;                               \
    } while (0)

bool range_ok(const ukfb_engine* e, int64_t first, int64_t count) {
    return e && first >= 0 && count >= 0 && first + count <= e->cap;
}

int fail(int code, const char* msg) {
    g_last_error = msg;
    return code;
}

// pack / convert host doubles into engine precision -------------------------------------------
template <class T> void convert(const double* src, T* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = T(src[i]);
}

int ensure_scratch(ukfb_engine* e, size_t bytes);

__global__ void narrow_kernel(const double* src, float* dst, size_t n) {
    const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i < n) dst[i] = float(src[i]);
}

int upload(ukfb_engine* e, void* dst_dev, size_t elem_offset, const double* src, size_t n) {
    if (n == 0) return UKFB_OK;
    if (e->prec == UKFB_F64) {
        HIP_TRY(hipMemcpyAsync(static_cast<double*>(dst_dev) + elem_offset, src, n * sizeof(double),
                               hipMemcpyHostToDevice, ukfb::main_stream(e)));
        ENGINE_SYNC(e);
    } else if (n < 16384) {
        std::vector<float> tmp(n);
        convert(src, tmp.data(), n);
        HIP_TRY(hipMemcpyAsync(static_cast<float*>(dst_dev) + elem_offset, tmp.data(), n * sizeof(float),
                               hipMemcpyHostToDevice, ukfb::main_stream(e)));
        ENGINE_SYNC(e);
    } else {
        // fp32 engine, large array: the doubles cross PCIe as they are and are narrowed on the device (a
        // single-threaded host loop over 12 M values per cycle took ten times longer than the copy)
        {
            const int rc = ensure_scratch(e, n * sizeof(double));
            if (rc) return rc;
        }
        HIP_TRY(hipMemcpyAsync(e->cvt_dev, src, n * sizeof(double), hipMemcpyHostToDevice, ukfb::main_stream(e)));
        hipLaunchKernelGGL(narrow_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ukfb::main_stream(e),
                           static_cast<const double*>(e->cvt_dev), static_cast<float*>(dst_dev) + elem_offset, n);
        HIP_TRY(hipGetLastError());
        ENGINE_SYNC(e);
    }
    return UKFB_OK;
}

// grow-only device scratch (see ukfb_engine::cvt_dev)
int ensure_scratch(ukfb_engine* e, size_t bytes) {
    if (e->cvt_bytes >= bytes) return UKFB_OK;
    ENGINE_SYNC(e);
    if (e->cvt_dev) HIP_TRY(hipFree(e->cvt_dev));
    e->cvt_dev = nullptr;
    e->cvt_bytes = 0;
    HIP_TRY(hipMalloc(&e->cvt_dev, bytes));
    e->cvt_bytes = bytes;
    return UKFB_OK;
}

// full D x D host covariances (double, row-major) <-> the engine's packed lower triangle (precision T), on the device:
// one thread per scalar.  Replaces single-threaded host loops over count * D * D values in initialize / get_state.
template <class T> __global__ void pack_cov_kernel(const double* full, T* packed, int64_t count, int D, int PK) {
    const int64_t gid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (gid >= count * PK) return;
    const int64_t i = gid / PK;
    const int e = int(gid % PK);
    int r = int((sqrtf(8.0f * float(e) + 1.0f) - 1.0f) * 0.5f);
    if (r * (r + 1) / 2 > e) --r;
    if ((r + 1) * (r + 2) / 2 <= e) ++r;
    const int c = e - r * (r + 1) / 2;
    packed[gid] = T(full[i * D * D + r * D + c]);          // the lower triangle is what Eigen's LLT reads
}
template <class T> __global__ void unpack_cov_kernel(const T* packed, double* full, int64_t count, int D, int PK) {
    const int64_t gid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (gid >= count * D * D) return;
    const int64_t i = gid / (int64_t(D) * D);
    const int rc = int(gid % (int64_t(D) * D)), r = rc / D, c = rc % D;
    const int hi = r > c ? r : c, lo = r > c ? c : r;
    full[gid] = double(packed[i * PK + hi * (hi + 1) / 2 + lo]);
}
__global__ void widen_kernel(const float* src, double* dst, size_t n) {
    const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i < n) dst[i] = double(src[i]);
}
constexpr int64_t COV_CHUNK = 65536;   // filters per pack / unpack pass (75 MB of scratch for D = 12)

int download(ukfb_engine* e, const void* src_dev, size_t elem_offset, double* dst, size_t n) {
    if (n == 0) return UKFB_OK;
    if (e->prec == UKFB_F64) {
        HIP_TRY(hipMemcpyAsync(dst, static_cast<const double*>(src_dev) + elem_offset, n * sizeof(double),
                               hipMemcpyDeviceToHost, ukfb::main_stream(e)));
        ENGINE_SYNC(e);
    } else if (n < 16384) {
        std::vector<float> tmp(n);
        HIP_TRY(hipMemcpyAsync(tmp.data(), static_cast<const float*>(src_dev) + elem_offset, n * sizeof(float),
                               hipMemcpyDeviceToHost, ukfb::main_stream(e)));
        ENGINE_SYNC(e);
        for (size_t i = 0; i < n; ++i) dst[i] = double(tmp[i]);
    } else {   // fp32 engine, large array: widened on the device, the doubles cross PCIe as they are
        int rc = ensure_scratch(e, n * sizeof(double));
        if (rc) return rc;
        hipLaunchKernelGGL(widen_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ukfb::main_stream(e),
                           static_cast<const float*>(src_dev) + elem_offset, static_cast<double*>(e->cvt_dev), n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(dst, e->cvt_dev, n * sizeof(double), hipMemcpyDeviceToHost, ukfb::main_stream(e)));
        ENGINE_SYNC(e);
    }
    return UKFB_OK;
}

template <class P> int upload_raw(ukfb_engine* e, P* dst_dev, const P* src, size_t n) {
    if (n == 0) return UKFB_OK;
    HIP_TRY(hipMemcpyAsync(dst_dev, src, n * sizeof(P), hipMemcpyHostToDevice, ukfb::main_stream(e)));
    ENGINE_SYNC(e);
    return UKFB_OK;
}

template <class P> int download_raw(ukfb_engine* e, const P* src_dev, P* dst, size_t n) {
    if (n == 0) return UKFB_OK;
    HIP_TRY(hipMemcpyAsync(dst, src_dev, n * sizeof(P), hipMemcpyDeviceToHost, ukfb::main_stream(e)));
    ENGINE_SYNC(e);
    return UKFB_OK;
}

int fill_scalar(ukfb_engine* e, void* dst_dev, size_t n, double value) {
    std::vector<double> v(n, value);
    return upload(e, dst_dev, 0, v.data(), n);
}

int launch(ukfb_engine* e, const ukfb::LaunchReq& r) {
    if (e->poisoned) return fail(UKFB_ERR_HIP, "engine poisoned by an earlier wait that timed out (UKFB_WAIT_TIMEOUT_S)");
    ON_DEVICE(e->device);
    if (r.wait_event) HIP_TRY(hipStreamWaitEvent(ukfb::main_stream(e), r.wait_event, 0));
    int rc;
    const bool wide = e->prec == UKFB_F32 && e->cfg.wide_arithmetic != 0;   // fp32 arrays, fp64 arithmetic (tuned layout only)
    if (e->model == UKFB_MODEL_POSE)
        rc = e->prec == UKFB_F64 ? ukfb::launch_pose_f64(e, r) : (wide ? ukfb::launch_pose_f32w(e, r) : ukfb::launch_pose_f32(e, r));
    else
        rc = e->prec == UKFB_F64 ? ukfb::launch_orient_f64(e, r) : (wide ? ukfb::launch_orient_f32w(e, r) : ukfb::launch_orient_f32(e, r));
    if (!rc && r.done_event) HIP_TRY(hipEventRecord(r.done_event, ukfb::main_stream(e)));
    return rc;
}

// ---- host-fed fused cycles: double-buffered staging on a copy stream ----------------------------------------------------
// ukfb_cycle / ukfb_cycle_uniform_q upload z and Q before they launch.  On the engine's own stream that upload queues
// behind the kernel of the previous call: copy and kernel alternate (1 M Pose filters fp64: 2.0 ms of PCIe + 1.1 ms of
// kernel per cycle).  Here the upload of call k + 1 runs on a separate stream into the staging set call k is not reading;
// the call returns when ITS copies have been consumed (the caller's buffers are free), not when the kernel is done.
struct Staged {
    void* z = nullptr;
    void* Q = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    int slot = 0;
};

int ensure_copy_path(ukfb_engine* e) {
    if (e->copy_stream) return UKFB_OK;
    const size_t n = size_t(e->cap), ts = e->tsize;
    // everything into locals; the engine sees the set only when all of it exists (a failure frees what was created and a
    // retry starts from nothing -- no leaked stream / events / buffers, no half-initialised handles)
    hipStream_t cs = nullptr;
    hipEvent_t evc[2] = {nullptr, nullptr}, evu[2] = {nullptr, nullptr};
    void* zs[2] = {nullptr, nullptr};
    void* qs[2] = {nullptr, nullptr};
    hipError_t err = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    for (int s = 0; s < 2 && err == hipSuccess; ++s) {
        err = hipEventCreateWithFlags(&evc[s], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&evu[s], hipEventDisableTiming);
        if (err == hipSuccess) err = hipMalloc(&zs[s], n * 3 * ts);
        if (err == hipSuccess) err = hipMalloc(&qs[s], n * 9 * ts);
    }
    if (err != hipSuccess) {
        for (int s = 0; s < 2; ++s) {
            if (zs[s]) (void)hipFree(zs[s]);
            if (qs[s]) (void)hipFree(qs[s]);
            if (evc[s]) (void)hipEventDestroy(evc[s]);
            if (evu[s]) (void)hipEventDestroy(evu[s]);
        }
        if (cs) (void)hipStreamDestroy(cs);
        ukfb::set_error("host-fed cycles: copy stream / staging", err);
        return UKFB_ERR_HIP;
    }
    for (int s = 0; s < 2; ++s) {
        e->ev_copy[s] = evc[s];
        e->ev_used[s] = evu[s];
        e->zc_stage[s] = zs[s];
        e->Qc_stage[s] = qs[s];
    }
    e->copy_stream = cs;
    return UKFB_OK;
}

// host doubles -> device array in engine precision, enqueued on the copy stream (no wait)
int upload_on_copy_stream(ukfb_engine* e, void* dst_dev, const double* src, size_t n) {
    if (n == 0) return UKFB_OK;
    if (e->prec == UKFB_F64) {
        HIP_TRY(hipMemcpyAsync(dst_dev, src, n * sizeof(double), hipMemcpyHostToDevice, e->copy_stream));
        return UKFB_OK;
    }
    // fp32: the doubles cross PCIe as they are and are narrowed on the device (cf. upload)
    HIP_TRY(hipMemcpyAsync(e->cvt_copy, src, n * sizeof(double), hipMemcpyHostToDevice, e->copy_stream));
    hipLaunchKernelGGL(narrow_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, e->copy_stream,
                       static_cast<const double*>(e->cvt_copy), static_cast<float*>(dst_dev), n);
    HIP_TRY(hipGetLastError());
    return UKFB_OK;
}

hipError_t wait_stream_polling(hipStream_t s);

int stage_cycle_inputs(ukfb_engine* e, const double* z, const double* Q, size_t nq, Staged* out) {
    if (!z || !Q) return fail(UKFB_ERR_INVALID_ARG, "z and Q must not be NULL");
    if (e->poisoned) return fail(UKFB_ERR_HIP, "engine poisoned by an earlier wait that timed out (UKFB_WAIT_TIMEOUT_S)");
    int rc = ensure_copy_path(e);
    if (rc) return rc;
    const size_t nz = size_t(e->cap) * 3;
    if (e->prec == UKFB_F32) {   // scratch for the widest array of the call; z and Q pass through it one after the other
        const size_t need = std::max(nz, nq) * sizeof(double);
        if (e->cvt_copy_bytes < need) {
            HIP_TRY(hipStreamSynchronize(e->copy_stream));
            if (e->cvt_copy) HIP_TRY(hipFree(e->cvt_copy));
            e->cvt_copy = nullptr;
            e->cvt_copy_bytes = 0;
            HIP_TRY(hipMalloc(&e->cvt_copy, need));
            e->cvt_copy_bytes = need;
        }
    }
    const int s = e->stage_slot;
    e->stage_slot ^= 1;
    if (e->stage_busy[s]) HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->ev_used[s], 0));   // the kernel that read this set is done
    rc = upload_on_copy_stream(e, e->zc_stage[s], z, nz);
    if (!rc) rc = upload_on_copy_stream(e, e->Qc_stage[s], Q, nq);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev_copy[s], e->copy_stream));
    // the caller's buffers are free again once the copies have been consumed: wait for the COPY stream only (bounded)
    {
        const hipError_t w = wait_stream_polling(e->copy_stream);
        if (w == hipErrorNotReady && g_wait_timed_out) {
            e->poisoned = true;
            g_wait_timed_out = false;
            return UKFB_ERR_HIP;
        }
        HIP_TRY(w);
    }
    out->z = e->zc_stage[s];
    out->Q = e->Qc_stage[s];
    out->ready = e->ev_copy[s];
    out->done = e->ev_used[s];
    out->slot = s;
    return UKFB_OK;
}

bool meas_model_ok(const ukfb_engine* e, int m) {
    if (e->model == UKFB_MODEL_POSE) return m >= UKFB_MEAS_POS3 && m <= UKFB_MEAS_ANGVEL3;
    return m == UKFB_MEAS_ORIENT_BODYVEL3;
}

// stage host measurement arrays into the engine's device buffers
int stage_measurements(ukfb_engine* e, const double* z, const double* Q, const int32_t* meas, const uint8_t* active) {
    if (!z || !Q) return fail(UKFB_ERR_INVALID_ARG, "z and Q must not be NULL");
    int rc = upload(e, e->z_stage, 0, z, size_t(e->cap) * 3);
    if (rc) return rc;
    rc = upload(e, e->Q_stage, 0, Q, size_t(e->cap) * 9);
    if (rc) return rc;
    if (meas) {
        rc = upload_raw(e, e->meas_stage, meas, size_t(e->cap));
        if (rc) return rc;
    }
    if (active) {
        rc = upload_raw(e, e->active_stage, active, size_t(e->cap));
        if (rc) return rc;
    }
    return UKFB_OK;
}

// Racc = Rn with block(6,6,3,3) = 2 acc.cov for every stored noise matrix (PoseUKF.cpp:190-191)
template <class T> __global__ void build_racc_kernel(const T* Rn, T* Racc, int64_t nmat, int D, const T* acc_cov9) {
    const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i >= nmat * D * D) return;
    const int rc = int(i % (int64_t(D) * D)), r = rc / D, c = rc % D;
    const bool vel = r >= 6 && r < 9 && c >= 6 && c < 9;
    Racc[i] = vel ? T(2) * acc_cov9[(r - 6) * 3 + (c - 6)] : Rn[i];
}

// Positive semidefinite? -- of the symmetric matrix the kernels use (they read the lower triangle).  Cholesky with a tolerance: a pivot
// below -tol fails; a pivot within tol must head a column of zeros (a zero direction, e.g. the all-zero default noise).
bool is_psd(const double* A, int D) {
    std::vector<double> L(size_t(D) * D, 0.0);
    double dmax = 0.0;
    for (int i = 0; i < D; ++i) dmax = std::max(dmax, std::fabs(A[size_t(i) * D + i]));
    const double tol = 1e-12 * dmax;
    for (int k = 0; k < D; ++k) {
        double x = A[size_t(k) * D + k];
        if (!std::isfinite(x)) return false;
        for (int j = 0; j < k; ++j) x -= L[size_t(k) * D + j] * L[size_t(k) * D + j];
        if (x < -tol) return false;
        const bool zero = x <= tol;
        const double d = zero ? 0.0 : std::sqrt(x);
        L[size_t(k) * D + k] = d;
        for (int i = k + 1; i < D; ++i) {
            double v = A[size_t(i) * D + k];
            if (!std::isfinite(v)) return false;
            for (int j = 0; j < k; ++j) v -= L[size_t(i) * D + j] * L[size_t(k) * D + j];
            if (zero) {
                if (std::fabs(v) > std::sqrt(tol * std::max(dmax, 1e-300)) + 1e-300) return false;
                L[size_t(i) * D + k] = 0.0;
            } else {
                L[size_t(i) * D + k] = v / d;
            }
        }
    }
    return true;
}

// ukfb_config::full_update_check: the host-side condition of the short update factorisation -- the batch-uniform process noise as the
// prediction adds it (rotation of its diagonal blocks and scaling by dt / dt^2 keep semidefiniteness; Pose acceleration branch: the
// velocity block replaced by 2 acc.cov, PoseUKF.cpp:190-191) is positive semidefinite
void refresh_noise_psd(ukfb_engine* e) {
    if (e->Rn_per_filter || e->Rn_host.size() != size_t(e->D) * e->D) {
        e->noise_psd = false;
        return;
    }
    bool ok = is_psd(e->Rn_host.data(), e->D);
    if (ok && e->model == UKFB_MODEL_POSE) {
        std::vector<double> Ra(e->Rn_host);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Ra[size_t(6 + r) * e->D + (6 + c)] = 2.0 * e->acc_cov[r * 3 + c];
        ok = is_psd(Ra.data(), e->D);
    }
    e->noise_psd = ok;
}

int rebuild_racc(ukfb_engine* e) {
    refresh_noise_psd(e);
    if (e->model != UKFB_MODEL_POSE) return UKFB_OK;
    const int64_t nmat = e->Rn_per_filter ? e->cap : 1;
    const size_t dd = size_t(e->D) * e->D;
    // both buffers persist for the life of the engine (Racc grows once, when the noise becomes per-filter)
    if (!e->Racc || e->Racc_mats < nmat) {
        if (e->Racc) {
            ENGINE_SYNC(e);
            HIP_TRY(hipFree(e->Racc));
        }
        e->Racc = nullptr;
        e->Racc_mats = 0;
        HIP_TRY(hipMalloc(&e->Racc, (size_t(nmat) * dd + NOISE_TABLE_PAD) * e->tsize));
        e->Racc_mats = nmat;
    }
    if (!e->acc_cov_dev) HIP_TRY(hipMalloc(&e->acc_cov_dev, 9 * sizeof(double)));
    int rc = upload(e, e->acc_cov_dev, 0, e->acc_cov, 9);
    if (rc) return rc;
    const int64_t total = nmat * int64_t(dd);
    const int blocks = int((total + 255) / 256);
    if (e->prec == UKFB_F64)
        hipLaunchKernelGGL(build_racc_kernel<double>, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), static_cast<const double*>(e->Rn),
                           static_cast<double*>(e->Racc), nmat, e->D, static_cast<const double*>(e->acc_cov_dev));
    else
        hipLaunchKernelGGL(build_racc_kernel<float>, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), static_cast<const float*>(e->Rn),
                           static_cast<float*>(e->Racc), nmat, e->D, static_cast<const float*>(e->acc_cov_dev));
    HIP_TRY(hipGetLastError());
    return UKFB_OK;   // stream-ordered: the next launch on the engine's stream sees the new table
}

// BodyStateMeasurement::toRigidBodyState for a batch: one thread per output scalar (coalesced stores).
template <class T> __global__ void export_body_states_kernel(const T* mu, const T* cov, int64_t first, int64_t count, T* out) {
    const int64_t gid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (gid >= count * 49) return;
    const int64_t i = gid / 49, f = first + i;
    const int j = int(gid % 49);
    const T* m = mu + f * 13;
    T v;
    if (j < 7) {
        v = m[j];                                   // position, orientation
    } else if (j < 10) {                            // velocity rotated into the navigation frame (:32)
        const T qx = m[3], qy = m[4], qz = m[5], qw = m[6];
        const T vx = m[7], vy = m[8], vz = m[9];
        T ux = qy * vz - qz * vy, uy = qz * vx - qx * vz, uz = qx * vy - qy * vx;
        ux += ux; uy += uy; uz += uz;
        const T r0 = vx + qw * ux + (qy * uz - qz * uy);
        const T r1 = vy + qw * uy + (qz * ux - qx * uz);
        const T r2 = vz + qw * uz + (qx * uy - qy * ux);
        v = (j == 7) ? r0 : ((j == 8) ? r1 : r2);
    } else if (j < 13) {
        v = m[j];                                   // angular velocity
    } else {
        const int b = (j - 13) / 9, rc = (j - 13) % 9, r = 3 * b + rc / 3, c = 3 * b + rc % 3;
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        v = cov[f * 78 + hi * (hi + 1) / 2 + lo];
    }
    out[gid] = v;
}
// fromRigidBodyState + initializeFilter: one thread per stored scalar (13 mean + 78 packed covariance)
template <class T> __global__ void import_body_states_kernel(const T* in, int64_t first, int64_t count, T* mu, T* cov) {
    const int64_t gid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (gid >= count * 91) return;
    const int64_t i = gid / 91, f = first + i;
    const int j = int(gid % 91);
    const T* rec = in + i * 49;
    if (j < 13) {
        mu[f * 13 + j] = rec[j];
    } else {
        const int e = j - 13;
        int r = int((sqrtf(8.0f * float(e) + 1.0f) - 1.0f) * 0.5f);
        if (r * (r + 1) / 2 > e) --r;
        if ((r + 1) * (r + 2) / 2 <= e) ++r;
        const int c = e - r * (r + 1) / 2;
        const bool same = (r / 3) == (c / 3);
        const T v = same ? rec[13 + (r / 3) * 9 + (r % 3) * 3 + (c % 3)] : T(0);
        cov[f * 78 + e] = v;
    }
}

namespace {
struct DevTmp {   // device scratch that is released on every exit path
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
};
}  // namespace

template <class T> int export_body_states(ukfb_engine* e, int64_t first, int64_t count, double* out) {
    DevTmp dev;
    HIP_TRY(hipMalloc(&dev.p, size_t(count) * 49 * sizeof(T)));
    const int blocks = int((count * 49 + 255) / 256);
    hipLaunchKernelGGL(export_body_states_kernel<T>, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), static_cast<const T*>(e->mu),
                       static_cast<const T*>(e->cov), first, count, static_cast<T*>(dev.p));
    HIP_TRY(hipGetLastError());
    return download(e, dev.p, 0, out, size_t(count) * 49);
}
template <class T> int import_body_states(ukfb_engine* e, int64_t first, int64_t count, const double* in) {
    DevTmp dev;
    HIP_TRY(hipMalloc(&dev.p, size_t(count) * 49 * sizeof(T)));
    int rc = upload(e, dev.p, 0, in, size_t(count) * 49);
    if (rc) return rc;
    const int blocks = int((count * 91 + 255) / 256);
    hipLaunchKernelGGL(import_body_states_kernel<T>, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), static_cast<const T*>(dev.p), first,
                       count, static_cast<T*>(e->mu), static_cast<T*>(e->cov));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(e->init + first, 1, size_t(count), ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->last_ts + first, 0, size_t(count) * sizeof(int64_t), ukfb::main_stream(e)));
    ENGINE_SYNC(e);
    return UKFB_OK;
}

// Class of a filter's update in a call: 0 = none (negative / invalid model id: prediction only), 1 = closed form (the eight
// linear sub-state selections of PoseUKF), 2 = sigma-point path (PoseUKF OrientationMeasurement, OrientationUKF body velocity)
__device__ __forceinline__ int update_class(int engine_model, int mid) {
    if (engine_model == UKFB_MODEL_POSE) return (mid < 0 || mid > 8) ? 0 : ((mid == 3) ? 2 : 1);
    return (mid == 9) ? 2 : 0;
}

// ---- device-side ordering of an event stream ------------------------------------------------------------
__global__ void events_init_kernel(const int64_t* filt, const int64_t* ts, int64_t n, int64_t cap, uint32_t* idx,
                                   int64_t* key_t, uint32_t* flags) {
    const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    idx[k] = uint32_t(k);
    key_t[k] = ts[k];
    if (filt[k] < 0 || filt[k] >= cap || ts[k] < 0) atomicOr(&flags[0], 1u);
}
__global__ void events_filter_keys_kernel(const int64_t* filt, const uint32_t* idx, int64_t n, int64_t cap, uint32_t* key_f) {
    const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const int64_t f = filt[idx[k]];
    key_f[k] = uint32_t((f < 0 || f >= cap) ? 0 : f);
}
__global__ void events_heads_kernel(const uint32_t* key_f_sorted, int64_t n, uint32_t* head) {
    const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    head[k] = (k == 0 || key_f_sorted[k] != key_f_sorted[k - 1]) ? uint32_t(k) : 0u;
}
// rank of a sample inside its filter's run = its round.  The sort key of the round-major order also carries the class of the
// sample's update (2 bits, most expensive first): inside a round the indirect launch then runs class-uniform wavefronts
// (all but the one or two that straddle a class boundary), as the model-class buckets of ukfb_cycle_dev do
__global__ void events_rank_kernel(const uint32_t* start, const uint32_t* idx, const int32_t* meas, int engine_model, int64_t n,
                                   uint32_t* key, uint32_t* flags) {
    const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (k >= n) return;
    const uint32_t r = uint32_t(k) - start[k];
    if (r >= (1u << 30)) atomicOr(&flags[0], 2u);   // a billion samples of one filter in one call: not representable in the key
    key[k] = (r << 2) | uint32_t(2 - update_class(engine_model, meas[idx[k]]));
    atomicMax(&flags[1], r);
}
// first position of every round in the (rank, filter)-ordered event list; off[rounds] = n
__global__ void events_round_offsets_kernel(const uint32_t* key_sorted, int64_t n, uint32_t* off) {
    const int64_t p = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (p >= n) return;
    const uint32_t r = key_sorted[p] >> 2;
    if (p == 0 || (key_sorted[p - 1] >> 2) != r) off[r] = uint32_t(p);
    if (p == n - 1) off[r + 1] = uint32_t(n);
}
// events in their final (round-major, filter-minor) order, compact and in the engine's precision: what the
// indirect launches of the fused kernel read directly (no per-round scatter, no capacity-sized staging)
template <class S, class T>
__global__ void events_gather_kernel(const int64_t* filt, const int64_t* ts, const int32_t* meas, const S* z, const S* Q,
                                     const uint32_t* order, int64_t n, int32_t* fidx_c, int64_t* ts_c, int32_t* meas_c, T* z_c,
                                     T* Q_c) {
    const int64_t p = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (p >= n) return;
    const int64_t i = order[p];
    fidx_c[p] = int32_t(filt[i]);
    ts_c[p] = ts[i];
    meas_c[p] = meas[i];
    for (int c = 0; c < 3; ++c) z_c[p * 3 + c] = T(z[i * 3 + c]);
    for (int c = 0; c < 9; ++c) Q_c[p * 9 + c] = T(Q[i * 9 + c]);
}


// ---- model-class buckets (ukfb_cycle_dev with per-filter model ids) ---------------------------------------
// A stable three-way partition (by update_class above) of the filter indices, two passes over the model ids: (1) per-block class
// counts, (2) ONE block turns them into exclusive prefix sums and class totals (3 x blocks values: 120 000 at the 40-million-
// filter capacity the tests exercise; every scatter block re-summing all of them was quadratic in the batch size), (3) every
// block scatters from its offsets.  The
// classes follow each other from the most expensive to the cheapest, each
// starts at a multiple of 4 (one wavefront = 4 filters); the gaps behind the classes are filled with -1 (padding) by the scatter kernel.
constexpr int BK_THREADS = 256, BK_PER_THREAD = 4, BK_BLOCK = BK_THREADS * BK_PER_THREAD;
__global__ void __launch_bounds__(BK_THREADS) bucket_count_kernel(const int32_t* meas, int64_t n, int engine_model, uint32_t* counts,
                                                                  int nblocks) {
    using Reduce = hipcub::BlockReduce<uint32_t, BK_THREADS>;
    __shared__ typename Reduce::TempStorage tmp;
    const int64_t base = int64_t(blockIdx.x) * BK_BLOCK + int64_t(threadIdx.x) * BK_PER_THREAD;
    uint32_t c1 = 0, c2 = 0, valid = 0;
#pragma unroll
    for (int j = 0; j < BK_PER_THREAD; ++j)
        if (base + j < n) {
            const int c = update_class(engine_model, meas[base + j]);
            c1 += c == 1;
            c2 += c == 2;
            ++valid;
        }
    // (counts of at most 1024 each: two of them share a word)
    const uint32_t packed = Reduce(tmp).Sum(c1 | (c2 << 16));
    __syncthreads();
    const uint32_t tot = Reduce(tmp).Sum(valid);
    if (threadIdx.x == 0) {
        const uint32_t n1 = packed & 0xFFFFu, n2 = packed >> 16;
        counts[blockIdx.x] = tot - n1 - n2;
        counts[nblocks + blockIdx.x] = n1;
        counts[2 * nblocks + blockIdx.x] = n2;
    }
}
// exclusive prefix sums of the per-block counts, class by class: before[c * nblocks + b] = sum of counts[c][0 .. b), total[c]
constexpr int BS_THREADS = 1024;
__global__ void __launch_bounds__(BS_THREADS) bucket_scan_kernel(const uint32_t* counts, int nblocks, uint32_t* before, uint32_t* total) {
    using Scan = hipcub::BlockScan<uint32_t, BS_THREADS>;
    __shared__ typename Scan::TempStorage tmp;
    __shared__ uint32_t carry;
    for (int c = 0; c < 3; ++c) {
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (int b0 = 0; b0 < nblocks; b0 += BS_THREADS) {
            const int b = b0 + int(threadIdx.x);
            const uint32_t v = b < nblocks ? counts[c * nblocks + b] : 0u;
            uint32_t ex, tile;
            Scan(tmp).ExclusiveSum(v, ex, tile);
            if (b < nblocks) before[c * nblocks + b] = carry + ex;
            __syncthreads();
            if (threadIdx.x == 0) carry += tile;
            __syncthreads();
        }
        if (threadIdx.x == 0) total[c] = carry;
        __syncthreads();
    }
}
// INLINE_SCAN (launches of up to BUCKET_INLINE_BLOCKS blocks): every block sums the per-block counts itself -- its own exclusive
// prefix and the class totals, 3 x nblocks values -- instead of reading them from bucket_scan_kernel's output: one launch fewer
// where the launches, not the sums, are what the grouping costs (four launches of ~5 us each per cycle at 262 144 filters were 6.6 %
// of config 5's cycle).  Block 0 also writes the padding (-1) behind each class and up to `items`: no memset of the list.
template <bool INLINE_SCAN>
__global__ void __launch_bounds__(BK_THREADS) bucket_scatter_kernel(const int32_t* meas, int64_t n, int engine_model, const uint32_t* counts,
                                                                    const uint32_t* before, const uint32_t* total, int nblocks,
                                                                    int32_t* order, uint32_t items) {
    using Scan = hipcub::BlockScan<uint32_t, BK_THREADS>;
    using Reduce64 = hipcub::BlockReduce<unsigned long long, BK_THREADS>;
    __shared__ typename Scan::TempStorage stmp;
    __shared__ typename Reduce64::TempStorage rtmp;
    __shared__ uint32_t start[3];
    if constexpr (INLINE_SCAN) {
        __shared__ uint32_t pre_s[3], tot_s[3];
        for (int c = 0; c < 3; ++c) {
            unsigned long long acc = 0;   // low word: counts of the blocks before this one, high word: of all blocks
            for (int b = int(threadIdx.x); b < nblocks; b += BK_THREADS) {
                const unsigned long long v = counts[c * nblocks + b];
                acc += (v << 32) | (b < int(blockIdx.x) ? v : 0ull);
            }
            const unsigned long long sum = Reduce64(rtmp).Sum(acc);
            if (threadIdx.x == 0) {
                pre_s[c] = uint32_t(sum);
                tot_s[c] = uint32_t(sum >> 32);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const uint32_t b1 = (tot_s[2] + 3u) & ~3u, b0 = b1 + ((tot_s[1] + 3u) & ~3u);
            start[2] = pre_s[2];
            start[1] = b1 + pre_s[1];
            start[0] = b0 + pre_s[0];
        }
        if (blockIdx.x == 0 && threadIdx.x < 16) {   // the gaps behind the three classes (at most 3 + 3 + 12 entries)
            const uint32_t b1 = (tot_s[2] + 3u) & ~3u, b0 = b1 + ((tot_s[1] + 3u) & ~3u);
            const uint32_t t = threadIdx.x;
            if (tot_s[2] + t < b1) order[tot_s[2] + t] = -1;
            if (b1 + tot_s[1] + t < b0) order[b1 + tot_s[1] + t] = -1;
            if (b0 + tot_s[0] + t < items) order[b0 + tot_s[0] + t] = -1;
        }
    } else {
        if (threadIdx.x == 0) {
            // the most expensive class first: the cheap wavefronts fill the tail of the launch
            const uint32_t b1 = (total[2] + 3u) & ~3u, b0 = b1 + ((total[1] + 3u) & ~3u);
            start[2] = before[2 * nblocks + blockIdx.x];
            start[1] = b1 + before[nblocks + blockIdx.x];
            start[0] = b0 + before[blockIdx.x];
        }
        if (blockIdx.x == 0 && threadIdx.x < 16) {
            const uint32_t b1 = (total[2] + 3u) & ~3u, b0 = b1 + ((total[1] + 3u) & ~3u);
            const uint32_t t = threadIdx.x;
            if (total[2] + t < b1) order[total[2] + t] = -1;
            if (b1 + total[1] + t < b0) order[b1 + total[1] + t] = -1;
            if (b0 + total[0] + t < items) order[b0 + total[0] + t] = -1;
        }
    }
    __syncthreads();
    const int64_t base = int64_t(blockIdx.x) * BK_BLOCK + int64_t(threadIdx.x) * BK_PER_THREAD;
    int cls[BK_PER_THREAD];
#pragma unroll
    for (int j = 0; j < BK_PER_THREAD; ++j) cls[j] = (base + j < n) ? update_class(engine_model, meas[base + j]) : -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < BK_PER_THREAD; ++j) mine += cls[j] == c;
        uint32_t rank;
        Scan(stmp).ExclusiveSum(mine, rank);
        __syncthreads();
        rank += start[c];
#pragma unroll
        for (int j = 0; j < BK_PER_THREAD; ++j)
            if (cls[j] == c) order[rank++] = int32_t(base + j);
    }
}
constexpr int BUCKET_INLINE_BLOCKS = 2048;   // up to 2 M filters: every scatter block sums 3 x 2048 counts at most
constexpr int64_t BUCKET_MIN_FILTERS = 16384;

// fills e->bucket_idx for the model ids in meas_dev; *items = entries of the list that a launch must cover (an upper bound
// known without reading anything back: every class is padded to a multiple of 4)
int build_model_buckets(ukfb_engine* e, const int32_t* meas_dev, int64_t* items) {
    const int64_t n = e->cap;
    const int nblocks = int((n + BK_BLOCK - 1) / BK_BLOCK);
    const size_t list = size_t(n) + 16;
    if (!e->bucket_idx || !e->bucket_counts) {
        // both or neither: the engine sees the pair only when every allocation succeeded (a half-published pair would let the
        // next call skip the allocation and launch through a null pointer)
        int32_t* idx = nullptr;
        uint32_t* cnt = nullptr;   // [3][blocks] counts, [3][blocks] exclusive prefix sums, [3] totals (+ 1 pad)
        if (hipMalloc(reinterpret_cast<void**>(&idx), list * sizeof(int32_t)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&cnt), (size_t(6) * nblocks + 4) * sizeof(uint32_t)) != hipSuccess) {
            if (idx) (void)hipFree(idx);
            return fail(UKFB_ERR_HIP, "model buckets: device allocation failed");
        }
        if (e->bucket_idx) (void)hipFree(e->bucket_idx);
        if (e->bucket_counts) (void)hipFree(e->bucket_counts);
        e->bucket_idx = idx;
        e->bucket_counts = cnt;
    }
    uint32_t* const before = e->bucket_counts + size_t(3) * nblocks;
    uint32_t* const total = e->bucket_counts + size_t(6) * nblocks;
    *items = (n + 3 + 3 + 3) / 4 * 4;   // sum of three counts each rounded up to 4 <= n + 9, itself rounded up to whole wavefronts
    // (the gaps of the list -- behind each class, up to *items -- are written by the scatter kernel: no memset)
    hipLaunchKernelGGL(bucket_count_kernel, dim3(nblocks), dim3(BK_THREADS), 0, ukfb::main_stream(e), meas_dev, n, e->model, e->bucket_counts,
                       nblocks);
    if (nblocks <= BUCKET_INLINE_BLOCKS) {
        hipLaunchKernelGGL(bucket_scatter_kernel<true>, dim3(nblocks), dim3(BK_THREADS), 0, ukfb::main_stream(e), meas_dev, n, e->model,
                           static_cast<const uint32_t*>(e->bucket_counts), static_cast<const uint32_t*>(before),
                           static_cast<const uint32_t*>(total), nblocks, e->bucket_idx, uint32_t(*items));
    } else {
        hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(BS_THREADS), 0, ukfb::main_stream(e),
                           static_cast<const uint32_t*>(e->bucket_counts), nblocks, before, total);
        hipLaunchKernelGGL(bucket_scatter_kernel<false>, dim3(nblocks), dim3(BK_THREADS), 0, ukfb::main_stream(e), meas_dev, n, e->model,
                           static_cast<const uint32_t*>(e->bucket_counts), static_cast<const uint32_t*>(before),
                           static_cast<const uint32_t*>(total), nblocks, e->bucket_idx, uint32_t(*items));
    }
    HIP_TRY(hipGetLastError());
    e->bucket_items = *items;
    return UKFB_OK;
}

__global__ void or_reduce_kernel(const uint32_t* st, int64_t n, uint32_t* out) {
    uint32_t v = 0;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        v |= st[i];
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicOr(out, v);
}

}  // namespace

namespace ukfb {
void set_error(const char* what, hipError_t err) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(err);
}
void set_error_text(const std::string& text) { g_last_error = text; }
}  // namespace ukfb

// Waiting by polling: hipEventSynchronize / hipStreamSynchronize sleep on an interrupt and were measured to
// wake up to ~55 ms late on a loaded host (1 run in 5), which is longer than the work being waited for.
// The wait is bounded: after a short spin the poll backs off to 50 us sleeps, and gives up with hipErrorNotReady
// after UKFB_WAIT_TIMEOUT_S seconds (default 120) so that a kernel that never finishes cannot pin a host core
// forever; the caller reports UKFB_ERR_HIP with "timed out".
static double wait_timeout_seconds() {
    static const double t = [] {
        const char* s = std::getenv("UKFB_WAIT_TIMEOUT_S");
        const double v = s ? std::atof(s) : 0.0;
        return v > 0.0 ? v : 120.0;
    }();
    return t;
}
template <class Query> static hipError_t wait_polling(Query&& query) {
    hipError_t r;
    g_wait_timed_out = false;
    for (int spin = 0; spin < 20000; ++spin)
        if ((r = query()) != hipErrorNotReady) return r;
    const auto t0 = std::chrono::steady_clock::now();
    while ((r = query()) == hipErrorNotReady) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_timeout_seconds()) {
            g_last_error = "timed out waiting for the engine's stream (UKFB_WAIT_TIMEOUT_S)";
            g_wait_timed_out = true;
            return hipErrorNotReady;
        }
    }
    return r;
}
namespace { hipError_t wait_stream_polling(hipStream_t s) { return wait_polling([s] { return hipStreamQuery(s); }); } }
static hipError_t wait_event_polling(hipEvent_t ev) { return wait_polling([ev] { return hipEventQuery(ev); }); }

namespace {
int engine_wait(ukfb_engine* e) {
    if (e->poisoned) return fail(UKFB_ERR_HIP, "engine poisoned by an earlier wait that timed out (UKFB_WAIT_TIMEOUT_S)");
    const hipError_t r = wait_stream_polling(ukfb::main_stream(e));
    if (r == hipSuccess) return UKFB_OK;
    if (r == hipErrorNotReady && g_wait_timed_out) {
        e->poisoned = true;
        g_wait_timed_out = false;
    } else {
        ukfb::set_error("waiting for the engine's stream", r);
    }
    return UKFB_ERR_HIP;
}
}  // namespace

extern "C" {

const char* ukfb_last_error(void) { return g_last_error.c_str(); }

int ukfb_default_config(ukfb_config* cfg) {
    if (!cfg) return UKFB_ERR_INVALID_ARG;
    cfg->mean_tol = 1e-6;
    cfg->mean_max_iter = 10000;   // ukfom's cap; the loop is wave-uniform and leaves on convergence, the cap costs nothing
    cfg->gate_chi2 = -1.0;
    cfg->min_time_delta = 1.0e-9;
    cfg->max_time_delta = std::numeric_limits<double>::max();
    cfg->lanes_per_filter = 16;
    cfg->bucket_models = 1;
    cfg->split_streams = 1;
    cfg->wide_arithmetic = 0;
    cfg->full_update_check = 0;
    return UKFB_OK;
}

static int create_engine(ukfb_engine* e, int64_t capacity, void* stream, bool use_given_stream) {
    if (stream || use_given_stream) {
        e->stream = static_cast<hipStream_t>(stream);   // (NULL with use_given_stream: the device's default stream)
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
        // the second stream of split launches (see ukf_launch.inc.hpp); a caller's stream keeps plain stream order
        HIP_TRY(hipStreamCreateWithFlags(&e->stream_b, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_a, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_b, hipEventDisableTiming));
    }
    const size_t n = size_t(capacity), ts = e->tsize;
    HIP_TRY(hipMalloc(&e->mu, n * e->S * ts));
    HIP_TRY(hipMalloc(&e->cov, n * e->PK * ts));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->status), n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->init), n));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->last_ts), n * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&e->Rn, (size_t(e->D) * e->D + NOISE_TABLE_PAD) * ts));
    HIP_TRY(hipMalloc(&e->in_a, n * 3 * ts));
    HIP_TRY(hipMalloc(&e->in_b, n * 3 * ts));
    HIP_TRY(hipMalloc(&e->z_stage, n * 3 * ts));
    HIP_TRY(hipMalloc(&e->Q_stage, n * 9 * ts));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->meas_stage), n * sizeof(int32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->active_stage), n));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->dt_stage), n * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->ts_stage), n * sizeof(int64_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->reduce_word), sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(e->mu, 0, n * e->S * ts, ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->cov, 0, n * e->PK * ts, ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->status, 0, n * sizeof(uint32_t), ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->init, 0, n, ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->last_ts, 0, n * sizeof(int64_t), ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->Rn, 0, size_t(e->D) * e->D * ts, ukfb::main_stream(e)));  // UnscentedKalmanFilter.hpp:29
    HIP_TRY(hipMemsetAsync(e->in_b, 0, n * 3 * ts, ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->z_stage, 0, n * 3 * ts, ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->Q_stage, 0, n * 9 * ts, ukfb::main_stream(e)));
    e->Rn_host.assign(size_t(e->D) * e->D, 0.0);
    int rc;
    if (e->model == UKFB_MODEL_POSE)  // acceleration.mu = NaN until set (PoseUKF.cpp:109)
        rc = fill_scalar(e, e->in_a, n * 3, std::numeric_limits<double>::quiet_NaN());
    else
        rc = fill_scalar(e, e->in_a, n * 3, 0.0);
    if (rc) return rc;
    HIP_TRY(hipEventCreate(&e->ev0));
    HIP_TRY(hipEventCreate(&e->ev1));
    ENGINE_SYNC(e);
    return rebuild_racc(e);
}

static int create_impl(ukfb_engine** out, int model, int precision, int64_t capacity, int device, void* stream,
                       bool use_given_stream) {
    if (!out || capacity <= 0 || (model != UKFB_MODEL_POSE && model != UKFB_MODEL_ORIENT) ||
        (precision != UKFB_F64 && precision != UKFB_F32))
        return fail(UKFB_ERR_INVALID_ARG, "ukfb_create: bad argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_last_error = "ukfb_create: no usable HIP device (the engine has no CPU fallback)";
        return UKFB_ERR_NO_DEVICE;
    }
    ON_DEVICE(device);
    ukfb_engine* e = new (std::nothrow) ukfb_engine();
    if (!e) return fail(UKFB_ERR_INVALID_ARG, "out of host memory");
    e->model = model;
    e->prec = precision;
    e->cap = capacity;
    e->device = device;
    e->S = model == UKFB_MODEL_POSE ? 13 : 14;
    e->D = model == UKFB_MODEL_POSE ? 12 : 13;
    e->PK = e->D * (e->D + 1) / 2;
    e->tsize = precision == UKFB_F64 ? 8 : 4;
    ukfb_default_config(&e->cfg);
    const int rc = create_engine(e, capacity, stream, use_given_stream);
    if (rc) {   // release the stream, the events and every buffer allocated so far; keep the error text
        const std::string msg = g_last_error;
        ukfb_destroy(e);
        g_last_error = msg;
        return rc;
    }
    *out = e;
    return UKFB_OK;
}

int ukfb_create(ukfb_engine** out, int model, int precision, int64_t capacity, int device, void* stream) {
    return create_impl(out, model, precision, capacity, device, stream, false);
}

int ukfb_create_on_stream(ukfb_engine** out, int model, int precision, int64_t capacity, int device, void* stream) {
    return create_impl(out, model, precision, capacity, device, stream, true);
}

int ukfb_layout_supported(int precision, int lanes_per_filter) {
    if (precision != UKFB_F64 && precision != UKFB_F32) return 0;
    if (lanes_per_filter == 0 || lanes_per_filter == 16) return 1;
    if (lanes_per_filter != 32 && lanes_per_filter != 64) return 0;
    return (precision == UKFB_F32 || UKFB_GENERIC_F64 != 0) ? 1 : 0;
}

int ukfb_destroy(ukfb_engine* e) {
    if (!e) return UKFB_OK;
    ukfb::DeviceScope device_scope(e->device);
    // bounded: a kernel that never finishes must not pin the host in the teardown either.  A poisoned engine (this wait or
    // an earlier one timed out) is abandoned as it is -- freeing memory under a kernel in flight would be worse than the leak
    (void)engine_wait(e);
    if (e->poisoned) {
        g_last_error = "ukfb_destroy: the engine's stream never drained; device memory left allocated, exit the process";
        delete e;
        return UKFB_ERR_HIP;
    }
    void* bufs[] = {e->mu, e->cov, e->status, e->init, e->last_ts, e->Rn, e->Racc, e->acc_cov_dev, e->in_a, e->in_b, e->z_stage,
                    e->Q_stage, e->meas_stage, e->active_stage, e->dt_stage, e->ts_stage, e->reduce_word, e->ev_dev, e->cvt_dev,
                    e->multi_dev, e->bucket_idx, e->bucket_counts};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    for (int s2 = 0; s2 < 2; ++s2) {
        if (e->zc_stage[s2]) (void)hipFree(e->zc_stage[s2]);
        if (e->Qc_stage[s2]) (void)hipFree(e->Qc_stage[s2]);
        if (e->ev_copy[s2]) (void)hipEventDestroy(e->ev_copy[s2]);
        if (e->ev_used[s2]) (void)hipEventDestroy(e->ev_used[s2]);
    }
    if (e->cvt_copy) (void)hipFree(e->cvt_copy);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->ev_a) (void)hipEventDestroy(e->ev_a);
    if (e->ev_b) (void)hipEventDestroy(e->ev_b);
    if (e->stream_b) (void)hipStreamDestroy(e->stream_b);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return UKFB_OK;
}

int ukfb_set_config(ukfb_engine* e, const ukfb_config* cfg) {
    if (!e || !cfg) return UKFB_ERR_INVALID_ARG;
    ukfb_config c = *cfg;
    if (c.lanes_per_filter == 0) c.lanes_per_filter = 16;
    if (c.lanes_per_filter != 16 && c.lanes_per_filter != 32 && c.lanes_per_filter != 64)
        return fail(UKFB_ERR_INVALID_ARG, "lanes_per_filter must be 16, 32 or 64");
    if (!ukfb_layout_supported(e->prec, c.lanes_per_filter))
        return fail(UKFB_ERR_INVALID_ARG, "lanes_per_filter 32 / 64 in fp64 is a diagnostic build option (make GENERIC_F64=1)");
    if (c.mean_max_iter < 1) return fail(UKFB_ERR_INVALID_ARG, "mean_max_iter must be >= 1");
    if (c.wide_arithmetic != 0 && c.wide_arithmetic != 1) return fail(UKFB_ERR_INVALID_ARG, "wide_arithmetic must be 0 or 1");
    if (c.full_update_check != 0 && c.full_update_check != 1) return fail(UKFB_ERR_INVALID_ARG, "full_update_check must be 0 or 1");
    if (c.wide_arithmetic && e->prec == UKFB_F32 && c.lanes_per_filter != 16)
        return fail(UKFB_ERR_INVALID_ARG, "wide_arithmetic runs on the tuned layout only (lanes_per_filter 16)");
    e->cfg = c;
    return UKFB_OK;
}

int ukfb_get_config(const ukfb_engine* e, ukfb_config* cfg) {
    if (!e || !cfg) return UKFB_ERR_INVALID_ARG;
    *cfg = e->cfg;
    return UKFB_OK;
}

int ukfb_sync(ukfb_engine* e) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    return engine_wait(e);
}

int ukfb_describe(const ukfb_engine* e, int* model, int* precision, int64_t* capacity, int* S, int* D, int* PK) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (model) *model = e->model;
    if (precision) *precision = e->prec;
    if (capacity) *capacity = e->cap;
    if (S) *S = e->S;
    if (D) *D = e->D;
    if (PK) *PK = e->PK;
    return UKFB_OK;
}

int ukfb_initialize(ukfb_engine* e, int64_t first, int64_t count, const double* mu, const double* cov) {
    if (!range_ok(e, first, count) || !mu || !cov) return fail(UKFB_ERR_OUT_OF_RANGE, "ukfb_initialize: bad range");
    ON_DEVICE(e->device);
    const int D = e->D, PK = e->PK;
    int rc = upload(e, e->mu, size_t(first) * e->S, mu, size_t(count) * e->S);
    if (rc) return rc;
    // covariances: the full matrices cross PCIe as they are, the packing (and narrowing) runs on the device
    for (int64_t lo = 0; lo < count; lo += COV_CHUNK) {
        const int64_t m = std::min(COV_CHUNK, count - lo);
        const size_t nfull = size_t(m) * D * D;
        rc = ensure_scratch(e, nfull * sizeof(double));
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(e->cvt_dev, cov + size_t(lo) * D * D, nfull * sizeof(double), hipMemcpyHostToDevice, ukfb::main_stream(e)));
        const unsigned blocks = unsigned((size_t(m) * PK + 255) / 256);
        if (e->prec == UKFB_F64)
            hipLaunchKernelGGL(pack_cov_kernel<double>, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), static_cast<const double*>(e->cvt_dev),
                               static_cast<double*>(e->cov) + size_t(first + lo) * PK, m, D, PK);
        else
            hipLaunchKernelGGL(pack_cov_kernel<float>, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), static_cast<const double*>(e->cvt_dev),
                               static_cast<float*>(e->cov) + size_t(first + lo) * PK, m, D, PK);
        HIP_TRY(hipGetLastError());
        ENGINE_SYNC(e);   // the scratch is reused by the next chunk
    }
    HIP_TRY(hipMemsetAsync(e->init + first, 1, size_t(count), ukfb::main_stream(e)));
    HIP_TRY(hipMemsetAsync(e->last_ts + first, 0, size_t(count) * sizeof(int64_t), ukfb::main_stream(e)));
    ENGINE_SYNC(e);
    return UKFB_OK;
}

int ukfb_get_state(ukfb_engine* e, int64_t first, int64_t count, double* mu, double* cov, uint8_t* initialised) {
    if (!range_ok(e, first, count)) return fail(UKFB_ERR_OUT_OF_RANGE, "ukfb_get_state: bad range");
    ON_DEVICE(e->device);
    const int D = e->D, PK = e->PK;
    int rc;
    if (mu) {
        rc = download(e, e->mu, size_t(first) * e->S, mu, size_t(count) * e->S);
        if (rc) return rc;
    }
    if (cov) {   // mirrored to full matrices (and widened) on the device, chunk by chunk
        for (int64_t lo = 0; lo < count; lo += COV_CHUNK) {
            const int64_t m = std::min(COV_CHUNK, count - lo);
            const size_t nfull = size_t(m) * D * D;
            rc = ensure_scratch(e, nfull * sizeof(double));
            if (rc) return rc;
            const unsigned blocks = unsigned((nfull + 255) / 256);
            if (e->prec == UKFB_F64)
                hipLaunchKernelGGL(unpack_cov_kernel<double>, dim3(blocks), dim3(256), 0, ukfb::main_stream(e),
                                   static_cast<const double*>(e->cov) + size_t(first + lo) * PK, static_cast<double*>(e->cvt_dev), m, D, PK);
            else
                hipLaunchKernelGGL(unpack_cov_kernel<float>, dim3(blocks), dim3(256), 0, ukfb::main_stream(e),
                                   static_cast<const float*>(e->cov) + size_t(first + lo) * PK, static_cast<double*>(e->cvt_dev), m, D, PK);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(cov + size_t(lo) * D * D, e->cvt_dev, nfull * sizeof(double), hipMemcpyDeviceToHost, ukfb::main_stream(e)));
            ENGINE_SYNC(e);
        }
    }
    if (initialised) {
        rc = download_raw(e, e->init + first, initialised, size_t(count));
        if (rc) return rc;
    }
    return UKFB_OK;
}

int ukfb_get_status(ukfb_engine* e, int64_t first, int64_t count, uint32_t* status) {
    if (!range_ok(e, first, count) || !status) return fail(UKFB_ERR_OUT_OF_RANGE, "ukfb_get_status: bad range");
    ON_DEVICE(e->device);
    return download_raw(e, e->status + first, status, size_t(count));
}

int ukfb_get_status_summary(ukfb_engine* e, uint32_t* or_of_all) {
    if (!e || !or_of_all) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    HIP_TRY(hipMemsetAsync(e->reduce_word, 0, sizeof(uint32_t), ukfb::main_stream(e)));
    const int blocks = int(std::min<int64_t>((e->cap + 255) / 256, 1024));
    hipLaunchKernelGGL(or_reduce_kernel, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), e->status, e->cap, e->reduce_word);
    HIP_TRY(hipGetLastError());
    return download_raw(e, e->reduce_word, or_of_all, 1);
}

int ukfb_set_last_measurement_time(ukfb_engine* e, int64_t first, int64_t count, const int64_t* t_us) {
    if (!range_ok(e, first, count) || !t_us) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    ON_DEVICE(e->device);
    return upload_raw(e, e->last_ts + first, t_us, size_t(count));
}

int ukfb_get_last_measurement_time(ukfb_engine* e, int64_t first, int64_t count, int64_t* t_us) {
    if (!range_ok(e, first, count) || !t_us) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    ON_DEVICE(e->device);
    return download_raw(e, e->last_ts + first, t_us, size_t(count));
}

int ukfb_device_views(ukfb_engine* e, void** mu_dev, void** cov_packed_dev, uint32_t** status_dev) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (mu_dev) *mu_dev = e->mu;
    if (cov_packed_dev) *cov_packed_dev = e->cov;
    if (status_dev) *status_dev = e->status;
    return UKFB_OK;
}

int ukfb_set_process_noise(ukfb_engine* e, const double* R) {
    if (!e || !R) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    const size_t dd = size_t(e->D) * e->D;
    e->Rn_host.assign(R, R + dd);
    {   // blocks [0:3,0:3] and [3:6,3:6] are the ones predictionStepImpl rotates (PoseUKF.cpp:184-185, OrientationUKF.cpp:84-85)
        bool iso = true;
        for (int b = 0; b < 6; b += 3)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    const double v = R[size_t(b + r) * e->D + (b + c)];
                    iso = iso && (r == c ? v == R[size_t(b) * e->D + b] : v == 0.0);
                }
        e->noise_iso = iso;
    }
    if (e->Rn_per_filter) {
        std::vector<double> all(size_t(e->cap) * dd);
        for (int64_t i = 0; i < e->cap; ++i) std::memcpy(all.data() + size_t(i) * dd, R, dd * sizeof(double));
        int rc = upload(e, e->Rn, 0, all.data(), all.size());
        return rc ? rc : rebuild_racc(e);
    }
    int rc = upload(e, e->Rn, 0, R, dd);
    return rc ? rc : rebuild_racc(e);
}

int ukfb_set_process_noise_per_filter(ukfb_engine* e, int64_t first, int64_t count, const double* R) {
    if (!range_ok(e, first, count) || !R) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    ON_DEVICE(e->device);
    const size_t dd = size_t(e->D) * e->D;
    if (!e->Rn_per_filter) {
        void* big = nullptr;
        HIP_TRY(hipMalloc(&big, (size_t(e->cap) * dd + NOISE_TABLE_PAD) * e->tsize));
        ENGINE_SYNC(e);
        HIP_TRY(hipFree(e->Rn));
        e->Rn = big;
        e->Rn_per_filter = true;
        std::vector<double> all(size_t(e->cap) * dd);
        for (int64_t i = 0; i < e->cap; ++i)
            std::memcpy(all.data() + size_t(i) * dd, e->Rn_host.data(), dd * sizeof(double));
        int rc = upload(e, e->Rn, 0, all.data(), all.size());
        if (rc) return rc;
    }
    int rc = upload(e, e->Rn, size_t(first) * dd, R, size_t(count) * dd);
    return rc ? rc : rebuild_racc(e);
}

int ukfb_get_process_noise(ukfb_engine* e, int64_t filter, double* R) {
    if (!range_ok(e, filter, 1) || !R) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    ON_DEVICE(e->device);
    const size_t dd = size_t(e->D) * e->D;
    return download(e, e->Rn, e->Rn_per_filter ? size_t(filter) * dd : 0, R, dd);
}

int ukfb_pose_set_acceleration(ukfb_engine* e, int64_t first, int64_t count, const double* acc_mu,
                               const double* acc_cov) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (e->model != UKFB_MODEL_POSE) return fail(UKFB_ERR_WRONG_MODEL, "Pose engines only");
    if (!range_ok(e, first, count)) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    ON_DEVICE(e->device);
    if (acc_cov) {
        std::memcpy(e->acc_cov, acc_cov, 9 * sizeof(double));
        int rc = rebuild_racc(e);
        if (rc) return rc;
    }
    if (acc_mu) return upload(e, e->in_a, size_t(first) * 3, acc_mu, size_t(count) * 3);
    return UKFB_OK;
}

int ukfb_pose_bind_acceleration_dev(ukfb_engine* e, const void* acc_mu_dev) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (e->model != UKFB_MODEL_POSE) return fail(UKFB_ERR_WRONG_MODEL, "Pose engines only");
    e->in_a_bound = acc_mu_dev;
    return UKFB_OK;
}

int ukfb_orient_set_params(ukfb_engine* e, double gyro_bias_tau, double acc_bias_tau, const double earth_rotation[3]) {
    if (!e || !earth_rotation) return UKFB_ERR_INVALID_ARG;
    if (e->model != UKFB_MODEL_ORIENT) return fail(UKFB_ERR_WRONG_MODEL, "Orient engines only");
    e->tau_g = gyro_bias_tau;
    e->tau_a = acc_bias_tau;
    for (int k = 0; k < 3; ++k) e->earth[k] = earth_rotation[k];
    return UKFB_OK;
}

int ukfb_orient_set_inputs(ukfb_engine* e, int64_t first, int64_t count, const double* gyro, const double* acc) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (e->model != UKFB_MODEL_ORIENT) return fail(UKFB_ERR_WRONG_MODEL, "Orient engines only");
    if (!range_ok(e, first, count)) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    ON_DEVICE(e->device);
    // checkMeasurment (OrientationUKF.cpp:55,61): a non-finite row keeps the previously latched value
    auto latch = [&](void* dev, const double* src) -> int {
        bool all_finite = true;
        for (int64_t i = 0; i < count * 3 && all_finite; ++i) all_finite = std::isfinite(src[i]);
        if (all_finite) return upload(e, dev, size_t(first) * 3, src, size_t(count) * 3);
        std::vector<double> cur(static_cast<size_t>(count) * 3, 0.0);
        int rc = download(e, dev, size_t(first) * 3, cur.data(), cur.size());
        if (rc) return rc;
        std::vector<uint32_t> st(static_cast<size_t>(count), 0u);
        rc = download_raw(e, e->status + first, st.data(), st.size());
        if (rc) return rc;
        for (int64_t i = 0; i < count; ++i) {
            const double* r = src + i * 3;
            if (std::isfinite(r[0]) && std::isfinite(r[1]) && std::isfinite(r[2]))
                std::memcpy(cur.data() + i * 3, r, 3 * sizeof(double));
            else
                st[size_t(i)] |= UKFB_ST_ERR_NONFINITE_MEAS;
        }
        rc = upload(e, dev, size_t(first) * 3, cur.data(), cur.size());
        if (rc) return rc;
        return upload_raw(e, e->status + first, st.data(), st.size());
    };
    int rc = UKFB_OK;
    if (gyro) rc = latch(e->in_b, gyro);
    if (rc) return rc;
    if (acc) rc = latch(e->in_a, acc);
    return rc;
}

int ukfb_orient_bind_inputs_dev(ukfb_engine* e, const void* gyro_dev, const void* acc_dev) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (e->model != UKFB_MODEL_ORIENT) return fail(UKFB_ERR_WRONG_MODEL, "Orient engines only");
    e->in_b_bound = gyro_dev;
    e->in_a_bound = acc_dev;
    return UKFB_OK;
}

int ukfb_orient_get_rotation_rate(ukfb_engine* e, int64_t first, int64_t count, double* out) {
    if (!e || !out) return UKFB_ERR_INVALID_ARG;
    if (e->model != UKFB_MODEL_ORIENT) return fail(UKFB_ERR_WRONG_MODEL, "Orient engines only");
    if (!range_ok(e, first, count)) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    ON_DEVICE(e->device);
    // rotation_rate.mu - bias_gyro - q^-1 * earth_rotation (OrientationUKF.cpp:74-77): a read-out of
    // latched input and mean, not part of the predict/update arithmetic, evaluated host-side.
    std::vector<double> mu(size_t(count) * 14), w(size_t(count) * 3);
    int rc = download(e, e->mu, size_t(first) * 14, mu.data(), mu.size());
    if (rc) return rc;
    const void* gy = e->in_b_bound ? e->in_b_bound : e->in_b;
    rc = download(e, gy, size_t(first) * 3, w.data(), w.size());
    if (rc) return rc;
    for (int64_t i = 0; i < count; ++i) {
        const double* m = mu.data() + i * 14;
        const double n2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3];
        const double q[4] = {-m[0] / n2, -m[1] / n2, -m[2] / n2, m[3] / n2};
        const double* v = e->earth;
        double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
        ux += ux; uy += uy; uz += uz;
        const double r0 = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
        const double r1 = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
        const double r2 = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
        out[i * 3 + 0] = w[size_t(i) * 3 + 0] - m[7] - r0;
        out[i * 3 + 1] = w[size_t(i) * 3 + 1] - m[8] - r1;
        out[i * 3 + 2] = w[size_t(i) * 3 + 2] - m[9] - r2;
    }
    return UKFB_OK;
}

// ---- BodyStateMeasurement adapters ----------------------------------------------------------------
int ukfb_pose_export_body_states(ukfb_engine* e, int64_t first, int64_t count, double* out) {
    if (!e || !out) return UKFB_ERR_INVALID_ARG;
    if (e->model != UKFB_MODEL_POSE) return fail(UKFB_ERR_WRONG_MODEL, "Pose engines only");
    if (!range_ok(e, first, count)) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    if (count == 0) return UKFB_OK;
    ON_DEVICE(e->device);
    return e->prec == UKFB_F64 ? export_body_states<double>(e, first, count, out) : export_body_states<float>(e, first, count, out);
}

int ukfb_pose_import_body_states(ukfb_engine* e, int64_t first, int64_t count, const double* in) {
    if (!e || !in) return UKFB_ERR_INVALID_ARG;
    if (e->model != UKFB_MODEL_POSE) return fail(UKFB_ERR_WRONG_MODEL, "Pose engines only");
    if (!range_ok(e, first, count)) return fail(UKFB_ERR_OUT_OF_RANGE, "bad range");
    if (count == 0) return UKFB_OK;
    ON_DEVICE(e->device);
    return e->prec == UKFB_F64 ? import_body_states<double>(e, first, count, in) : import_body_states<float>(e, first, count, in);
}

// ---- predict ------------------------------------------------------------------------------------
int ukfb_predict(ukfb_engine* e, double dt) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.dt_uniform = dt;
    return launch(e, r);
}

int ukfb_predict_dt_dev(ukfb_engine* e, const double* dt_dev) {
    if (!e || !dt_dev) return UKFB_ERR_INVALID_ARG;
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.dt_dev = dt_dev;
    return launch(e, r);
}

int ukfb_predict_dt(ukfb_engine* e, const double* dt) {
    if (!e || !dt) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    int rc = upload_raw(e, e->dt_stage, dt, size_t(e->cap));
    if (rc) return rc;
    return ukfb_predict_dt_dev(e, e->dt_stage);
}

int ukfb_predict_timestamps_dev(ukfb_engine* e, const int64_t* ts_us_dev) {
    if (!e || !ts_us_dev) return UKFB_ERR_INVALID_ARG;
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.ts_dev = ts_us_dev;
    return launch(e, r);
}

int ukfb_predict_timestamps(ukfb_engine* e, const int64_t* ts_us) {
    if (!e || !ts_us) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    int rc = upload_raw(e, e->ts_stage, ts_us, size_t(e->cap));
    if (rc) return rc;
    return ukfb_predict_timestamps_dev(e, e->ts_stage);
}

// ---- update -------------------------------------------------------------------------------------
int ukfb_update_dev(ukfb_engine* e, int meas_model_uniform, const int32_t* meas_model_dev, const void* z_dev,
                    const void* Q_dev) {
    if (!e || !z_dev || !Q_dev) return UKFB_ERR_INVALID_ARG;
    if (!meas_model_dev && !meas_model_ok(e, meas_model_uniform))
        return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    ukfb::LaunchReq r;
    r.do_update = true;
    r.meas_uniform = meas_model_uniform;
    r.meas_dev = meas_model_dev;
    r.z_dev = z_dev;
    r.Q_dev = Q_dev;
    return launch(e, r);
}

int ukfb_update(ukfb_engine* e, int meas_model, const double* z, const double* Q, const uint8_t* active) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (!meas_model_ok(e, meas_model)) return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    ON_DEVICE(e->device);
    if (!active) {   // the common form: samples uploaded on the copy stream while the previous kernel runs (see stage_cycle_inputs)
        Staged sg;
        const int src = stage_cycle_inputs(e, z, Q, size_t(e->cap) * 9, &sg);
        if (src) return src;
        ukfb::LaunchReq r;
        r.do_update = true;
        r.meas_uniform = meas_model;
        r.z_dev = sg.z;
        r.Q_dev = sg.Q;
        r.wait_event = sg.ready;
        r.done_event = sg.done;
        r.no_split = true;
        const int lrc = launch(e, r);
        if (!lrc) e->stage_busy[sg.slot] = true;
        return lrc;
    }
    int rc = stage_measurements(e, z, Q, nullptr, active);
    if (rc) return rc;
    ukfb::LaunchReq r;
    r.do_update = true;
    r.meas_uniform = meas_model;
    r.z_dev = e->z_stage;
    r.Q_dev = e->Q_stage;
    r.active_dev = active ? e->active_stage : nullptr;
    return launch(e, r);
}

int ukfb_update_mixed(ukfb_engine* e, const int32_t* meas_model, const double* z, const double* Q) {
    if (!e || !meas_model) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    int rc = stage_measurements(e, z, Q, meas_model, nullptr);
    if (rc) return rc;
    ukfb::LaunchReq r;
    r.do_update = true;
    r.meas_dev = e->meas_stage;
    r.z_dev = e->z_stage;
    r.Q_dev = e->Q_stage;
    return launch(e, r);
}

// ---- fused cycle --------------------------------------------------------------------------------
int ukfb_cycle_dev(ukfb_engine* e, double dt, int meas_model_uniform, const int32_t* meas_model_dev, const void* z_dev,
                   const void* Q_dev) {
    if (!e || !z_dev || !Q_dev) return UKFB_ERR_INVALID_ARG;
    if (!meas_model_dev && !meas_model_ok(e, meas_model_uniform))
        return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.do_update = true;
    r.dt_uniform = dt;
    r.meas_uniform = meas_model_uniform;
    r.meas_dev = meas_model_dev;
    r.z_dev = z_dev;
    r.Q_dev = Q_dev;
    if (meas_model_dev && e->cfg.bucket_models && e->cfg.lanes_per_filter == 16 && e->cap >= BUCKET_MIN_FILTERS &&
        e->cap <= 0x7fffffff - 16) {
        // Mixed stream: wavefronts of four neighbouring filters would each run the most expensive path any of the four
        // needs.  Group the filters by the class of their update first (device side, nothing read back), then ONE indirect
        // launch over the grouped list: every wavefront is class-uniform, the sigma-point branch of the update is taken by
        // the wavefronts of that class only.
        if (e->poisoned) return fail(UKFB_ERR_HIP, "engine poisoned by an earlier wait that timed out (UKFB_WAIT_TIMEOUT_S)");
        ON_DEVICE(e->device);
        int64_t items = 0;
        const int rc = build_model_buckets(e, meas_model_dev, &items);
        if (rc) return rc;
        r.filter_index_dev = e->bucket_idx;
        r.inputs_by_filter = true;
        r.n_items = items;
    }
    return launch(e, r);
}

// meas_dev != NULL: per-filter model ids, a ring [slots][capacity] like z and Q (meas_model is then ignored)
static int cycle_multi_impl(ukfb_engine* e, int cycles, double dt, int meas_model, const int32_t* meas_dev, int slots,
                            int first_slot, const void* in_a_dev, const void* in_b_dev, const void* z_dev, const void* Q_dev) {
    if (!e || !z_dev || !Q_dev) return UKFB_ERR_INVALID_ARG;
    if (cycles < 0 || slots < 1 || first_slot < 0 || first_slot >= slots)
        return fail(UKFB_ERR_INVALID_ARG, "cycles >= 0, slots >= 1, 0 <= first_slot < slots");
    if (!meas_dev && !meas_model_ok(e, meas_model))
        return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    if (cycles == 0) return UKFB_OK;
    if (e->cfg.lanes_per_filter != 16) {
        // the one-wavefront-per-filter layouts have no multi-cycle kernel: one launch per cycle, same results.  The
        // status word must still be the OR over the cycles.
        const size_t w = e->tsize;
        const void* keep_a = e->in_a_bound;
        const void* keep_b = e->in_b_bound;
        int rc = UKFB_OK;
        for (int c = 0; c < cycles && rc == UKFB_OK; ++c) {
            const size_t s = size_t((first_slot + c) % slots) * size_t(e->cap);
            if (in_a_dev) e->in_a_bound = static_cast<const char*>(in_a_dev) + s * 3 * w;
            if (in_b_dev) e->in_b_bound = static_cast<const char*>(in_b_dev) + s * 3 * w;
            ukfb::LaunchReq r;
            r.do_predict = true;
            r.do_update = true;
            r.dt_uniform = dt;
            r.meas_uniform = meas_model;
            r.meas_dev = meas_dev ? meas_dev + s : nullptr;
            r.z_dev = static_cast<const char*>(z_dev) + s * 3 * w;
            r.Q_dev = static_cast<const char*>(Q_dev) + s * 9 * w;
            r.status_accumulate = c > 0;
            rc = launch(e, r);
        }
        e->in_a_bound = keep_a;
        e->in_b_bound = keep_b;
        return rc;
    }
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.do_update = true;
    r.dt_uniform = dt;
    r.meas_uniform = meas_model;
    r.meas_dev = meas_dev;
    r.z_dev = z_dev;
    r.Q_dev = Q_dev;
    r.cycles = cycles;
    r.slots = slots;
    r.first_slot = first_slot;
    r.in_a_slots = in_a_dev;
    r.in_b_slots = in_b_dev;
    return launch(e, r);
}

int ukfb_cycle_multi_dev(ukfb_engine* e, int cycles, double dt, int meas_model, int slots, int first_slot,
                         const void* in_a_dev, const void* in_b_dev, const void* z_dev, const void* Q_dev) {
    return cycle_multi_impl(e, cycles, dt, meas_model, nullptr, slots, first_slot, in_a_dev, in_b_dev, z_dev, Q_dev);
}

int ukfb_cycle_multi_mixed_dev(ukfb_engine* e, int cycles, double dt, int slots, int first_slot, const void* in_a_dev,
                               const void* in_b_dev, const int32_t* meas_model_dev, const void* z_dev, const void* Q_dev) {
    if (!meas_model_dev) return UKFB_ERR_INVALID_ARG;
    return cycle_multi_impl(e, cycles, dt, -1, meas_model_dev, slots, first_slot, in_a_dev, in_b_dev, z_dev, Q_dev);
}

int ukfb_cycle_schedule_dev(ukfb_engine* e, int cycles, const double* dt, const int32_t* meas_model, int slots, int first_slot,
                            const void* in_a_dev, const void* in_b_dev, const void* z_dev, const void* Q_dev) {
    if (!e || !dt || !meas_model || !z_dev || !Q_dev) return UKFB_ERR_INVALID_ARG;
    if (cycles < 0 || slots < 1 || first_slot < 0 || first_slot >= slots)
        return fail(UKFB_ERR_INVALID_ARG, "cycles >= 0, slots >= 1, 0 <= first_slot < slots");
    for (int c = 0; c < cycles; ++c)
        if (meas_model[c] >= 0 && !meas_model_ok(e, meas_model[c]))
            return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    const bool tuned = e->cfg.lanes_per_filter == 16;
    const size_t w = e->tsize;
    const void* keep_a = e->in_a_bound;
    const void* keep_b = e->in_b_bound;
    int rc = UKFB_OK;
    // the tuned layout: launches of up to UKFB_MAX_MULTI_CYCLES cycles, the schedule travels in the kernel arguments;
    // the one-wavefront-per-filter layouts: one launch per cycle
    const int chunk = tuned ? UKFB_MAX_MULTI_CYCLES : 1;
    for (int c0 = 0; c0 < cycles && rc == UKFB_OK; c0 += chunk) {
        const int nc = (cycles - c0 < chunk) ? (cycles - c0) : chunk;
        ukfb::LaunchReq r;
        r.do_predict = true;
        r.status_accumulate = c0 > 0;         // the status word is the OR over ALL cycles of the call
        if (tuned) {
            r.do_update = true;
            r.z_dev = z_dev;
            r.Q_dev = Q_dev;
            r.cycles = nc;
            r.slots = slots;
            r.first_slot = (first_slot + c0) % slots;
            r.in_a_slots = in_a_dev;
            r.in_b_slots = in_b_dev;
            r.sched_dt = dt + c0;
            r.sched_model = meas_model + c0;
        } else {
            const size_t s = size_t((first_slot + c0) % slots) * size_t(e->cap);
            if (in_a_dev) e->in_a_bound = static_cast<const char*>(in_a_dev) + s * 3 * w;
            if (in_b_dev) e->in_b_bound = static_cast<const char*>(in_b_dev) + s * 3 * w;
            r.do_update = meas_model[c0] >= 0;
            r.dt_uniform = dt[c0];
            r.meas_uniform = meas_model[c0];
            r.z_dev = static_cast<const char*>(z_dev) + s * 3 * w;
            r.Q_dev = static_cast<const char*>(Q_dev) + s * 9 * w;
        }
        rc = launch(e, r);
    }
    e->in_a_bound = keep_a;
    e->in_b_bound = keep_b;
    return rc;
}

int ukfb_cycle_multi(ukfb_engine* e, int cycles, double dt, int meas_model, const double* in_a, const double* in_b,
                     const double* z, const double* Q) {
    if (!e || !z || !Q) return UKFB_ERR_INVALID_ARG;
    if (cycles < 0) return fail(UKFB_ERR_INVALID_ARG, "cycles >= 0");
    if (!meas_model_ok(e, meas_model)) return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    if (cycles == 0) return UKFB_OK;
    ON_DEVICE(e->device);
    // device rings of this call's samples, one slot per cycle: [z | Q | in_a | in_b]
    const size_t per = size_t(cycles) * size_t(e->cap);
    const size_t nz = per * 3, nq = per * 9, na = in_a ? per * 3 : 0, nb = in_b ? per * 3 : 0;
    const size_t bytes = (nz + nq + na + nb) * e->tsize;
    if (e->multi_bytes < bytes) {
        ENGINE_SYNC(e);
        if (e->multi_dev) HIP_TRY(hipFree(e->multi_dev));
        e->multi_dev = nullptr;
        e->multi_bytes = 0;
        HIP_TRY(hipMalloc(&e->multi_dev, bytes));
        e->multi_bytes = bytes;
    }
    char* base = static_cast<char*>(e->multi_dev);
    void* z_dev = base;
    void* Q_dev = base + nz * e->tsize;
    void* a_dev = in_a ? base + (nz + nq) * e->tsize : nullptr;
    void* b_dev = in_b ? base + (nz + nq + na) * e->tsize : nullptr;
    int rc = upload(e, z_dev, 0, z, nz);
    if (!rc) rc = upload(e, Q_dev, 0, Q, nq);
    if (!rc && in_a) rc = upload(e, a_dev, 0, in_a, na);
    if (!rc && in_b) rc = upload(e, b_dev, 0, in_b, nb);
    if (rc) return rc;
    return ukfb_cycle_multi_dev(e, cycles, dt, meas_model, cycles, 0, a_dev, b_dev, z_dev, Q_dev);
}

// ---- batch-uniform measurement covariance -------------------------------------------------------
int ukfb_cycle_uniform_q_dev(ukfb_engine* e, double dt, int meas_model, const void* z_dev, const void* Q9_dev) {
    if (!e || !z_dev || !Q9_dev) return UKFB_ERR_INVALID_ARG;
    if (!meas_model_ok(e, meas_model)) return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.do_update = true;
    r.dt_uniform = dt;
    r.meas_uniform = meas_model;
    r.z_dev = z_dev;
    r.Q_dev = Q9_dev;
    r.q_uniform = true;
    return launch(e, r);
}

int ukfb_cycle_uniform_q(ukfb_engine* e, double dt, int meas_model, const double* z, const double* Q9) {
    if (!e || !z || !Q9) return UKFB_ERR_INVALID_ARG;
    if (!meas_model_ok(e, meas_model)) return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    ON_DEVICE(e->device);
    Staged sg;
    const int rc = stage_cycle_inputs(e, z, Q9, 9, &sg);
    if (rc) return rc;
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.do_update = true;
    r.dt_uniform = dt;
    r.meas_uniform = meas_model;
    r.z_dev = sg.z;
    r.Q_dev = sg.Q;
    r.q_uniform = true;
    r.wait_event = sg.ready;
    r.done_event = sg.done;
    r.no_split = true;
    const int lrc = launch(e, r);
    if (!lrc) e->stage_busy[sg.slot] = true;
    return lrc;
}

int ukfb_update_uniform_q(ukfb_engine* e, int meas_model, const double* z, const double* Q9, const uint8_t* active) {
    if (!e || !z || !Q9) return UKFB_ERR_INVALID_ARG;
    if (!meas_model_ok(e, meas_model)) return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    ON_DEVICE(e->device);
    int rc = upload(e, e->z_stage, 0, z, size_t(e->cap) * 3);
    if (!rc) rc = upload(e, e->Q_stage, 0, Q9, 9);
    if (!rc && active) rc = upload_raw(e, e->active_stage, active, size_t(e->cap));
    if (rc) return rc;
    ukfb::LaunchReq r;
    r.do_update = true;
    r.meas_uniform = meas_model;
    r.z_dev = e->z_stage;
    r.Q_dev = e->Q_stage;
    r.q_uniform = true;
    r.active_dev = active ? e->active_stage : nullptr;
    return launch(e, r);
}

int ukfb_cycle(ukfb_engine* e, double dt, int meas_model, const double* z, const double* Q) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (!meas_model_ok(e, meas_model)) return fail(UKFB_ERR_WRONG_MODEL, "measurement model id not valid for this engine");
    ON_DEVICE(e->device);
    Staged sg;
    const int rc = stage_cycle_inputs(e, z, Q, size_t(e->cap) * 9, &sg);
    if (rc) return rc;
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.do_update = true;
    r.dt_uniform = dt;
    r.meas_uniform = meas_model;
    r.z_dev = sg.z;
    r.Q_dev = sg.Q;
    r.wait_event = sg.ready;
    r.done_event = sg.done;
    r.no_split = true;
    const int lrc = launch(e, r);
    if (!lrc) e->stage_busy[sg.slot] = true;
    return lrc;
}

int ukfb_cycle_timestamps_dev(ukfb_engine* e, const int64_t* ts_us_dev, const int32_t* meas_model_dev, const void* z_dev,
                              const void* Q_dev) {
    if (!e || !ts_us_dev || !meas_model_dev || !z_dev || !Q_dev) return UKFB_ERR_INVALID_ARG;
    ukfb::LaunchReq r;
    r.do_predict = true;
    r.do_update = true;
    r.ts_dev = ts_us_dev;
    r.meas_dev = meas_model_dev;
    r.z_dev = z_dev;
    r.Q_dev = Q_dev;
    return launch(e, r);
}

int ukfb_cycle_timestamps(ukfb_engine* e, const int64_t* ts_us, const int32_t* meas_model, const double* z, const double* Q) {
    if (!e || !ts_us || !meas_model) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    int rc = stage_measurements(e, z, Q, meas_model, nullptr);
    if (rc) return rc;
    rc = upload_raw(e, e->ts_stage, ts_us, size_t(e->cap));
    if (rc) return rc;
    return ukfb_cycle_timestamps_dev(e, e->ts_stage, e->meas_stage, e->z_stage, e->Q_stage);
}

// ---- time-ordered asynchronous measurement stream -------------------------------------------------
}  // extern "C"

// Device pipeline shared by both entry points.  Events are in device memory (z, Q in precision S); everything
// else lives in the engine's grow-only workspace `ev_dev`.  Ordering: radix sort by timestamp, then stable radix
// sort by filter index (hipCUB) = per-filter time order with arrival order for equal stamps; the rank of a sample
// inside its filter's run is its round.
namespace {
struct Carver {   // bump allocator over the workspace (256-byte aligned pieces)
    char* base;
    size_t used = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    template <class P> P* take(size_t count) {
        P* p = base ? reinterpret_cast<P*>(base + used) : nullptr;
        used += (count * sizeof(P) + 255) / 256 * 256;
        return p;
    }
};

template <class S>
int process_events_device(ukfb_engine* e, int64_t n, const int64_t* d_f, const int64_t* d_t, const int32_t* d_m, const S* d_z,
                          const S* d_q, size_t ws_offset, uint32_t* status_or, int64_t* rounds) {
    if (e->cap > 0x7fffffff) return fail(UKFB_ERR_INVALID_ARG, "ukfb_process_events: capacity exceeds 32-bit filter indices");
    const int bits_f = std::max(1, int(std::ceil(std::log2(double(std::max<int64_t>(e->cap, 2))))));
    size_t tmp_sort_t = 0, tmp_sort_f = 0, tmp_scan = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort_t, static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr),
                                               static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), int(n), 0, 64,
                                               ukfb::main_stream(e)));
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort_f, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                               static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), int(n), 0, 32,
                                               ukfb::main_stream(e)));
    HIP_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, tmp_scan, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                              hipcub::Max(), int(n), ukfb::main_stream(e)));
    const size_t tmp_bytes = std::max(tmp_sort_t, std::max(tmp_sort_f, tmp_scan));
    Carver c(static_cast<char*>(e->ev_dev) + ws_offset);
    const size_t ne = size_t(n);
    uint32_t* idx_a = c.take<uint32_t>(ne);
    uint32_t* idx_b = c.take<uint32_t>(ne);
    uint32_t* idx_c = c.take<uint32_t>(ne);
    uint32_t* idx_d = c.take<uint32_t>(ne);
    int64_t* key_t_a = c.take<int64_t>(ne);
    int64_t* key_t_b = c.take<int64_t>(ne);
    uint32_t* key_f_a = c.take<uint32_t>(ne);
    uint32_t* key_f_b = c.take<uint32_t>(ne);
    uint32_t* head = c.take<uint32_t>(ne);
    uint32_t* start = c.take<uint32_t>(ne);
    uint32_t* rank = c.take<uint32_t>(ne);
    uint32_t* rank_sorted = c.take<uint32_t>(ne);
    uint32_t* off = c.take<uint32_t>(ne + 1);
    int32_t* fidx_c = c.take<int32_t>(ne);
    int64_t* ts_c = c.take<int64_t>(ne);
    int32_t* meas_c = c.take<int32_t>(ne);
    char* z_c = c.take<char>(3 * ne * e->tsize);
    char* Q_c = c.take<char>(9 * ne * e->tsize);
    uint32_t* flags = c.take<uint32_t>(2);
    char* tmp = c.take<char>(tmp_bytes);
    if (ws_offset + c.used > e->ev_bytes) return fail(UKFB_ERR_INVALID_ARG, "ukfb_process_events: workspace too small (internal)");

    // ---- ordering, all on the device: by timestamp, then (stable) by filter = per-filter time order; the position of
    // a sample inside its filter's run is its round; then (stable) by round = round-major, filter-minor
    const int blocks = int((n + 255) / 256);
    HIP_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(uint32_t), ukfb::main_stream(e)));
    hipLaunchKernelGGL(events_init_kernel, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), d_f, d_t, n, e->cap, idx_a, key_t_a, flags);
    size_t tb = tmp_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, key_t_a, key_t_b, idx_a, idx_b, int(n), 0, 64, ukfb::main_stream(e)));
    hipLaunchKernelGGL(events_filter_keys_kernel, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), d_f, idx_b, n, e->cap, key_f_a);
    tb = tmp_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, key_f_a, key_f_b, idx_b, idx_c, int(n), 0, bits_f, ukfb::main_stream(e)));
    hipLaunchKernelGGL(events_heads_kernel, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), key_f_b, n, head);
    tb = tmp_bytes;
    HIP_TRY(hipcub::DeviceScan::InclusiveScan(tmp, tb, head, start, hipcub::Max(), int(n), ukfb::main_stream(e)));
    hipLaunchKernelGGL(events_rank_kernel, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), start, idx_c, d_m, e->model, n, rank, flags);
    tb = tmp_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, rank, rank_sorted, idx_c, idx_d, int(n), 0, 32, ukfb::main_stream(e)));
    hipLaunchKernelGGL(events_round_offsets_kernel, dim3(blocks), dim3(256), 0, ukfb::main_stream(e), rank_sorted, n, off);
    if (e->prec == UKFB_F64)
        hipLaunchKernelGGL((events_gather_kernel<S, double>), dim3(blocks), dim3(256), 0, ukfb::main_stream(e), d_f, d_t, d_m, d_z, d_q, idx_d,
                           n, fidx_c, ts_c, meas_c, reinterpret_cast<double*>(z_c), reinterpret_cast<double*>(Q_c));
    else
        hipLaunchKernelGGL((events_gather_kernel<S, float>), dim3(blocks), dim3(256), 0, ukfb::main_stream(e), d_f, d_t, d_m, d_z, d_q, idx_d,
                           n, fidx_c, ts_c, meas_c, reinterpret_cast<float*>(z_c), reinterpret_cast<float*>(Q_c));
    HIP_TRY(hipGetLastError());
    // ---- the one host read of the call: input validity, number of rounds and where each round starts (launch
    // geometry has to be known on the host)
    uint32_t hflags[2] = {0, 0};
    int rc = download_raw(e, flags, hflags, 2);
    if (rc) return rc;
    if (hflags[0] & 1u) return fail(UKFB_ERR_OUT_OF_RANGE, "ukfb_process_events: filter index or timestamp out of range");
    if (hflags[0] & 2u) return fail(UKFB_ERR_OUT_OF_RANGE, "ukfb_process_events: more than 2^30 samples of one filter in one call");
    const int64_t nrounds = int64_t(hflags[1]) + 1;
    std::vector<uint32_t> hoff(size_t(nrounds) + 1);
    rc = download_raw(e, off, hoff.data(), hoff.size());
    if (rc) return rc;

    // ---- one indirect fused launch per round over exactly the filters that have a sample in it: the cost of the
    // call follows the number of events, not rounds x capacity.  Status words accumulate (OR) across rounds.
    HIP_TRY(hipMemsetAsync(e->status, 0, size_t(e->cap) * sizeof(uint32_t), ukfb::main_stream(e)));
    for (int64_t r = 0; r < nrounds; ++r) {
        const size_t o = hoff[size_t(r)], cnt = size_t(hoff[size_t(r) + 1]) - o;
        if (cnt == 0) continue;
        ukfb::LaunchReq q;
        q.do_predict = true;
        q.do_update = true;
        q.ts_dev = ts_c + o;
        q.meas_dev = meas_c + o;
        q.z_dev = z_c + 3 * o * e->tsize;
        q.Q_dev = Q_c + 9 * o * e->tsize;
        q.filter_index_dev = fidx_c + o;
        q.n_items = int64_t(cnt);
        q.status_accumulate = true;
        rc = launch(e, q);
        if (rc) return rc;
    }
    if (rounds) *rounds = nrounds;
    if (status_or) return ukfb_get_status_summary(e, status_or);
    return UKFB_OK;
}

// bytes of workspace process_events_device needs for n events (an upper bound that does not depend on hipCUB's
// temporary-storage query: 4x the key/value arrays covers rocPRIM's double buffers)
size_t events_workspace_bytes(int64_t n) {
    const size_t ne = size_t(n);
    // 4 index + 2 time-key + 2 filter-key + head/start/rank/rank_sorted/off + compact events (int32, int64, int32,
    // 12 scalars of <= 8 bytes) + alignment slack + radix-sort temporaries
    return (4 * 4 + 2 * 8 + 2 * 4 + 5 * 4 + 4 + 8 + 4 + 12 * 8) * ne + 32 * 256 + (size_t(64) << 20) / 4 + 24 * ne;
}

int ensure_events_arena(ukfb_engine* e, size_t bytes) {
    if (bytes <= e->ev_bytes) return UKFB_OK;
    if (e->ev_dev) HIP_TRY(hipFree(e->ev_dev));
    e->ev_dev = nullptr;
    e->ev_bytes = 0;
    HIP_TRY(hipMalloc(&e->ev_dev, bytes));
    e->ev_bytes = bytes;
    return UKFB_OK;
}
}  // namespace

extern "C" {

int ukfb_process_events(ukfb_engine* e, int64_t n_events, const int64_t* filter, const int64_t* ts_us,
                        const int32_t* meas_model, const double* z, const double* Q, uint32_t* status_or, int64_t* rounds) {
    if (!e || n_events < 0 || n_events > 0x7fffffff || (n_events > 0 && (!filter || !ts_us || !meas_model || !z || !Q)))
        return fail(UKFB_ERR_INVALID_ARG, "ukfb_process_events: bad argument");
    ON_DEVICE(e->device);
    if (status_or) *status_or = 0;
    if (rounds) *rounds = 0;
    if (n_events == 0) return UKFB_OK;
    // raw events go to the device as they are (one copy per array); ordering, conversion to the engine's
    // precision and the per-round scatter all happen there
    const size_t ne = size_t(n_events);
    Carver raw(nullptr);
    raw.take<int64_t>(ne); raw.take<int64_t>(ne); raw.take<int32_t>(ne); raw.take<double>(3 * ne); raw.take<double>(9 * ne);
    int rc = ensure_events_arena(e, raw.used + events_workspace_bytes(n_events));
    if (rc) return rc;
    Carver c(e->ev_dev);
    int64_t* d_f = c.take<int64_t>(ne);
    int64_t* d_t = c.take<int64_t>(ne);
    int32_t* d_m = c.take<int32_t>(ne);
    double* d_z = c.take<double>(3 * ne);
    double* d_q = c.take<double>(9 * ne);
    HIP_TRY(hipMemcpyAsync(d_f, filter, ne * sizeof(int64_t), hipMemcpyHostToDevice, ukfb::main_stream(e)));
    HIP_TRY(hipMemcpyAsync(d_t, ts_us, ne * sizeof(int64_t), hipMemcpyHostToDevice, ukfb::main_stream(e)));
    HIP_TRY(hipMemcpyAsync(d_m, meas_model, ne * sizeof(int32_t), hipMemcpyHostToDevice, ukfb::main_stream(e)));
    HIP_TRY(hipMemcpyAsync(d_z, z, 3 * ne * sizeof(double), hipMemcpyHostToDevice, ukfb::main_stream(e)));
    HIP_TRY(hipMemcpyAsync(d_q, Q, 9 * ne * sizeof(double), hipMemcpyHostToDevice, ukfb::main_stream(e)));
    rc = process_events_device<double>(e, n_events, d_f, d_t, d_m, d_z, d_q, c.used, status_or, rounds);
    ENGINE_SYNC(e);   // the caller's buffers are free again
    return rc;
}

int ukfb_process_events_dev(ukfb_engine* e, int64_t n_events, const int64_t* filter_dev, const int64_t* ts_us_dev,
                            const int32_t* meas_model_dev, const void* z_dev, const void* Q_dev, uint32_t* status_or,
                            int64_t* rounds) {
    if (!e || n_events < 0 || n_events > 0x7fffffff ||
        (n_events > 0 && (!filter_dev || !ts_us_dev || !meas_model_dev || !z_dev || !Q_dev)))
        return fail(UKFB_ERR_INVALID_ARG, "ukfb_process_events_dev: bad argument");
    ON_DEVICE(e->device);
    if (status_or) *status_or = 0;
    if (rounds) *rounds = 0;
    if (n_events == 0) return UKFB_OK;
    int rc = ensure_events_arena(e, events_workspace_bytes(n_events));
    if (rc) return rc;
    if (e->prec == UKFB_F64)
        return process_events_device<double>(e, n_events, filter_dev, ts_us_dev, meas_model_dev, static_cast<const double*>(z_dev),
                                             static_cast<const double*>(Q_dev), 0, status_or, rounds);
    return process_events_device<float>(e, n_events, filter_dev, ts_us_dev, meas_model_dev, static_cast<const float*>(z_dev),
                                        static_cast<const float*>(Q_dev), 0, status_or, rounds);
}

// ---- measurement of the engine ------------------------------------------------------------------
int ukfb_last_launch_info(const ukfb_engine* e, char* kernel_name, int name_capacity, int* lds_bytes,
                          int* filters_per_workgroup, int64_t* grid) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    if (kernel_name && name_capacity > 0) {
        std::snprintf(kernel_name, size_t(name_capacity), "%s", e->last_kernel.c_str());
    }
    if (lds_bytes) *lds_bytes = e->last_lds;
    if (filters_per_workgroup) *filters_per_workgroup = e->last_fpw;
    if (grid) *grid = e->last_grid;
    return UKFB_OK;
}

int ukfb_last_model_groups(ukfb_engine* e, int32_t* list, int64_t capacity, int64_t* items) {
    if (!e || !items || (capacity > 0 && !list)) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    *items = e->bucket_items;
    if (!e->bucket_idx || e->bucket_items == 0) return fail(UKFB_ERR_INVALID_ARG, "no launch of this engine has grouped its filters by update class");
    const int64_t n = capacity < e->bucket_items ? capacity : e->bucket_items;
    if (n > 0) {
        HIP_TRY(hipStreamSynchronize(ukfb::main_stream(e)));
        HIP_TRY(hipMemcpy(list, e->bucket_idx, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return UKFB_OK;
}

int ukfb_timer_begin(ukfb_engine* e) {
    if (!e) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    HIP_TRY(hipEventRecord(e->ev0, ukfb::main_stream(e)));
    return UKFB_OK;
}

int ukfb_timer_end(ukfb_engine* e, float* elapsed_ms) {
    if (!e || !elapsed_ms) return UKFB_ERR_INVALID_ARG;
    ON_DEVICE(e->device);
    if (e->poisoned) return fail(UKFB_ERR_HIP, "engine poisoned by an earlier wait that timed out (UKFB_WAIT_TIMEOUT_S)");
    HIP_TRY(hipEventRecord(e->ev1, ukfb::main_stream(e)));
    {
        const hipError_t w = wait_event_polling(e->ev1);
        if (w == hipErrorNotReady && g_wait_timed_out) {
            e->poisoned = true;
            g_wait_timed_out = false;
            return UKFB_ERR_HIP;
        }
        HIP_TRY(w);
    }
    HIP_TRY(hipEventElapsedTime(elapsed_ms, e->ev0, e->ev1));
    return UKFB_OK;
}

}  // extern "C"
