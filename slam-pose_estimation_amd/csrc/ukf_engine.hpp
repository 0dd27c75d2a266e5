// ukf_engine.hpp -- host-side state of one batched UKF engine and the launch request that the
// C-ABI (ukf_batch.hip) hands to the per-(precision, model) translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/ukf_batch.h"

#ifndef UKFB_MAX_MULTI_CYCLES
#define UKFB_MAX_MULTI_CYCLES 32   // cycles of one multi-cycle launch with a schedule (the host splits longer ones)
#endif

#ifndef UKFB_GENERIC_F64
#define UKFB_GENERIC_F64 0   // 1: also build the fp64 one-wavefront-per-filter kernels (AGPR-backed, diagnostics only)
#endif

struct ukfb_engine {
    int model = 0, prec = 0, device = 0;
    int64_t cap = 0;
    int S = 0, D = 0, PK = 0;
    size_t tsize = 8;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // Small batches run a launch as two halves on two streams, so that the tail of one half overlaps the head of the next
    // launch's other half (consecutive launches of one stream do not overlap).  stream_b carries the second half; ev_a / ev_b
    // order it against everything else, which stays on `stream` (ukfb::main_stream joins first).  Engines that own their stream only.
    hipStream_t stream_b = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    bool split_pending = false;   // a second half is in flight on stream_b that `stream` has not waited for yet
    bool poisoned = false;   // a bounded wait gave up on this engine's stream: every later call fails fast (see ukfb_sync)
    ukfb_config cfg{};

    // per-filter device state
    void* mu = nullptr;          // [cap][S]   engine precision
    void* cov = nullptr;         // [cap][PK]  packed lower triangle
    uint32_t* status = nullptr;  // [cap]
    uint8_t* init = nullptr;     // [cap]
    int64_t* last_ts = nullptr;  // [cap]

    // process noise: batch-uniform D*D, or per filter after ukfb_set_process_noise_per_filter
    void* Rn = nullptr;
    void* Racc = nullptr;  // Pose: acceleration-branch noise (Rn with the velocity block replaced)
    int64_t Racc_mats = 0; // matrices Racc has room for (1, or capacity once the noise is per filter)
    void* acc_cov_dev = nullptr;  // 9 doubles' worth of staging for rebuild_racc (persistent)
    bool Rn_per_filter = false;
    bool noise_psd = true;   // the batch-uniform process noise (and, Pose, its acceleration-branch form) is positive semidefinite
    bool noise_iso = true;   // rotated diagonal blocks of the (batch-uniform) process noise are s * I; the zero default is
    std::vector<double> Rn_host;  // uniform copy

    // latched inputs
    void* in_a = nullptr;  // Pose acc.mu / Orient acceleration.mu   [cap][3]
    void* in_b = nullptr;  // Orient rotation_rate.mu                [cap][3]
    const void* in_a_bound = nullptr;
    const void* in_b_bound = nullptr;
    double acc_cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // Measurement.hpp:12 (Identity)
    double tau_g = 1.0, tau_a = 1.0, earth[3] = {0, 0, 0};

    // staging for host-pointer entry points
    void* z_stage = nullptr;        // [cap][3]
    void* Q_stage = nullptr;        // [cap][9]
    int32_t* meas_stage = nullptr;  // [cap]
    uint8_t* active_stage = nullptr;
    double* dt_stage = nullptr;
    int64_t* ts_stage = nullptr;
    uint32_t* reduce_word = nullptr;  // status OR-reduction target
    // Host-fed fused cycles (ukfb_cycle, ukfb_cycle_uniform_q): the samples of call k + 1 are uploaded on a COPY stream into
    // the other of two staging sets while the kernel of call k still runs on the engine's stream (created at the first such
    // call).  ev_copy[s]: set s is uploaded; ev_used[s]: the kernel that read set s has finished.
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
    void* zc_stage[2] = {nullptr, nullptr};   // [cap][3]
    void* Qc_stage[2] = {nullptr, nullptr};   // [cap][9]
    bool stage_busy[2] = {false, false};
    int stage_slot = 0;
    void* cvt_copy = nullptr;                 // fp32 engines: scratch of the copy stream (host doubles narrowed on the device)
    size_t cvt_copy_bytes = 0;
    // fp32 engines: device scratch for host doubles that are narrowed on the device (grow-only)
    void* cvt_dev = nullptr;
    size_t cvt_bytes = 0;
    // ukfb_cycle_multi: device rings of the host samples of one call (grow-only)
    void* multi_dev = nullptr;
    size_t multi_bytes = 0;
    // model-class buckets of ukfb_cycle_dev with per-filter model ids: filter list ordered by class, padded to whole
    // wavefronts [cap + 16], and the per-block class counts [3][blocks]
    int32_t* bucket_idx = nullptr;
    int64_t bucket_items = 0;    // entries of bucket_idx the last grouped launch covered (0: none yet)
    uint32_t* bucket_counts = nullptr;
    // ukfb_process_events: device workspace (grow-only)
    void* ev_dev = nullptr;
    size_t ev_bytes = 0;

    // last launch (for bench.py / profiles)
    std::string last_kernel;
    int last_lds = 0, last_fpw = 0;
    int64_t last_grid = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace ukfb {

// Untyped launch request; the typed KArgs<T> is built inside the (T, M) translation unit.
struct LaunchReq {
    bool do_predict = false, do_update = false;
    double dt_uniform = 0.0;
    const double* dt_dev = nullptr;
    const int64_t* ts_dev = nullptr;
    int meas_uniform = -1;
    const int32_t* meas_dev = nullptr;
    const void* z_dev = nullptr;
    const void* Q_dev = nullptr;
    bool q_uniform = false;       // Q_dev holds ONE 3x3 (9 scalars) for the whole batch
    const uint8_t* active_dev = nullptr;
    // indirect launch over a list of filters (event rounds): n_items entries of filter_index_dev; -1: every filter
    const int32_t* filter_index_dev = nullptr;
    int64_t n_items = -1;
    bool inputs_by_filter = false;   // the list also indexes the per-call inputs; negative entries are padding (buckets)
    bool status_accumulate = false;
    // multi-cycle launch (ukfb_cycle_multi_dev): cycles > 0 selects it; in_a_slots / in_b_slots replace the latched inputs
    int cycles = 0, first_slot = 0, slots = 1;
    const void* in_a_slots = nullptr;
    const void* in_b_slots = nullptr;
    const double* sched_dt = nullptr;        // host, [cycles]: per-cycle time steps and models (both or neither)
    const int32_t* sched_model = nullptr;
    // host-fed cycles: the launch waits for `wait_event` (inputs uploaded on the copy stream) and records `done_event`
    // behind the kernel; no_split keeps it one kernel on the engine's stream (the event then covers all of it)
    hipEvent_t wait_event = nullptr, done_event = nullptr;
    bool no_split = false;
};

int launch_pose_f64(ukfb_engine* e, const LaunchReq& r);
int launch_pose_f32(ukfb_engine* e, const LaunchReq& r);
int launch_orient_f64(ukfb_engine* e, const LaunchReq& r);
int launch_orient_f32(ukfb_engine* e, const LaunchReq& r);
int launch_pose_f32w(ukfb_engine* e, const LaunchReq& r);     // fp32 engines, wide arithmetic (ukfb_config::wide_arithmetic)
int launch_orient_f32w(ukfb_engine* e, const LaunchReq& r);

void set_error(const char* what, hipError_t err);
void set_error_text(const std::string& text);   // what ukfb_last_error() returns on this thread

// Every entry point works on its engine's device and hands the calling thread back on the device it came with: a host that
// drives several engines from one thread (ukfb_group_*), or shares the thread with another HIP user, keeps its own notion of
// "current device".  Nothing is switched (one thread-local read) when the caller already is on the engine's device.
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev) {
        int cur = -1;
        err = hipGetDevice(&cur);
        if (err == hipSuccess && cur != dev) {
            err = hipSetDevice(dev);
            if (err == hipSuccess) prev = cur;
        }
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// The engine's stream for anything but a split launch: first makes it wait for the second half of the last split launch.
inline hipStream_t main_stream(ukfb_engine* e) {
    if (e->split_pending) {
        (void)hipEventRecord(e->ev_b, e->stream_b);
        (void)hipStreamWaitEvent(e->stream, e->ev_b, 0);
        e->split_pending = false;
    }
    return e->stream;
}

}  // namespace ukfb
