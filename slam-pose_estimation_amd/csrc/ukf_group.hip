// ukf_group.hip -- device groups: ONE host process drives the engines of several MI355X (include/ukf_batch.h, ukfb_group_*).
//
// north_star's multi-GPU shape for a C++ host: the filters of a batch are independent (every filter of the reference
// owns its own ukf object, src/UnscentedKalmanFilter.hpp:150), so a batch splits into contiguous shards, one engine and
// one stream per device, with NO collective on the data path.  The only exchange is the result gather, an RCCL all-gather
// of the mean states over xGMI (ncclCommInitAll + ncclAllGather, /opt/rocm/include/rccl/rccl.h:236,678).
//
// RCCL is bound at run time (dlopen of librccl.so.1 at the first gather): libukf_batch.so carries no link-time
// dependency on it, and a process that already holds an RCCL (PyTorch ships one under the same soname) shares that copy
// instead of loading a second one.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "ukf_engine.hpp"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString; }
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // a copy that is already in the process first (same soname), then the system's
        x.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!x.lib) x.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!x.lib) x.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!x.lib) x.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (x.lib) {
            x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.lib, "ncclCommInitAll"));
            x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.lib, "ncclCommDestroy"));
            x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.lib, "ncclAllGather"));
            x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.lib, "ncclGroupStart"));
            x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.lib, "ncclGroupEnd"));
            x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.lib, "ncclGetErrorString"));
        }
        return x;
    }();
    return r;
}

}  // namespace

struct ukfb_group {
    int model = 0, prec = 0, S = 0, D = 0;
    size_t tsize = 8;
    int64_t total = 0;
    std::vector<int> devices;
    std::vector<ukfb_engine*> engines;
    std::vector<int64_t> first, count;
    // result gather: one communicator per shard (created at the first gather), padded staging per shard
    std::vector<ncclComm_t> comms;
    std::vector<void*> send_pad, recv_pad;
    std::vector<hipEvent_t> staged, pulled;   // copy exchange (shards sharing a device): shard s staged / shard r pulled every block
    int last_gather_exchange = 0;             // 0 none yet, 1 RCCL all-gather, 2 peer copies
    int64_t max_count = 0;
    // event routing (ukfb_group_process_events): one pinned, grow-only buffer per shard
    std::vector<void*> route_buf;
    std::vector<size_t> route_cap;
};

namespace {

int gfail(int code, const std::string& msg) {
    ukfb::set_error_text(msg);   // one error channel: ukfb_last_error()
    return code;
}

// rc of an engine call: the engine has left its text in ukfb_last_error() already
int efail(int rc) { return rc; }

bool shard_ok(const ukfb_group* g, int r) { return g && r >= 0 && r < int(g->engines.size()); }

// [first, first + count) of the batch cut by shard r: (offset inside the caller's arrays, offset inside the shard, length)
struct Cut {
    int64_t src, dst, len;
};
Cut cut(const ukfb_group* g, int r, int64_t first, int64_t count) {
    const int64_t lo = std::max(first, g->first[r]), hi = std::min(first + count, g->first[r] + g->count[r]);
    return {lo - first, lo - g->first[r], std::max<int64_t>(0, hi - lo)};
}

void release_gather(ukfb_group* g) {
    for (size_t r = 0; r < g->comms.size(); ++r)
        if (g->comms[r]) (void)rccl().CommDestroy(g->comms[r]);
    g->comms.clear();
    for (size_t r = 0; r < g->send_pad.size(); ++r) {
        ukfb::DeviceScope on_device(g->devices[r]);
        if (g->send_pad[r]) (void)hipFree(g->send_pad[r]);
        if (g->recv_pad[r]) (void)hipFree(g->recv_pad[r]);
        if (r < g->staged.size() && g->staged[r]) (void)hipEventDestroy(g->staged[r]);
        if (r < g->pulled.size() && g->pulled[r]) (void)hipEventDestroy(g->pulled[r]);
    }
    g->send_pad.clear();
    g->recv_pad.clear();
    g->staged.clear();
    g->pulled.clear();
}

// Host-array calls over a large batch: every shard's call stages and uploads its range, which occupies the calling thread
// for the duration of the copy -- one thread per shard keeps the PCIe links of all devices busy at once.  Small batches and
// the device-pointer calls (which only enqueue) stay on the calling thread: a thread costs more than they do.
constexpr int64_t THREAD_MIN_FILTERS = 32768;
template <class F> int fan_out(ukfb_group* g, int64_t filters_touched, F&& call) {   // call(shard) -> rc, text in ukfb_last_error()
    const size_t n = g->engines.size();
    if (n == 1 || filters_touched < THREAD_MIN_FILTERS) {
        for (size_t r = 0; r < n; ++r) {
            const int rc = call(r);
            if (rc) return rc;
        }
        return UKFB_OK;
    }
    std::vector<int> rcs(n, UKFB_OK);
    std::vector<std::string> errs(n);
    const auto run = [&](size_t r) {
        rcs[r] = call(r);
        if (rcs[r]) errs[r] = ukfb_last_error();   // (the text is per thread)
    };
    std::vector<std::thread> workers;
    for (size_t r = 1; r < n; ++r) workers.emplace_back(run, r);
    run(0);
    for (std::thread& w : workers) w.join();
    for (size_t r = 0; r < n; ++r)
        if (rcs[r]) {
            ukfb::set_error_text(errs[r]);
            return rcs[r];
        }
    return UKFB_OK;
}

void release_routing(ukfb_group* g) {
    for (size_t r = 0; r < g->route_buf.size(); ++r)
        if (g->route_buf[r]) (void)hipHostFree(g->route_buf[r]);
    g->route_buf.clear();
    g->route_cap.clear();
}

}  // namespace

extern "C" {

int ukfb_group_shard_range(int64_t total, int n_shards, int shard, int64_t* first, int64_t* count) {
    if (total < 0 || n_shards <= 0 || shard < 0 || shard >= n_shards) return UKFB_ERR_INVALID_ARG;
    const int64_t base = total / n_shards, extra = total % n_shards;
    if (count) *count = base + (shard < extra ? 1 : 0);
    if (first) *first = shard * base + std::min<int64_t>(shard, extra);
    return UKFB_OK;
}

int ukfb_group_create(ukfb_group** out, int model, int precision, int64_t total_filters, const int* devices, int n_devices) {
    if (!out || !devices || n_devices <= 0 || total_filters < n_devices)
        return gfail(UKFB_ERR_INVALID_ARG, "ukfb_group_create: need devices and at least one filter per shard");
    *out = nullptr;
    ukfb_group* g = new (std::nothrow) ukfb_group();
    if (!g) return gfail(UKFB_ERR_INVALID_ARG, "out of host memory");
    g->model = model;
    g->prec = precision;
    g->total = total_filters;
    g->tsize = precision == UKFB_F64 ? 8 : 4;
    for (int r = 0; r < n_devices; ++r) {
        int64_t f = 0, c = 0;
        ukfb_group_shard_range(total_filters, n_devices, r, &f, &c);
        ukfb_engine* e = nullptr;
        const int rc = ukfb_create(&e, model, precision, c, devices[r], nullptr);   // its own stream on its own device
        if (rc) {
            const std::string msg = ukfb_last_error();
            ukfb_group_destroy(g);
            ukfb::set_error_text(msg);
            return rc;
        }
        g->devices.push_back(devices[r]);
        g->engines.push_back(e);
        g->first.push_back(f);
        g->count.push_back(c);
        g->max_count = std::max(g->max_count, c);
    }
    ukfb_describe(g->engines[0], nullptr, nullptr, nullptr, &g->S, &g->D, nullptr);
    *out = g;
    return UKFB_OK;
}

int ukfb_group_destroy(ukfb_group* g) {
    if (!g) return UKFB_OK;
    release_gather(g);
    int rc = UKFB_OK;
    // (the routing buffers feed asynchronous copies: every shard's stream drains in ukfb_destroy below before its memory
    // goes, but the pinned buffers must outlive those copies -- ukfb_group_process_events returns only after they completed)
    release_routing(g);
    for (ukfb_engine* e : g->engines) {
        const int r = ukfb_destroy(e);
        rc = rc ? rc : r;
    }
    delete g;
    return rc;
}

int ukfb_group_size(const ukfb_group* g) { return g ? int(g->engines.size()) : -1; }

int ukfb_group_shard(ukfb_group* g, int shard, ukfb_engine** engine, int* device, int64_t* first, int64_t* count) {
    if (!shard_ok(g, shard)) return gfail(UKFB_ERR_INVALID_ARG, "ukfb_group_shard: no such shard");
    if (engine) *engine = g->engines[shard];
    if (device) *device = g->devices[shard];
    if (first) *first = g->first[shard];
    if (count) *count = g->count[shard];
    return UKFB_OK;
}

int ukfb_group_set_config(ukfb_group* g, const ukfb_config* cfg) {
    if (!g || !cfg) return UKFB_ERR_INVALID_ARG;
    for (ukfb_engine* e : g->engines) {
        const int rc = ukfb_set_config(e, cfg);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_initialize(ukfb_group* g, int64_t first, int64_t count, const double* mu, const double* cov) {
    if (!g || !mu || !cov || first < 0 || count < 0 || first + count > g->total)
        return gfail(UKFB_ERR_OUT_OF_RANGE, "ukfb_group_initialize: bad range");
    const size_t DD = size_t(g->D) * g->D;
    return fan_out(g, count, [&](size_t r) {
        const Cut c = cut(g, int(r), first, count);
        return c.len ? ukfb_initialize(g->engines[r], c.dst, c.len, mu + size_t(c.src) * g->S, cov + size_t(c.src) * DD) : int(UKFB_OK);
    });
}

int ukfb_group_get_state(ukfb_group* g, int64_t first, int64_t count, double* mu, double* cov, uint8_t* initialised) {
    if (!g || first < 0 || count < 0 || first + count > g->total)
        return gfail(UKFB_ERR_OUT_OF_RANGE, "ukfb_group_get_state: bad range");
    const size_t DD = size_t(g->D) * g->D;
    return fan_out(g, count, [&](size_t r) {
        const Cut c = cut(g, int(r), first, count);
        return c.len ? ukfb_get_state(g->engines[r], c.dst, c.len, mu ? mu + size_t(c.src) * g->S : nullptr,
                                      cov ? cov + size_t(c.src) * DD : nullptr, initialised ? initialised + c.src : nullptr)
                     : int(UKFB_OK);
    });
}

int ukfb_group_get_status(ukfb_group* g, int64_t first, int64_t count, uint32_t* status) {
    if (!g || !status || first < 0 || count < 0 || first + count > g->total)
        return gfail(UKFB_ERR_OUT_OF_RANGE, "ukfb_group_get_status: bad range");
    for (size_t r = 0; r < g->engines.size(); ++r) {
        const Cut c = cut(g, int(r), first, count);
        if (!c.len) continue;
        const int rc = ukfb_get_status(g->engines[r], c.dst, c.len, status + c.src);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_get_status_summary(ukfb_group* g, uint32_t* or_of_all) {
    if (!g || !or_of_all) return UKFB_ERR_INVALID_ARG;
    uint32_t all = 0;
    for (ukfb_engine* e : g->engines) {
        uint32_t v = 0;
        const int rc = ukfb_get_status_summary(e, &v);
        if (rc) return efail(rc);
        all |= v;
    }
    *or_of_all = all;
    return UKFB_OK;
}

int ukfb_group_set_process_noise(ukfb_group* g, const double* R) {
    if (!g || !R) return UKFB_ERR_INVALID_ARG;
    for (ukfb_engine* e : g->engines) {
        const int rc = ukfb_set_process_noise(e, R);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_pose_set_acceleration(ukfb_group* g, int64_t first, int64_t count, const double* acc_mu, const double* acc_cov) {
    if (!g || first < 0 || count < 0 || first + count > g->total)
        return gfail(UKFB_ERR_OUT_OF_RANGE, "ukfb_group_pose_set_acceleration: bad range");
    for (size_t r = 0; r < g->engines.size(); ++r) {
        const Cut c = cut(g, int(r), first, count);
        // (the batch-uniform covariance goes to every shard, also to one the range does not touch)
        const int rc = ukfb_pose_set_acceleration(g->engines[r], c.len ? c.dst : 0, c.len, (acc_mu && c.len) ? acc_mu + size_t(c.src) * 3 : nullptr,
                                                  acc_cov);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_orient_set_params(ukfb_group* g, double gyro_bias_tau, double acc_bias_tau, const double earth_rotation[3]) {
    if (!g) return UKFB_ERR_INVALID_ARG;
    for (ukfb_engine* e : g->engines) {
        const int rc = ukfb_orient_set_params(e, gyro_bias_tau, acc_bias_tau, earth_rotation);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_orient_set_inputs(ukfb_group* g, int64_t first, int64_t count, const double* gyro, const double* acc) {
    if (!g || first < 0 || count < 0 || first + count > g->total)
        return gfail(UKFB_ERR_OUT_OF_RANGE, "ukfb_group_orient_set_inputs: bad range");
    for (size_t r = 0; r < g->engines.size(); ++r) {
        const Cut c = cut(g, int(r), first, count);
        if (!c.len) continue;
        const int rc = ukfb_orient_set_inputs(g->engines[r], c.dst, c.len, gyro ? gyro + size_t(c.src) * 3 : nullptr,
                                              acc ? acc + size_t(c.src) * 3 : nullptr);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

// ---- the hot path, fanned out: every call below only ENQUEUES on the shards' streams (one after the other from this
// thread, the devices then run concurrently) and returns; ukfb_group_sync waits for all of them
int ukfb_group_cycle_dev(ukfb_group* g, double dt, int meas_model, const void* const* z_dev, const void* const* Q_dev) {
    if (!g || !z_dev || !Q_dev) return UKFB_ERR_INVALID_ARG;
    for (size_t r = 0; r < g->engines.size(); ++r) {
        const int rc = ukfb_cycle_dev(g->engines[r], dt, meas_model, nullptr, z_dev[r], Q_dev[r]);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_cycle_multi_dev(ukfb_group* g, int cycles, double dt, int meas_model, int slots, int first_slot,
                               const void* const* in_a_dev, const void* const* in_b_dev, const void* const* z_dev,
                               const void* const* Q_dev) {
    if (!g || !z_dev || !Q_dev) return UKFB_ERR_INVALID_ARG;
    for (size_t r = 0; r < g->engines.size(); ++r) {
        const int rc = ukfb_cycle_multi_dev(g->engines[r], cycles, dt, meas_model, slots, first_slot, in_a_dev ? in_a_dev[r] : nullptr,
                                            in_b_dev ? in_b_dev[r] : nullptr, z_dev[r], Q_dev[r]);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_pose_bind_acceleration_dev(ukfb_group* g, const void* const* acc_mu_dev) {
    if (!g) return UKFB_ERR_INVALID_ARG;
    for (size_t r = 0; r < g->engines.size(); ++r) {
        const int rc = ukfb_pose_bind_acceleration_dev(g->engines[r], acc_mu_dev ? acc_mu_dev[r] : nullptr);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_orient_bind_inputs_dev(ukfb_group* g, const void* const* gyro_dev, const void* const* acc_dev) {
    if (!g) return UKFB_ERR_INVALID_ARG;
    for (size_t r = 0; r < g->engines.size(); ++r) {
        const int rc = ukfb_orient_bind_inputs_dev(g->engines[r], gyro_dev ? gyro_dev[r] : nullptr, acc_dev ? acc_dev[r] : nullptr);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

// host arrays over the whole batch (z [total][3], Q [total][3][3]): every shard uploads and launches its range
int ukfb_group_cycle(ukfb_group* g, double dt, int meas_model, const double* z, const double* Q) {
    if (!g || !z || !Q) return UKFB_ERR_INVALID_ARG;
    return fan_out(g, g->total, [&](size_t r) {
        return ukfb_cycle(g->engines[r], dt, meas_model, z + size_t(g->first[r]) * 3, Q + size_t(g->first[r]) * 9);
    });
}

int ukfb_group_predict(ukfb_group* g, double dt) {
    if (!g) return UKFB_ERR_INVALID_ARG;
    for (ukfb_engine* e : g->engines) {
        const int rc = ukfb_predict(e, dt);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_update(ukfb_group* g, int meas_model, const double* z, const double* Q) {
    if (!g || !z || !Q) return UKFB_ERR_INVALID_ARG;
    return fan_out(g, g->total, [&](size_t r) {
        return ukfb_update(g->engines[r], meas_model, z + size_t(g->first[r]) * 3, Q + size_t(g->first[r]) * 9, nullptr);
    });
}

// per-filter model ids on the devices (the mixed asynchronous stream of BASELINE config 5, one pointer per shard)
int ukfb_group_cycle_mixed_dev(ukfb_group* g, double dt, const int32_t* const* meas_model_dev, const void* const* z_dev,
                               const void* const* Q_dev) {
    if (!g || !meas_model_dev || !z_dev || !Q_dev) return UKFB_ERR_INVALID_ARG;
    for (size_t r = 0; r < g->engines.size(); ++r) {
        const int rc = ukfb_cycle_dev(g->engines[r], dt, 0, meas_model_dev[r], z_dev[r], Q_dev[r]);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

// fused predictionStepFromSampleTime + integrateMeasurement per filter, host arrays over the whole batch
int ukfb_group_cycle_timestamps(ukfb_group* g, const int64_t* ts_us, const int32_t* meas_model, const double* z, const double* Q) {
    if (!g || !ts_us || !meas_model) return UKFB_ERR_INVALID_ARG;
    return fan_out(g, g->total, [&](size_t r) {
        const size_t o = size_t(g->first[r]);
        return ukfb_cycle_timestamps(g->engines[r], ts_us + o, meas_model + o, z ? z + o * 3 : nullptr, Q ? Q + o * 9 : nullptr);
    });
}

// The time-ordered asynchronous stream over a sharded batch: events are routed to the shard that owns their filter (a
// stable partition: per filter the arrival order survives, which is all the ordering the stream has), every shard then
// orders and applies its own events.  The shards run concurrently, one host thread each -- a shard's call returns only
// when the host knows its number of rounds.
int ukfb_group_process_events(ukfb_group* g, int64_t n_events, const int64_t* filter, const int64_t* ts_us, const int32_t* meas_model,
                              const double* z, const double* Q, uint32_t* status_or, int64_t* rounds) {
    if (!g || n_events < 0 || n_events > 0x7fffffff || (n_events > 0 && (!filter || !ts_us || !meas_model || !z || !Q)))
        return gfail(UKFB_ERR_INVALID_ARG, "ukfb_group_process_events: bad argument");
    if (status_or) *status_or = 0;
    if (rounds) *rounds = 0;
    if (n_events == 0) return UKFB_OK;
    const size_t n = g->engines.size();
    struct Part {
        uint32_t st = 0;
        int64_t rounds = 0;
        int rc = UKFB_OK;
        std::string err;
    };
    if (n > 255) return gfail(UKFB_ERR_INVALID_ARG, "ukfb_group_process_events: more than 255 shards");
    std::vector<Part> parts(n);
    // pass 1 (this thread): the owner of every event and the shard sizes; pass 2 (the shard's thread): every shard collects
    // its own events, in arrival order, into its pinned routing buffer and hands them to its engine
    std::vector<uint8_t> owner(static_cast<size_t>(n_events));
    std::vector<size_t> counts(n, 0);
    {
        const double per_filter = double(n) / double(g->total);   // (a multiply, not a 64-bit division per event)
        for (int64_t i = 0; i < n_events; ++i) {
            const int64_t f = filter[i];
            if (f < 0 || f >= g->total) return gfail(UKFB_ERR_OUT_OF_RANGE, "ukfb_group_process_events: filter index outside the batch");
            size_t r = std::min(n - 1, size_t(double(f) * per_filter));   // shards differ by one filter at most: the guess is off by one at most
            while (f < g->first[r]) --r;
            while (f >= g->first[r] + g->count[r]) ++r;
            owner[size_t(i)] = uint8_t(r);
            ++counts[r];
        }
    }
    if (g->route_buf.empty()) {
        g->route_buf.assign(n, nullptr);
        g->route_cap.assign(n, 0);
    }
    constexpr size_t EVENT_BYTES = 2 * sizeof(int64_t) + sizeof(int32_t) + 12 * sizeof(double);
    const auto run = [&](size_t r) {
        Part& p = parts[r];
        ukfb_engine* e = g->engines[r];
        const size_t m = counts[r];
        ukfb::DeviceScope on_device(e->device);
        if (on_device.err != hipSuccess) {
            p.rc = UKFB_ERR_HIP;
            p.err = "ukfb_group_process_events: hipSetDevice";
            return;
        }
        if (m == 0) {
            // no sample for this shard: what the call does to a filter without samples -- a fresh, empty status word
            if (hipMemsetAsync(e->status, 0, size_t(e->cap) * sizeof(uint32_t), ukfb::main_stream(e)) != hipSuccess) {
                p.rc = UKFB_ERR_HIP;
                p.err = "ukfb_group_process_events: status reset failed";
            }
            return;
        }
        const int64_t f0 = g->first[r];
        if (m == size_t(n_events)) {   // every event belongs to this shard: the caller's arrays as they are (indices shifted if need be)
            std::vector<int64_t> shifted;
            if (f0 != 0) {
                shifted.resize(m);
                for (size_t k = 0; k < m; ++k) shifted[k] = filter[k] - f0;
            }
            p.rc = ukfb_process_events(e, n_events, f0 ? shifted.data() : filter, ts_us, meas_model, z, Q, &p.st, &p.rounds);
            if (p.rc) p.err = ukfb_last_error();
            return;
        }
        const size_t need = m * EVENT_BYTES + 64;
        if (g->route_cap[r] < need) {
            if (g->route_buf[r]) (void)hipHostFree(g->route_buf[r]);
            g->route_buf[r] = nullptr;
            g->route_cap[r] = 0;
            const size_t want = need + need / 4;
            if (hipHostMalloc(&g->route_buf[r], want, hipHostMallocDefault) != hipSuccess) {
                p.rc = UKFB_ERR_HIP;
                p.err = "ukfb_group_process_events: pinned routing buffer";
                return;
            }
            g->route_cap[r] = want;
        }
        // doubles first (8-byte aligned), then the 64-bit and 32-bit integers
        double* pz = static_cast<double*>(g->route_buf[r]);
        double* pq = pz + 3 * m;
        int64_t* pf = reinterpret_cast<int64_t*>(pq + 9 * m);
        int64_t* pt = pf + m;
        int32_t* pm = reinterpret_cast<int32_t*>(pt + m);
        size_t k = 0;
        for (int64_t i = 0; i < n_events; ++i) {
            if (owner[size_t(i)] != r) continue;
            pf[k] = filter[i] - f0;
            pt[k] = ts_us[i];
            pm[k] = meas_model[i];
            std::memcpy(pz + 3 * k, z + size_t(i) * 3, 3 * sizeof(double));
            std::memcpy(pq + 9 * k, Q + size_t(i) * 9, 9 * sizeof(double));
            ++k;
        }
        p.rc = ukfb_process_events(e, int64_t(m), pf, pt, pm, pz, pq, &p.st, &p.rounds);   // (returns after its copies completed)
        if (p.rc) p.err = ukfb_last_error();   // (the text is per thread)
    };
    size_t busy = 0;
    for (size_t r = 0; r < n; ++r) busy += counts[r] ? 1 : 0;
    if (busy <= 1) {
        for (size_t r = 0; r < n; ++r) run(r);
    } else {
        std::vector<std::thread> workers;
        for (size_t r = 1; r < n; ++r) workers.emplace_back(run, r);
        run(0);
        for (std::thread& w : workers) w.join();
    }
    uint32_t st = 0;
    int64_t rd = 0;
    for (const Part& p : parts) {
        if (p.rc) return gfail(p.rc, p.err);
        st |= p.st;
        rd = std::max(rd, p.rounds);
    }
    if (status_or) *status_or = st;
    if (rounds) *rounds = rd;   // launches of the shard that needed most
    return UKFB_OK;
}

int ukfb_group_sync(ukfb_group* g) {
    if (!g) return UKFB_ERR_INVALID_ARG;
    int rc = UKFB_OK;
    for (ukfb_engine* e : g->engines) {
        const int r = ukfb_sync(e);
        if (r && !rc) rc = efail(r);
    }
    return rc;
}

// HIP-event timing over all shards: begin / end bracket a region on every shard's stream, elapsed = the slowest shard
int ukfb_group_timer_begin(ukfb_group* g) {
    if (!g) return UKFB_ERR_INVALID_ARG;
    for (ukfb_engine* e : g->engines) {
        const int rc = ukfb_timer_begin(e);
        if (rc) return efail(rc);
    }
    return UKFB_OK;
}

int ukfb_group_timer_end(ukfb_group* g, float* elapsed_ms_max, float* elapsed_ms_per_shard) {
    if (!g || !elapsed_ms_max) return UKFB_ERR_INVALID_ARG;
    float mx = 0.f;
    for (size_t r = 0; r < g->engines.size(); ++r) {
        float ms = 0.f;
        const int rc = ukfb_timer_end(g->engines[r], &ms);
        if (rc) return efail(rc);
        if (elapsed_ms_per_shard) elapsed_ms_per_shard[r] = ms;
        mx = std::max(mx, ms);
    }
    *elapsed_ms_max = mx;
    return UKFB_OK;
}

// ---- result gather.  out_dev[r] (device r, engine precision, [total][S]) receives the means of ALL filters in batch order.
// Shards may differ by one filter (ukfb_group_shard_range), so the exchange runs on buffers padded to the largest shard:
//   1. stage    every shard copies its means into its padded send buffer           (its own stream)
//   2. exchange recv_pad[r] = [send_pad[0] | send_pad[1] | ...] on every device    (RCCL all-gather over xGMI; shards that
//               share a device -- where RCCL has no rank to give them -- exchange by peer copies instead)
//   3. compact  one copy per shard takes the padding out: out[first[s] ...] = recv_pad[r][s]   (ragged shards)
// Steps 1 and 3 are the same code for both exchanges, so a one-GPU box exercises them with N > 1 shards
// (tests/test_gpu_group.py: 2, 3 and 8 shards on device 0, ragged totals).
namespace {
int gather_stage(ukfb_group* g, std::vector<hipStream_t>& streams) {
    const size_t n = g->engines.size(), row = size_t(g->S) * g->tsize;
    for (size_t r = 0; r < n; ++r) {
        ukfb_engine* e = g->engines[r];
        ukfb::DeviceScope on_device(g->devices[r]);
        if (on_device.err != hipSuccess) return gfail(UKFB_ERR_HIP, "hipSetDevice");
        streams[r] = ukfb::main_stream(e);
        if (hipMemcpyAsync(g->send_pad[r], e->mu, size_t(g->count[r]) * row, hipMemcpyDeviceToDevice, streams[r]) != hipSuccess)
            return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: staging copy failed");
    }
    return UKFB_OK;
}

int gather_exchange_rccl(ukfb_group* g, const std::vector<hipStream_t>& streams) {
    Rccl& rc = rccl();
    const size_t n = g->engines.size();
    const ncclDataType_t dt = g->prec == UKFB_F64 ? ncclFloat64 : ncclFloat32;
    ncclResult_t st = rc.GroupStart();
    for (size_t r = 0; r < n && st == ncclSuccess; ++r)
        st = rc.AllGather(g->send_pad[r], g->recv_pad[r], size_t(g->max_count) * g->S, dt, g->comms[r], streams[r]);
    const ncclResult_t st2 = rc.GroupEnd();
    if (st != ncclSuccess || st2 != ncclSuccess)
        return gfail(UKFB_ERR_HIP, std::string("ncclAllGather: ") + rc.GetErrorString(st != ncclSuccess ? st : st2));
    return UKFB_OK;
}

// Shards on one device (or any device list RCCL cannot take): stream r pulls every shard's padded block once that shard's
// staging copy has finished (one event per shard).
int gather_exchange_copies(ukfb_group* g, const std::vector<hipStream_t>& streams) {
    const size_t n = g->engines.size(), pad_bytes = size_t(g->max_count) * size_t(g->S) * g->tsize;
    for (size_t s = 0; s < n; ++s) {
        ukfb::DeviceScope on_device(g->devices[s]);
        if (on_device.err != hipSuccess || hipEventRecord(g->staged[s], streams[s]) != hipSuccess)
            return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: event record failed");
    }
    for (size_t r = 0; r < n; ++r) {
        ukfb::DeviceScope on_device(g->devices[r]);
        if (on_device.err != hipSuccess) return gfail(UKFB_ERR_HIP, "hipSetDevice");
        for (size_t s = 0; s < n; ++s) {
            if (s != r && hipStreamWaitEvent(streams[r], g->staged[s], 0) != hipSuccess)
                return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: stream wait failed");
            if (hipMemcpyPeerAsync(static_cast<char*>(g->recv_pad[r]) + s * pad_bytes, g->devices[r], g->send_pad[s], g->devices[s],
                                   pad_bytes, streams[r]) != hipSuccess)
                return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: exchange copy failed");
        }
    }
    // (a later staging copy of shard s must not overwrite send_pad[s] while another stream still reads it)
    for (size_t r = 0; r < n; ++r) {
        ukfb::DeviceScope on_device(g->devices[r]);
        if (on_device.err != hipSuccess || hipEventRecord(g->pulled[r], streams[r]) != hipSuccess)
            return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: event record failed");
    }
    for (size_t s = 0; s < n; ++s)
        for (size_t r = 0; r < n; ++r)
            if (r != s && hipStreamWaitEvent(streams[s], g->pulled[r], 0) != hipSuccess)
                return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: stream wait failed");
    return UKFB_OK;
}

int gather_compact(ukfb_group* g, const std::vector<hipStream_t>& streams, void* const* out_dev) {
    const size_t n = g->engines.size(), row = size_t(g->S) * g->tsize, pad_bytes = size_t(g->max_count) * row;
    for (size_t r = 0; r < n; ++r) {
        ukfb::DeviceScope on_device(g->devices[r]);
        if (on_device.err != hipSuccess) return gfail(UKFB_ERR_HIP, "hipSetDevice");
        for (size_t s = 0; s < n; ++s)
            if (g->count[s] > 0 &&
                hipMemcpyAsync(static_cast<char*>(out_dev[r]) + size_t(g->first[s]) * row,
                               static_cast<const char*>(g->recv_pad[r]) + s * pad_bytes, size_t(g->count[s]) * row,
                               hipMemcpyDeviceToDevice, streams[r]) != hipSuccess)
                return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: compaction copy failed");
    }
    return UKFB_OK;
}
}  // namespace

int ukfb_group_gather_means(ukfb_group* g, void* const* out_dev) {
    if (!g || !out_dev) return UKFB_ERR_INVALID_ARG;
    const size_t n = g->engines.size();
    bool shared = false;   // two shards on one device: RCCL wants one rank per device
    for (size_t a = 0; a < n; ++a)
        for (size_t b = a + 1; b < n; ++b) shared = shared || g->devices[a] == g->devices[b];
    const size_t row = size_t(g->S) * g->tsize, pad_bytes = size_t(g->max_count) * row;
    if (g->send_pad.empty()) {
        if (!shared) {
            Rccl& rc = rccl();
            if (!rc.ok()) return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: RCCL (librccl.so.1) could not be loaded");
            g->comms.assign(n, nullptr);
            const ncclResult_t st = rc.CommInitAll(g->comms.data(), int(n), g->devices.data());
            if (st != ncclSuccess) {
                g->comms.clear();
                return gfail(UKFB_ERR_HIP, std::string("ncclCommInitAll: ") + rc.GetErrorString(st));
            }
        }
        g->send_pad.assign(n, nullptr);
        g->recv_pad.assign(n, nullptr);
        g->staged.assign(n, nullptr);
        g->pulled.assign(n, nullptr);
        for (size_t r = 0; r < n; ++r) {
            ukfb::DeviceScope on_device(g->devices[r]);
            if (on_device.err != hipSuccess || hipMalloc(&g->send_pad[r], pad_bytes) != hipSuccess ||
                hipMalloc(&g->recv_pad[r], pad_bytes * n) != hipSuccess || hipMemset(g->send_pad[r], 0, pad_bytes) != hipSuccess ||
                hipEventCreateWithFlags(&g->staged[r], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&g->pulled[r], hipEventDisableTiming) != hipSuccess) {
                release_gather(g);
                return gfail(UKFB_ERR_HIP, "ukfb_group_gather_means: staging allocation failed");
            }
        }
    }
    std::vector<hipStream_t> streams(n);
    int rc = gather_stage(g, streams);
    if (!rc) rc = shared ? gather_exchange_copies(g, streams) : gather_exchange_rccl(g, streams);
    if (!rc) rc = gather_compact(g, streams, out_dev);
    g->last_gather_exchange = shared ? 2 : 1;
    return rc;   // stream-ordered: ukfb_group_sync (or the shard's stream) completes it
}

int ukfb_group_last_gather_exchange(const ukfb_group* g) { return g ? g->last_gather_exchange : -1; }

}  // extern "C"
