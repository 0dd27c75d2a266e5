// ukf_launch.inc.hpp -- typed launch of ukf_kernel<T, M, G, predict, update>; included by the four
// per-(precision, model) translation units so that they compile in parallel.
#pragma once

#include <cstdio>
#include <cstdlib>

#include "ukf_engine.hpp"
#include "ukf_kernel.hpp"
#include "ukf_kernel16.hpp"

namespace ukfb {

template <class T, class M, int G> static int launch_g(ukfb_engine* e, const LaunchReq& r, const KArgs<T>& args) {
    constexpr int FPW = 64 / G;
    const int64_t grid = (args.n + FPW - 1) / FPW;
    const int lds = FPW * lds_bytes_per_filter<T, M>();
    const char* mode = r.do_predict ? (r.do_update ? "cycle" : "predict") : "update";
    e->last_kernel = std::string("ukf_kernel<") + (sizeof(T) == 8 ? "f64" : "f32") + "," +
                     (M::MODEL == 0 ? "pose" : "orient") + ",G" + std::to_string(G) + "," + mode + ">";
    e->last_lds = lds;
    e->last_fpw = FPW;
    e->last_grid = grid;
    if (grid == 0) return UKFB_OK;
    const dim3 gd((unsigned)grid), bd(64);
    if (r.do_predict && r.do_update)
        hipLaunchKernelGGL((ukf_kernel<T, M, G, true, true>), gd, bd, lds, main_stream(e), args);
    else if (r.do_predict)
        hipLaunchKernelGGL((ukf_kernel<T, M, G, true, false>), gd, bd, lds, main_stream(e), args);
    else
        hipLaunchKernelGGL((ukf_kernel<T, M, G, false, true>), gd, bd, lds, main_stream(e), args);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("kernel launch", err);
        return UKFB_ERR_HIP;
    }
    return UKFB_OK;
}

// The instantiation of the tuned kernel for a launch shape: go(kernel) is called with the matching ukf_kernel16<...> (one of
// twelve per (engine precision, model, compute type)).  level: 0 general, 1 streams only, 2 plain (ukf_kernel16.hpp).
template <class T, class M, class TC, class Go>
static void with_kernel16(bool indirect, bool multi, bool do_predict, bool do_update, int level, Go&& go) {
    using MC = typename M::template rebind<TC>;
    if (indirect) {
        if (level >= 1) go(ukf_kernel16<TC, MC, true, true, false, true, 1, T>);
        else go(ukf_kernel16<TC, MC, true, true, false, true, 0, T>);
    } else if (multi) {
        if (level == 2) go(ukf_kernel16<TC, MC, true, true, true, false, 2, T>);
        else go(ukf_kernel16<TC, MC, true, true, true, false, 0, T>);
    } else if (do_predict && do_update) {
        if (level == 2) go(ukf_kernel16<TC, MC, true, true, false, false, 2, T>);
        else if (level == 1) go(ukf_kernel16<TC, MC, true, true, false, false, 1, T>);
        else go(ukf_kernel16<TC, MC, true, true, false, false, 0, T>);
    } else if (do_predict) {
        if (level >= 1) go(ukf_kernel16<TC, MC, true, false, false, false, 2, T>);
        else go(ukf_kernel16<TC, MC, true, false, false, false, 0, T>);
    } else {
        if (level == 2) go(ukf_kernel16<TC, MC, false, true, false, false, 2, T>);
        else if (level == 1) go(ukf_kernel16<TC, MC, false, true, false, false, 1, T>);
        else go(ukf_kernel16<TC, MC, false, true, false, false, 0, T>);
    }
}

#if defined(UKFB_STAMPS) || defined(UKFB_COUNTS)
// Diagnostic builds (tools/phase_stamps.py, tools/trip_counts.py): every launch is synchronous; the per-marker s_memtime stamps of
// all wavefronts are reduced to mean cycles between consecutive executed markers and appended to the file
// named by UKFB_STAMP_OUT (one line per launch: kernel name, then marker_index:mean_delta pairs).
template <class T, class M, class TC = T> static int launch_row16_stamped(ukfb_engine* e, const LaunchReq& r, KArgs<T> args, int64_t grid, int lds, int level) {
    static unsigned long long* dbuf = nullptr;
    static size_t dcap = 0;
    const size_t need = size_t(grid) * UKFB_MAX_STAMPS;
    if (need > dcap) {
        if (dbuf) (void)hipFree(dbuf);
        if (hipMalloc(reinterpret_cast<void**>(&dbuf), need * 8) != hipSuccess) return UKFB_ERR_HIP;
        dcap = need;
    }
    (void)hipMemsetAsync(dbuf, 0, need * 8, main_stream(e));
    args.stamps = dbuf;
    const dim3 gd((unsigned)grid), bd(64);
#if defined(UKFB_COUNTS)
    {
        unsigned long long zero[DBG_N] = {0};
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(ukfb_dbg), zero, sizeof(zero), 0, hipMemcpyHostToDevice, main_stream(e));
    }
#endif
    with_kernel16<T, M, TC>(args.fidx != nullptr, r.cycles > 0, r.do_predict, r.do_update, level,
                            [&](auto kern) { hipLaunchKernelGGL(kern, gd, bd, lds, main_stream(e), args); });
    if (hipStreamSynchronize(main_stream(e)) != hipSuccess) return UKFB_ERR_HIP;
    const char* path = getenv("UKFB_STAMP_OUT");
    if (!path) return UKFB_OK;
    std::vector<unsigned long long> h(need);
    if (hipMemcpy(h.data(), dbuf, need * 8, hipMemcpyDeviceToHost) != hipSuccess) return UKFB_ERR_HIP;
#if defined(UKFB_COUNTS)
    {   // histogram of every counter slot over the wavefronts of the launch, then the exp / log path counters
        unsigned long long dbg[DBG_N] = {0};
        if (hipMemcpyFromSymbol(dbg, HIP_SYMBOL(ukfb_dbg), sizeof(dbg)) != hipSuccess) return UKFB_ERR_HIP;
        if (FILE* f = fopen(path, "a")) {
            fprintf(f, "%s grid=%lld", e->last_kernel.c_str(), (long long)grid);
            for (int k = 0; k < 4; ++k) {
                int64_t hist[18] = {0};
                bool any = false;
                for (int64_t w = 0; w < grid; ++w) {
                    const unsigned long long v = h[size_t(w) * UKFB_MAX_STAMPS + k];
                    if (!v) continue;
                    any = true;
                    ++hist[(v - 1) < 17 ? (v - 1) : 17];
                }
                if (!any) continue;
                fprintf(f, " slot%d=", k);
                for (int b = 0; b < 18; ++b)
                    if (hist[b]) fprintf(f, "%d:%lld,", b, (long long)hist[b]);
            }
            fprintf(f, " dbg=");
            for (int k = 0; k < DBG_N; ++k) fprintf(f, "%llu,", dbg[k]);
            fprintf(f, "\n");
            fclose(f);
        }
        return UKFB_OK;
    }
#endif
    double sum[UKFB_MAX_STAMPS] = {0};
    int64_t cnt[UKFB_MAX_STAMPS] = {0};
    double life = 0;
    for (int64_t w = 0; w < grid; ++w) {
        const unsigned long long* s = h.data() + size_t(w) * UKFB_MAX_STAMPS;
        int prev = -1;
        unsigned long long first = 0, last = 0;
        for (int k = 0; k < UKFB_MAX_STAMPS; ++k) {
            if (!s[k]) continue;
            if (prev >= 0) { sum[prev] += double(s[k] - s[prev]); ++cnt[prev]; } else first = s[k];
            prev = k;
            last = s[k];
        }
        life += double(last - first);
    }
    if (FILE* f = fopen(path, "a")) {
        fprintf(f, "%s grid=%lld life=%.1f", e->last_kernel.c_str(), (long long)grid, life / double(grid));
        for (int k = 0; k < UKFB_MAX_STAMPS; ++k)
            if (cnt[k]) fprintf(f, " %d:%.1f", k, sum[k] / double(cnt[k]));
        fprintf(f, "\n");
        fclose(f);
    }
    return UKFB_OK;
}
#endif

constexpr int64_t SPLIT_MIN_FILTERS = 16384;
// (UKFB_SPLIT_MAX in the environment moves the upper bound: measurements only)
static int64_t split_max_filters() {
    static const int64_t v = [] {
        const char* s = std::getenv("UKFB_SPLIT_MAX");
        const long long x = s ? std::atoll(s) : 0;
        return int64_t(x > 0 ? x : 262144);
    }();
    return v;
}

// tuned kernel: one DPP row per filter (ukf_kernel16.hpp).  T: the engine's precision = the type of every array in HBM;
// TC: the type the kernel computes in (TC = double with T = float: the wide-arithmetic mode of the fp32 engines, whose LDS slice
// and register budget are the fp64 kernel's).
template <class T, class M, class TC = T> static int launch_row16(ukfb_engine* e, const LaunchReq& r, const KArgs<T>& args) {
    using MC = typename M::template rebind<TC>;
    constexpr int FPW = 4;
    const int64_t grid = (args.n + FPW - 1) / FPW;
    // (UKFB_LDS_PAD_BYTES in the environment requests that much more dynamic LDS per workgroup: occupancy experiments only)
    static const int lds_pad = [] { const char* s = std::getenv("UKFB_LDS_PAD_BYTES"); return s ? std::atoi(s) : 0; }();
    const int lds = FPW * lds_bytes_per_filter16<TC, MC>() + lds_pad;
    const bool multi = r.cycles > 0;   // ukfb_cycle_multi_dev: fused cycles only (checked by the caller)
    // What the kernel may take as compile-time facts about this launch (ukf_kernel16<..., PLAIN>):
    //   streams only (level 1)  no per-filter timestamps / time steps / activity flags, the accept-any gate, a fresh status word
    //   plain        (level 2)  ... and ONE full-3-vector measurement model for the launch (prediction-only launches: level 1 = 2)
    // Indirect launches qualify for level 1 when their list is a bucketed filter list (event rounds carry timestamps); multi-cycle
    // launches for level 2 when they have no schedule.
    const char* const plain_env = std::getenv("UKFB_NO_PLAIN_KERNEL");   // (A/B and tests: =1 keeps the general kernel; read per launch)
    const bool plain_off = plain_env && plain_env[0] == '1';
    const bool streams_only = !plain_off && !args.ts && !args.dt && !args.active && !args.status_accumulate && args.gate_chi2 < T(0) &&
                              (!multi || !args.cyc_sched) && (!args.fidx || args.fidx_inputs);
    const bool full3 = !args.meas && (M::MODEL != 0 ? args.meas_uniform == 9
                                                    : (args.meas_uniform == 0 || args.meas_uniform == 4 || args.meas_uniform == 8));
    int level = 0;
    if (streams_only) {
        if (args.fidx) level = 1;
        else if (multi) level = full3 ? 2 : 0;
        else if (!r.do_update) level = 2;
        else level = full3 ? 2 : 1;
    }
    const char* const suffix = level == 2 ? "-plain" : (level == 1 ? "-streams" : "");
    const std::string mode = std::string(multi ? "multicycle" : (r.do_predict ? (r.do_update ? (args.fidx_inputs ? "cycle-bucketed" : "cycle") : "predict") : "update")) + suffix;
    e->last_kernel = std::string("ukf_kernel16<") + (sizeof(T) == 8 ? "f64" : (sizeof(TC) == 8 ? "f32-wide" : "f32")) + "," +
                     (M::MODEL == 0 ? "pose" : "orient") + "," + mode + ">";
    e->last_lds = lds;
    e->last_fpw = FPW;
    e->last_grid = grid;
    if (grid == 0) return UKFB_OK;
    const dim3 gd((unsigned)grid), bd(64);
#if defined(UKFB_STAMPS) || defined(UKFB_COUNTS)
    return launch_row16_stamped<T, M, TC>(e, r, args, grid, lds, level);
#endif
    // Two half launches on two streams (engines that own their stream, direct launches of SPLIT_MIN <= n < SPLIT_MAX filters):
    // launch k + 1's first half follows launch k's first half on `stream`, its second half follows launch k's second half on
    // stream_b -- the tail of one half (a partly empty last round of workgroups) overlaps the head of the other stream's next
    // kernel.  Filters are independent, the halves touch disjoint filters: results are bit-identical.
    if (!args.fidx && !r.no_split && e->stream_b && e->cfg.split_streams && args.n >= SPLIT_MIN_FILTERS && args.n < split_max_filters()) {
        KArgs<T> h1 = args, h2 = args;
        h1.n = (args.n / 2 + FPW - 1) / FPW * FPW;
        h2.item0 = h1.n;
        const dim3 g1((unsigned)(h1.n / FPW)), g2((unsigned)((args.n - h1.n + FPW - 1) / FPW));
        hipStream_t sa = e->stream, sb = e->stream_b;    // (raw: a pending second half is NOT joined, that is the point)
        (void)hipEventRecord(e->ev_a, sa);               // stream_b sees everything enqueued on `stream` so far
        (void)hipStreamWaitEvent(sb, e->ev_a, 0);
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, g1, bd, lds, sa, h1);
            hipLaunchKernelGGL(kern, g2, bd, lds, sb, h2);
        };
        with_kernel16<T, M, TC>(false, multi, r.do_predict, r.do_update, level, go);
        e->split_pending = true;
        const hipError_t serr = hipGetLastError();
        if (serr != hipSuccess) {
            set_error("kernel launch (split)", serr);
            return UKFB_ERR_HIP;
        }
        return UKFB_OK;
    }
    if (args.fidx) {   // indirect launch (event rounds): the fused cycle over a list of filters
        if (multi || !(r.do_predict && r.do_update)) {
            set_error("indirect launches run the single fused cycle", hipErrorInvalidValue);
            return UKFB_ERR_INVALID_ARG;
        }
    }
    with_kernel16<T, M, TC>(args.fidx != nullptr, multi, r.do_predict, r.do_update, level,
                            [&](auto kern) { hipLaunchKernelGGL(kern, gd, bd, lds, main_stream(e), args); });
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("kernel launch", err);
        return UKFB_ERR_HIP;
    }
    return UKFB_OK;
}

template <class T, class M> static int launch_typed(ukfb_engine* e, const LaunchReq& r) {
    KArgs<T> a{};
    a.n = r.n_items >= 0 ? r.n_items : e->cap;
    a.fidx = r.filter_index_dev;
    a.fidx_inputs = (r.filter_index_dev && r.inputs_by_filter) ? 1 : 0;
    a.status_accumulate = r.status_accumulate ? 1 : 0;
    a.mu = static_cast<T*>(e->mu);
    a.cov = static_cast<T*>(e->cov);
    a.status = e->status;
    a.initialised = e->init;
    a.Rn = static_cast<const T*>(e->Rn);
    a.Rn_stride = e->Rn_per_filter ? int64_t(M::D) * M::D : 0;
    a.noise_iso = (e->noise_iso && !e->Rn_per_filter) ? 1 : 0;
    a.upd_short_ok = (e->noise_psd && !e->Rn_per_filter && !e->cfg.full_update_check) ? 1 : 0;
    a.Racc = static_cast<const T*>(e->Racc);
    a.in_a = static_cast<const T*>(e->in_a_bound ? e->in_a_bound : e->in_a);
    a.in_b = static_cast<const T*>(e->in_b_bound ? e->in_b_bound : e->in_b);
    for (int k = 0; k < 9; ++k) a.acc_cov[k] = T(e->acc_cov[k]);
    a.ninv_tau_g = T(-1.0) / T(e->tau_g);
    a.ninv_tau_a = T(-1.0) / T(e->tau_a);
    for (int k = 0; k < 3; ++k) a.earth[k] = T(e->earth[k]);
    a.dt_uniform = r.dt_uniform;
    a.dt = r.dt_dev;
    a.ts = r.ts_dev;
    a.last_ts = e->last_ts;
    a.min_dt = e->cfg.min_time_delta;
    a.max_dt = e->cfg.max_time_delta;
    a.meas_uniform = r.meas_uniform;
    a.meas = r.meas_dev;
    a.z = static_cast<const T*>(r.z_dev);
    a.Q = static_cast<const T*>(r.Q_dev);
    a.q_uniform = (r.q_uniform && r.cycles == 0) ? 1 : 0;
    a.active = r.active_dev;
    a.mean_tol = T(e->cfg.mean_tol);
    a.mean_max_it = e->cfg.mean_max_iter;
    a.gate_chi2 = T(e->cfg.gate_chi2);
    if (r.cycles > 0) {
        a.cyc_count = r.cycles;
        a.cyc_first = r.first_slot;
        a.cyc_ring = r.slots;
        a.cyc_items = e->cap;
        a.cyc_in = (r.in_a_slots ? 1 : 0) | (r.in_b_slots ? 2 : 0);
        if (r.in_a_slots) a.in_a = static_cast<const T*>(r.in_a_slots);
        if (r.in_b_slots) a.in_b = static_cast<const T*>(r.in_b_slots);
        if (r.sched_dt && r.sched_model) {
            a.cyc_sched = 1;
            for (int c = 0; c < r.cycles && c < UKFB_MAX_MULTI_CYCLES; ++c) {
                a.cyc_dt[c] = r.sched_dt[c];
                a.cyc_model[c] = r.sched_model[c];
            }
        }
    }
    // The one-wavefront-per-filter layouts (ablation of the brief's literal decomposition) ship in fp32 only: their fp64
    // instantiations need more than 256 VGPRs and the compiler parks the excess in AGPRs -- live-range-split copies of the
    // kind DESIGN.md 4.5 distrusts with this toolchain.  -DUKFB_GENERIC_F64=1 (make GENERIC_F64=1) builds them for
    // diagnostics; ukfb_set_config refuses the setting otherwise (ukfb_layout_supported tells).
#if !defined(UKFB_LAUNCH_WIDE)
    if constexpr (sizeof(T) == 4 || UKFB_GENERIC_F64 != 0) {
        switch (e->cfg.lanes_per_filter) {
            case 64: return launch_g<T, M, 64>(e, r, a);
            case 32: return launch_g<T, M, 32>(e, r, a);
            default: break;
        }
    }
#endif
#if defined(UKFB_LAUNCH_WIDE)
    return launch_row16<T, M, double>(e, r, a);
#else
    return launch_row16<T, M>(e, r, a);
#endif
}

}  // namespace ukfb
