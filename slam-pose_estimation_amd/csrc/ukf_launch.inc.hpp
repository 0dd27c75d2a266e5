// ukf_launch.inc.hpp -- typed launch of ukf_kernel<T, M, G, predict, update>; included by the four
// per-(precision, model) translation units so that they compile in parallel.
#pragma once

#include "ukf_engine.hpp"
#include "ukf_kernel.hpp"
#include "ukf_kernel16.hpp"

namespace ukfb {

template <class T, class M, int G> static int launch_g(ukfb_engine* e, const LaunchReq& r, const KArgs<T>& args) {
    constexpr int FPW = 64 / G;
    const int64_t grid = (e->cap + FPW - 1) / FPW;
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
        hipLaunchKernelGGL((ukf_kernel<T, M, G, true, true>), gd, bd, lds, e->stream, args);
    else if (r.do_predict)
        hipLaunchKernelGGL((ukf_kernel<T, M, G, true, false>), gd, bd, lds, e->stream, args);
    else
        hipLaunchKernelGGL((ukf_kernel<T, M, G, false, true>), gd, bd, lds, e->stream, args);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("kernel launch", err);
        return UKFB_ERR_HIP;
    }
    return UKFB_OK;
}

// tuned kernel: one DPP row per filter (ukf_kernel16.hpp)
template <class T, class M> static int launch_row16(ukfb_engine* e, const LaunchReq& r, const KArgs<T>& args) {
    constexpr int FPW = 4;
    const int64_t grid = (e->cap + FPW - 1) / FPW;
    const int lds = FPW * lds_bytes_per_filter16<T, M>();
    const char* mode = r.do_predict ? (r.do_update ? "cycle" : "predict") : "update";
    e->last_kernel = std::string("ukf_kernel16<") + (sizeof(T) == 8 ? "f64" : "f32") + "," +
                     (M::MODEL == 0 ? "pose" : "orient") + "," + mode + ">";
    e->last_lds = lds;
    e->last_fpw = FPW;
    e->last_grid = grid;
    if (grid == 0) return UKFB_OK;
    const dim3 gd((unsigned)grid), bd(64);
    if (r.do_predict && r.do_update)
        hipLaunchKernelGGL((ukf_kernel16<T, M, true, true>), gd, bd, lds, e->stream, args);
    else if (r.do_predict)
        hipLaunchKernelGGL((ukf_kernel16<T, M, true, false>), gd, bd, lds, e->stream, args);
    else
        hipLaunchKernelGGL((ukf_kernel16<T, M, false, true>), gd, bd, lds, e->stream, args);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("kernel launch", err);
        return UKFB_ERR_HIP;
    }
    return UKFB_OK;
}

template <class T, class M> static int launch_typed(ukfb_engine* e, const LaunchReq& r) {
    KArgs<T> a{};
    a.n = e->cap;
    a.mu = static_cast<T*>(e->mu);
    a.cov = static_cast<T*>(e->cov);
    a.status = e->status;
    a.initialised = e->init;
    a.Rn = static_cast<const T*>(e->Rn);
    a.Rn_stride = e->Rn_per_filter ? int64_t(M::D) * M::D : 0;
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
    a.active = r.active_dev;
    a.mean_tol = T(e->cfg.mean_tol);
    a.mean_max_it = e->cfg.mean_max_iter;
    a.gate_chi2 = T(e->cfg.gate_chi2);
    switch (e->cfg.lanes_per_filter) {
        case 64: return launch_g<T, M, 64>(e, r, a);
        case 32: return launch_g<T, M, 32>(e, r, a);
        default: return launch_row16<T, M>(e, r, a);
    }
}

}  // namespace ukfb
