// ukf_oracle_capi.cpp -- C entry points over the CPU ORACLE (test infrastructure only).
//
// Batched loops around oracle/ukf_oracle.hpp so that tests/ (ctypes), __graft_entry__.smoke()
// and bench.py's cpu_baseline leg can run the oracle on the same AoS arrays the product's
// C-ABI (include/ukf_batch.h) takes.  Nothing in the product links or loads this library.
// PARITY UNPINNED -- see the header of ukf_oracle.hpp.
//
// Array conventions (all double at this boundary; prec=1 computes in float internally):
//   Pose:   mu[n][13] = p(3) q(x,y,z,w) v(3) w(3);  cov[n][12][12] row-major
//   Orient: mu[n][14] = q(x,y,z,w) v(3) bg(3) ba(3) g;  cov[n][13][13]
//   z[n][3], Q[n][3][3] (leading m x m block used), status[n] receives ST_* bits (overwritten).
#include "ukf_oracle.hpp"

#include <omp.h>

#include <vector>

using namespace ukf_oracle;

extern "C" {

struct ukfo_config {
    double mean_tol;
    int32_t mean_max_it;
    double gate_chi2;
    double min_dt;
    double max_dt;
};

void ukfo_default_config(ukfo_config* c) {
    c->mean_tol = 1e-6;
    c->mean_max_it = 10000;
    c->gate_chi2 = -1.0;
    c->min_dt = 1.0e-9;                                 // UnscentedKalmanFilter.hpp:31
    c->max_dt = std::numeric_limits<double>::max();     // UnscentedKalmanFilter.hpp:32
}

int ukfo_max_threads() { return omp_get_max_threads(); }

}  // extern "C"

namespace {

Config to_cfg(const ukfo_config* c) {
    Config k;
    if (c) {
        k.mean_tol = c->mean_tol;
        k.mean_max_it = c->mean_max_it;
        k.gate_chi2 = c->gate_chi2;
    }
    return k;
}

template <class T> void load(const double* src, T* dst, int n) {
    for (int i = 0; i < n; ++i) dst[i] = T(src[i]);
}
template <class T> void store(const T* src, double* dst, int n) {
    for (int i = 0; i < n; ++i) dst[i] = double(src[i]);
}

template <class T>
void pose_predict_batch(int64_t n, double* mu, double* cov, const double* R, int R_per_filter, const double* acc_mu,
                        const double* acc_cov, int acc_cov_per_filter, const double* dt, int dt_per_filter,
                        const ukfo_config* c, uint32_t* status, int threads) {
    const Config cfg = to_cfg(c);
    ukfo_config dc;
    ukfo_default_config(&dc);
    const double min_dt = c ? c->min_dt : dc.min_dt, max_dt = c ? c->max_dt : dc.max_dt;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) {
        const double d = dt[dt_per_filter ? i : 0];
        uint32_t st = gate_dt(d, min_dt, max_dt);
        if (st == ST_OK) {
            T m[13], s[144], r[144], a[3], ac[9];
            load(mu + i * 13, m, 13);
            load(cov + i * 144, s, 144);
            load(R + (R_per_filter ? i * 144 : 0), r, 144);
            const T* ap = nullptr;
            if (acc_mu) {
                load(acc_mu + i * 3, a, 3);
                ap = a;
                load(acc_cov + (acc_cov_per_filter ? i * 9 : 0), ac, 9);
            }
            st = pose_predict<T>(m, s, r, ap, ac, T(d), cfg);
            if (!(st & ST_ERR_CHOLESKY)) {
                store(m, mu + i * 13, 13);
                store(s, cov + i * 144, 144);
            }
        }
        if (status) status[i] = st;
    }
}

template <class T>
void pose_update_batch(int64_t n, double* mu, double* cov, const int32_t* model, int model_per_filter, const double* z,
                       const double* Q, const ukfo_config* c, uint32_t* status, int threads) {
    const Config cfg = to_cfg(c);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) {
        const int mid = model[model_per_filter ? i : 0];
        uint32_t st = ST_INACTIVE;
        if (mid >= 0) {
            T m[13], s[144], zz[3], q[9];
            load(mu + i * 13, m, 13);
            load(cov + i * 144, s, 144);
            load(z + i * 3, zz, 3);
            load(Q + i * 9, q, 9);
            st = pose_update<T>(m, s, mid, zz, q, cfg);
            if (!(st & (ST_ERR_CHOLESKY | ST_REJECTED_GATE | ST_INACTIVE))) {
                store(m, mu + i * 13, 13);
                store(s, cov + i * 144, 144);
            }
        }
        if (status) status[i] = st;
    }
}

template <class T>
void orient_predict_batch(int64_t n, double* mu, double* cov, const double* R, int R_per_filter, const double* acc,
                          const double* gyro, double tau_g, double tau_a, const double* earth, const double* dt,
                          int dt_per_filter, const ukfo_config* c, uint32_t* status, int threads) {
    const Config cfg = to_cfg(c);
    ukfo_config dc;
    ukfo_default_config(&dc);
    const double min_dt = c ? c->min_dt : dc.min_dt, max_dt = c ? c->max_dt : dc.max_dt;
    OrientParams<T> p;
    p.gyro_bias_tau = T(tau_g);
    p.acc_bias_tau = T(tau_a);
    for (int k = 0; k < 3; ++k) p.earth_rotation[k] = T(earth[k]);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) {
        const double d = dt[dt_per_filter ? i : 0];
        uint32_t st = gate_dt(d, min_dt, max_dt);
        if (st == ST_OK) {
            T m[14], s[169], r[169], a[3], w[3];
            load(mu + i * 14, m, 14);
            load(cov + i * 169, s, 169);
            load(R + (R_per_filter ? i * 169 : 0), r, 169);
            load(acc + i * 3, a, 3);
            load(gyro + i * 3, w, 3);
            st = orient_predict<T>(m, s, r, a, w, p, T(d), cfg);
            if (!(st & ST_ERR_CHOLESKY)) {
                store(m, mu + i * 14, 14);
                store(s, cov + i * 169, 169);
            }
        }
        if (status) status[i] = st;
    }
}

template <class T>
void orient_update_batch(int64_t n, double* mu, double* cov, const uint8_t* active, const double* z, const double* Q,
                         const ukfo_config* c, uint32_t* status, int threads) {
    const Config cfg = to_cfg(c);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) {
        uint32_t st = ST_INACTIVE;
        if (!active || active[i]) {
            T m[14], s[169], zz[3], q[9];
            load(mu + i * 14, m, 14);
            load(cov + i * 169, s, 169);
            load(z + i * 3, zz, 3);
            load(Q + i * 9, q, 9);
            st = orient_update<T>(m, s, zz, q, cfg);
            if (!(st & (ST_ERR_CHOLESKY | ST_REJECTED_GATE | ST_ERR_NONFINITE_MEAS))) {
                store(m, mu + i * 14, 14);
                store(s, cov + i * 169, 169);
            }
        }
        if (status) status[i] = st;
    }
}

int clamp_threads(int t) {
    if (t <= 0) return omp_get_max_threads();
    return t;
}

}  // namespace

extern "C" {

// PoseUKF::predictionStep(dt) for n filters.  acc_mu may be NULL (constant-velocity branch for
// every filter) or [n][3] with NaN rows selecting that branch per filter (PoseUKF.cpp:188).
int ukfo_pose_predict(int64_t n, int prec, double* mu, double* cov, const double* R, int R_per_filter,
                      const double* acc_mu, const double* acc_cov, int acc_cov_per_filter, const double* dt,
                      int dt_per_filter, const ukfo_config* cfg, uint32_t* status, int threads) {
    threads = clamp_threads(threads);
    if (prec == 0)
        pose_predict_batch<double>(n, mu, cov, R, R_per_filter, acc_mu, acc_cov, acc_cov_per_filter, dt, dt_per_filter,
                                   cfg, status, threads);
    else
        pose_predict_batch<float>(n, mu, cov, R, R_per_filter, acc_mu, acc_cov, acc_cov_per_filter, dt, dt_per_filter,
                                  cfg, status, threads);
    return 0;
}

// PoseUKF::integrateMeasurement for n filters; model[i] < 0 leaves filter i untouched.
int ukfo_pose_update(int64_t n, int prec, double* mu, double* cov, const int32_t* model, int model_per_filter,
                     const double* z, const double* Q, const ukfo_config* cfg, uint32_t* status, int threads) {
    threads = clamp_threads(threads);
    if (prec == 0)
        pose_update_batch<double>(n, mu, cov, model, model_per_filter, z, Q, cfg, status, threads);
    else
        pose_update_batch<float>(n, mu, cov, model, model_per_filter, z, Q, cfg, status, threads);
    return 0;
}

int ukfo_orient_predict(int64_t n, int prec, double* mu, double* cov, const double* R, int R_per_filter,
                        const double* acc, const double* gyro, double tau_g, double tau_a, const double* earth,
                        const double* dt, int dt_per_filter, const ukfo_config* cfg, uint32_t* status, int threads) {
    threads = clamp_threads(threads);
    if (prec == 0)
        orient_predict_batch<double>(n, mu, cov, R, R_per_filter, acc, gyro, tau_g, tau_a, earth, dt, dt_per_filter,
                                     cfg, status, threads);
    else
        orient_predict_batch<float>(n, mu, cov, R, R_per_filter, acc, gyro, tau_g, tau_a, earth, dt, dt_per_filter, cfg,
                                    status, threads);
    return 0;
}

int ukfo_orient_update(int64_t n, int prec, double* mu, double* cov, const uint8_t* active, const double* z,
                       const double* Q, const ukfo_config* cfg, uint32_t* status, int threads) {
    threads = clamp_threads(threads);
    if (prec == 0)
        orient_update_batch<double>(n, mu, cov, active, z, Q, cfg, status, threads);
    else
        orient_update_batch<float>(n, mu, cov, active, z, Q, cfg, status, threads);
    return 0;
}

// predictionStepFromSampleTime gate for n filters (UnscentedKalmanFilter.hpp:83-100): updates
// last_us in place, writes dt and the gate status.
int ukfo_gate_timestamps(int64_t n, const int64_t* ts_us, int64_t* last_us, double min_dt, double max_dt,
                         double* dt_out, uint32_t* status) {
    for (int64_t i = 0; i < n; ++i) status[i] = gate_timestamp(ts_us[i], &last_us[i], min_dt, max_dt, &dt_out[i]);
    return 0;
}

// OrientationUKF::getRotationRate (OrientationUKF.cpp:74-77)
int ukfo_orient_rotation_rate(int64_t n, const double* mu, const double* gyro, const double* earth, double* out) {
    for (int64_t i = 0; i < n; ++i) orient_rotation_rate<double>(mu + i * 14, gyro + i * 3, earth, out + i * 3);
    return 0;
}

// Unit-level hooks for the known-answer tests.
void ukfo_so3_exp(const double* v, double scale, double* q) { so3_exp<double>(v, scale, q); }
void ukfo_so3_log(const double* q, double* v) { so3_log<double>(q, v); }
void ukfo_so3_exp_f32(const double* v, double scale, double* q) {
    float vv[3] = {float(v[0]), float(v[1]), float(v[2])}, qq[4];
    so3_exp<float>(vv, float(scale), qq);
    for (int k = 0; k < 4; ++k) q[k] = qq[k];
}
void ukfo_quat_rotate(const double* q, const double* v, double* r) { quat_rotate<double>(q, v, r); }
void ukfo_quat_to_matrix(const double* q, double* R) { quat_to_matrix<double>(q, R); }
void ukfo_pose_boxplus(double* x, const double* d) { PoseManifold<double>::boxplus(x, d); }
void ukfo_pose_boxminus(const double* x, const double* y, double* d) { PoseManifold<double>::boxminus(x, y, d); }
void ukfo_orient_boxplus(double* x, const double* d) { OrientManifold<double>::boxplus(x, d); }
void ukfo_orient_boxminus(const double* x, const double* y, double* d) { OrientManifold<double>::boxminus(x, y, d); }
int ukfo_cholesky12(const double* A, double* L) { return cholesky_lower<double, 12>(A, L) ? 0 : 1; }
void ukfo_pose_process(double* x, const double* acc, double dt) { pose_process<double>(x, acc, dt); }
void ukfo_orient_process(double* x, const double* acc, const double* gyro, double tau_g, double tau_a,
                         const double* earth, double dt) {
    OrientParams<double> p;
    p.gyro_bias_tau = tau_g;
    p.acc_bias_tau = tau_a;
    for (int k = 0; k < 3; ++k) p.earth_rotation[k] = earth[k];
    orient_process<double>(x, acc, gyro, p, dt);
}
double ukfo_earthw() { return earth_angular_velocity(); }

}  // extern "C"
