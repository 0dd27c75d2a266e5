// ukf_oracle.hpp -- CPU ORACLE (test infrastructure, NOT the product path).
//
// A dependency-free C++17 restatement of the arithmetic that the reference
// (rock-slam/slam-pose_estimation) reaches through `ukfom::ukf<WState>` in the
// un-vendored, un-versioned third-party library slam/mtk (MTK / ukfom), plus the
// in-tree process / measurement models of the reference.
//
// PARITY UNPINNED: Eigen, boost, MTK and base-types are absent from the build
// container, so the reference cannot be compiled, and the reference holds no test,
// fixture or golden vector for this path.  What pins this
// oracle instead: an independent NumPy restatement (oracle/ukf_numpy.py) agreeing to
// <=1e-12, closed-form known answers (linear Kalman equivalence, identity model,
// exp/log round trips) and the committed fixtures under tests/golden/.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
// file.  The product (HIP engine behind include/ukf_batch.h) never links it.
//
// Published algorithm restated here: Hertzberg, Wagner, Frese, Schroeder,
// "Integrating generic sensor fusion algorithms with sound state representations
// through encapsulation of manifolds", Information Fusion 14(1), 2013 (UKF on
// manifolds: Cholesky sigma points without weights, boxplus/boxminus, iterated
// manifold mean, 1/2-weighted covariance), as implemented by MTK's ukfom/ukf.hpp,
// mtk/types/SOn.hpp, mtk/types/vect.hpp, mtk/src/mtkmath.hpp.
//
// Reference call sites this follows (file:line relative to /root/reference/src):
//   UnscentedKalmanFilter.hpp:83-125       time gate             -> gate_dt(), gate_timestamp()
//   pose_with_velocity/PoseUKF.cpp:7-69    measurement models    -> pose_measurement()
//   pose_with_velocity/PoseUKF.cpp:75-97   process models        -> pose_process()
//   pose_with_velocity/PoseUKF.cpp:112-173 integrateMeasurement  -> pose_update()
//   pose_with_velocity/PoseUKF.cpp:180-196 predictionStepImpl    -> pose_predict()
//   orientation_estimator/OrientationUKF.cpp:12-32  processModel -> orient_process()
//   orientation_estimator/OrientationUKF.cpp:34-39  velocityMeasurementModel
//   orientation_estimator/OrientationUKF.cpp:65-72  integrateMeasurement -> orient_update()
//   orientation_estimator/OrientationUKF.cpp:79-89  predictionStepImpl   -> orient_predict()
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace ukf_oracle {

// ---------------------------------------------------------------------------------
// status bits (same values as include/ukf_batch.h UKFB_ST_*)
// ---------------------------------------------------------------------------------
enum : uint32_t {
    ST_OK = 0u,
    ST_SKIPPED_FIRST_TS = 1u << 0,   // UnscentedKalmanFilter.hpp:86-90
    ST_SKIPPED_SMALL_DT = 1u << 1,   // UnscentedKalmanFilter.hpp:114-118
    ST_ERR_NEG_DT = 1u << 2,         // UnscentedKalmanFilter.hpp:110-113
    ST_ERR_DT_TOO_LARGE = 1u << 3,   // UnscentedKalmanFilter.hpp:119-122
    ST_ERR_NONFINITE_MEAS = 1u << 4, // UnscentedKalmanFilter.hpp:142-147
    ST_ERR_CHOLESKY = 1u << 5,       // ukfom: "Cholesky decomposition failed"
    ST_WARN_MEAN_NOCONV = 1u << 6,   // ukfom: "meanSigmaPoints() did not converge"
    ST_UNINITIALISED = 1u << 7,      // UnscentedKalmanFilter.hpp:53,59
    ST_INACTIVE = 1u << 8,           // batched engine only: filter masked out of this call
    ST_REJECTED_GATE = 1u << 9,      // batched engine only: mahalanobis gate rejected update
};

// measurement model ids (same values as include/ukf_batch.h UKFB_MEAS_*)
enum : int {
    MEAS_NONE = -1,
    MEAS_POS3 = 0,         // PoseUKF.cpp:7-12,  112-117
    MEAS_POS_XY = 1,       // PoseUKF.cpp:14-19, 119-124
    MEAS_POS_Z = 2,        // PoseUKF.cpp:21-26, 126-131
    MEAS_ORIENT_SO3 = 3,   // PoseUKF.cpp:28-33, 133-138
    MEAS_VEL3 = 4,         // PoseUKF.cpp:35-40, 140-145
    MEAS_VEL_XY = 5,       // PoseUKF.cpp:42-47, 147-152
    MEAS_VEL_Z = 6,        // PoseUKF.cpp:49-54, 154-159
    MEAS_XVEL_YAWVEL = 7,  // PoseUKF.cpp:56-62, 161-166
    MEAS_ANGVEL3 = 8,      // PoseUKF.cpp:64-69, 168-173
    MEAS_ORIENT_BODYVEL3 = 9,  // OrientationUKF.cpp:34-39, 65-72
};

struct Config {
    double mean_tol = 1e-6;     // ukfom meanSigmaPoints: while (mean_delta.norm() > 1e-6 ...)
    int mean_max_it = 10000;    // ukfom meanSigmaPoints: max_it
    double gate_chi2 = -1.0;    // <0: accept_any_mahalanobis_distance (PoseUKF.cpp:116)
};

// ---------------------------------------------------------------------------------
// MTK math (mtk/src/mtkmath.hpp) -- SURVEY Appendix A.1
// ---------------------------------------------------------------------------------
template <class T> struct MtkTolerance;
template <> struct MtkTolerance<double> { static constexpr double value = 1e-11; };
template <> struct MtkTolerance<float> { static constexpr float value = 1e-5f; };

// cos(sqrt(x2)) and sin(sqrt(x2))/sqrt(x2); Taylor pairs below eps^(1/4).
template <class T> inline void cos_sinc_sqrt(T x2, T& cosi, T& sinc) {
    const T taylor_0_bound = std::numeric_limits<T>::epsilon();
    const T taylor_2_bound = std::sqrt(taylor_0_bound);
    const T taylor_n_bound = std::sqrt(taylor_2_bound);
    if (x2 >= taylor_n_bound) {
        T x = std::sqrt(x2);
        cosi = std::cos(x);
        sinc = std::sin(x) / x;
        return;
    }
    const T inv[] = {T(1 / 3.), T(1 / 4.), T(1 / 5.), T(1 / 6.), T(1 / 7.), T(1 / 8.), T(1 / 9.)};
    cosi = T(1);
    sinc = T(1);
    T term = T(-1 / 2.) * x2;
    for (int i = 0; i < 3; ++i) {
        cosi += term;
        term *= inv[2 * i];
        sinc += term;
        term *= -inv[2 * i + 1] * x2;
    }
}

// Quaternions are stored in Eigen coefficient order (x, y, z, w).
enum { QX = 0, QY = 1, QZ = 2, QW = 3 };

// Eigen::Quaternion product a*b (Hamilton).
template <class T> inline void quat_mul(const T* a, const T* b, T* r) {
    T w = a[QW] * b[QW] - a[QX] * b[QX] - a[QY] * b[QY] - a[QZ] * b[QZ];
    T x = a[QW] * b[QX] + a[QX] * b[QW] + a[QY] * b[QZ] - a[QZ] * b[QY];
    T y = a[QW] * b[QY] + a[QY] * b[QW] + a[QZ] * b[QX] - a[QX] * b[QZ];
    T z = a[QW] * b[QZ] + a[QZ] * b[QW] + a[QX] * b[QY] - a[QY] * b[QX];
    r[QX] = x; r[QY] = y; r[QZ] = z; r[QW] = w;
}

template <class T> inline void quat_conj(const T* a, T* r) {
    r[QX] = -a[QX]; r[QY] = -a[QY]; r[QZ] = -a[QZ]; r[QW] = a[QW];
}

// Eigen::QuaternionBase::inverse(): conjugate / squaredNorm.
template <class T> inline void quat_inverse(const T* a, T* r) {
    T n2 = a[QX] * a[QX] + a[QY] * a[QY] + a[QZ] * a[QZ] + a[QW] * a[QW];
    r[QX] = -a[QX] / n2; r[QY] = -a[QY] / n2; r[QZ] = -a[QZ] / n2; r[QW] = a[QW] / n2;
}

// Eigen::QuaternionBase::_transformVector: v + w*uv + vec x uv with uv = 2 (vec x v).
template <class T> inline void quat_rotate(const T* q, const T* v, T* r) {
    T ux = q[QY] * v[2] - q[QZ] * v[1];
    T uy = q[QZ] * v[0] - q[QX] * v[2];
    T uz = q[QX] * v[1] - q[QY] * v[0];
    ux += ux; uy += uy; uz += uz;
    T cx = q[QY] * uz - q[QZ] * uy;
    T cy = q[QZ] * ux - q[QX] * uz;
    T cz = q[QX] * uy - q[QY] * ux;
    T r0 = v[0] + q[QW] * ux + cx;
    T r1 = v[1] + q[QW] * uy + cy;
    T r2 = v[2] + q[QW] * uz + cz;
    r[0] = r0; r[1] = r1; r[2] = r2;
}

// Eigen::QuaternionBase::toRotationMatrix (row-major 3x3).
template <class T> inline void quat_to_matrix(const T* q, T* R) {
    const T tx = T(2) * q[QX], ty = T(2) * q[QY], tz = T(2) * q[QZ];
    const T twx = tx * q[QW], twy = ty * q[QW], twz = tz * q[QW];
    const T txx = tx * q[QX], txy = ty * q[QX], txz = tz * q[QX];
    const T tyy = ty * q[QY], tyz = tz * q[QY], tzz = tz * q[QZ];
    R[0] = T(1) - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
    R[3] = txy + twz;          R[4] = T(1) - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = T(1) - (txx + tyy);
}

// MTK::SO3::exp(vec, scale): w = cos(|v| scale/2), vec = sinc(|v| scale/2) (scale/2) v.
template <class T> inline void so3_exp(const T* v, T scale, T* q) {
    const T s = scale / T(2);
    const T norm2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    T c, sc;
    cos_sinc_sqrt(s * s * norm2, c, sc);
    const T mult = sc * s;
    q[QX] = mult * v[0]; q[QY] = mult * v[1]; q[QZ] = mult * v[2]; q[QW] = c;
}

// MTK::SO3::log(q) = MTK::log(w, vec, scale=2, plus_minus_periodicity=true).
template <class T> inline void so3_log(const T* q, T* r) {
    T nv = std::sqrt(q[QX] * q[QX] + q[QY] * q[QY] + q[QZ] * q[QZ]);
    if (nv < MtkTolerance<T>::value) nv = MtkTolerance<T>::value;
    const T s = T(2) / nv * std::atan(nv / q[QW]);
    r[0] = s * q[QX]; r[1] = s * q[QY]; r[2] = s * q[QZ];
}

// MTK::SO3::boxplus: q <- q * exp(vec, scale) (right multiplication, no renormalisation).
template <class T> inline void so3_boxplus(T* q, const T* v, T scale = T(1)) {
    T d[4];
    so3_exp(v, scale, d);
    T r[4];
    quat_mul(q, d, r);
    q[0] = r[0]; q[1] = r[1]; q[2] = r[2]; q[3] = r[3];
}

// MTK::SO3::boxminus: log(other.conjugate() * q).
template <class T> inline void so3_boxminus(const T* q, const T* other, T* r) {
    T oc[4], d[4];
    quat_conj(other, oc);
    quat_mul(oc, q, d);
    so3_log(d, r);
}

// ---------------------------------------------------------------------------------
// Manifolds (MTK_BUILD_MANIFOLD compounds)
// ---------------------------------------------------------------------------------
// PoseWithVelocity.hpp:18-23: position(vect3) orientation(SO3) velocity(vect3) angular_velocity(vect3).
// stored: p[0..2] q[3..6](x,y,z,w) v[7..9] w[10..12]; tangent: p 0-2, q 3-5, v 6-8, w 9-11.
template <class T> struct PoseManifold {
    static constexpr int S = 13, D = 12;
    enum { P = 0, Q = 3, V = 7, W = 10 };
    static void boxplus(T* x, const T* d, T scale = T(1)) {
        for (int k = 0; k < 3; ++k) x[P + k] += scale * d[k];
        so3_boxplus(x + Q, d + 3, scale);
        for (int k = 0; k < 3; ++k) x[V + k] += scale * d[6 + k];
        for (int k = 0; k < 3; ++k) x[W + k] += scale * d[9 + k];
    }
    static void boxminus(const T* x, const T* y, T* d) {
        for (int k = 0; k < 3; ++k) d[k] = x[P + k] - y[P + k];
        so3_boxminus(x + Q, y + Q, d + 3);
        for (int k = 0; k < 3; ++k) d[6 + k] = x[V + k] - y[V + k];
        for (int k = 0; k < 3; ++k) d[9 + k] = x[W + k] - y[W + k];
    }
};

// OrientationState.hpp:20-26: orientation(SO3) velocity(vect3) bias_gyro(vect3) bias_acc(vect3) gravity(vect1).
// stored: q[0..3] v[4..6] bg[7..9] ba[10..12] g[13]; tangent: q 0-2, v 3-5, bg 6-8, ba 9-11, g 12.
template <class T> struct OrientManifold {
    static constexpr int S = 14, D = 13;
    enum { Q = 0, V = 4, BG = 7, BA = 10, G = 13 };
    static void boxplus(T* x, const T* d, T scale = T(1)) {
        so3_boxplus(x + Q, d, scale);
        for (int k = 0; k < 10; ++k) x[V + k] += scale * d[3 + k];
    }
    static void boxminus(const T* x, const T* y, T* d) {
        so3_boxminus(x + Q, y + Q, d);
        for (int k = 0; k < 10; ++k) d[3 + k] = x[V + k] - y[V + k];
    }
};

// Measurement spaces: plain Eigen vectors (m = 1..3) or RotationType (SO3).
template <class T, int M> struct VectManifold {
    static constexpr int S = M, D = M;
    static void boxplus(T* x, const T* d, T scale = T(1)) {
        for (int k = 0; k < M; ++k) x[k] += scale * d[k];
    }
    static void boxminus(const T* x, const T* y, T* d) {
        for (int k = 0; k < M; ++k) d[k] = x[k] - y[k];
    }
};
template <class T> struct SO3Manifold {
    static constexpr int S = 4, D = 3;
    static void boxplus(T* x, const T* d, T scale = T(1)) { so3_boxplus(x, d, scale); }
    static void boxminus(const T* x, const T* y, T* d) { so3_boxminus(x, y, d); }
};

// ---------------------------------------------------------------------------------
// Small dense helpers (row-major)
// ---------------------------------------------------------------------------------
// Eigen::LLT unblocked (size < 32): left-looking column Cholesky; false if pivot <= 0.
template <class T, int D> inline bool cholesky_lower(const T* A, T* L) {
    for (int i = 0; i < D * D; ++i) L[i] = T(0);
    for (int k = 0; k < D; ++k) {
        T x = A[k * D + k];
        for (int j = 0; j < k; ++j) x -= L[k * D + j] * L[k * D + j];
        if (!(x > T(0))) return false;
        x = std::sqrt(x);
        L[k * D + k] = x;
        for (int i = k + 1; i < D; ++i) {
            T a = A[i * D + k];
            for (int j = 0; j < k; ++j) a -= L[i * D + j] * L[k * D + j];
            L[i * D + k] = a / x;
        }
    }
    return true;
}

// Eigen fixed-size inverse for m <= 3 (1/x, adjugate/det).
template <class T> inline void inverse_small(const T* Sm, int m, T* Si) {
    if (m == 1) {
        Si[0] = T(1) / Sm[0];
    } else if (m == 2) {
        const T invdet = T(1) / (Sm[0] * Sm[3] - Sm[1] * Sm[2]);
        Si[0] = Sm[3] * invdet;  Si[1] = -Sm[1] * invdet;
        Si[2] = -Sm[2] * invdet; Si[3] = Sm[0] * invdet;
    } else {
        auto a = [&](int r, int c) { return Sm[r * 3 + c]; };
        const T c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
        const T c10 = a(2, 1) * a(0, 2) - a(2, 2) * a(0, 1);  // cofactor<1,0>
        const T c20 = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);  // cofactor<2,0>
        const T det = c00 * a(0, 0) + c10 * a(1, 0) + c20 * a(2, 0);
        const T invdet = T(1) / det;
        // inverse(r,c) = cofactor<c,r> / det
        Si[0] = c00 * invdet;
        Si[1] = c10 * invdet;
        Si[2] = c20 * invdet;
        Si[3] = (a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2)) * invdet;
        Si[4] = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * invdet;
        Si[5] = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * invdet;
        Si[6] = (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0)) * invdet;
        Si[7] = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * invdet;
        Si[8] = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * invdet;
    }
}

// ---------------------------------------------------------------------------------
// ukfom::ukf arithmetic -- SURVEY Appendix A.3-A.5
// ---------------------------------------------------------------------------------
// generate_sigma_points(mu, delta, sigma, X): X0 = mu+delta, X(2j+1) = mu+(delta+L.col(j)),
// X(2j+2) = mu+(delta-L.col(j)).  No scaling, no weights.
template <class T, class M>
inline bool generate_sigma_points(const T* mu, const T* delta, const T* sigma, T* X /*[2D+1][S]*/) {
    constexpr int S = M::S, D = M::D;
    T L[D * D];
    if (!cholesky_lower<T, D>(sigma, L)) return false;
    T d[D];
    for (int s = 0; s < S; ++s) X[s] = mu[s];
    for (int k = 0; k < D; ++k) d[k] = delta ? delta[k] : T(0);
    M::boxplus(X, d);
    int i = 1;
    for (int j = 0; j < D; ++j) {
        T* Xp = X + (i++) * S;
        for (int s = 0; s < S; ++s) Xp[s] = mu[s];
        for (int k = 0; k < D; ++k) d[k] = (delta ? delta[k] : T(0)) + L[k * D + j];
        M::boxplus(Xp, d);
        T* Xm = X + (i++) * S;
        for (int s = 0; s < S; ++s) Xm[s] = mu[s];
        for (int k = 0; k < D; ++k) d[k] = (delta ? delta[k] : T(0)) - L[k * D + j];
        M::boxplus(Xm, d);
    }
    return true;
}

// meanSigmaPoints: reference = X[0]; do { mean_delta = avg(Xi - reference);
// reference = reference + mean_delta; } while (norm > tol && ++i < max_it).
template <class T, class M>
inline bool mean_sigma_points(const T* X, int n, const Config& cfg, T* ref, int* iterations = nullptr) {
    constexpr int S = M::S, D = M::D;
    for (int s = 0; s < S; ++s) ref[s] = X[s];
    int it = 0;
    T norm;
    do {
        T md[D];
        for (int k = 0; k < D; ++k) md[k] = T(0);
        for (int i = 0; i < n; ++i) {
            T d[D];
            M::boxminus(X + i * S, ref, d);
            for (int k = 0; k < D; ++k) md[k] += d[k];
        }
        T n2 = T(0);
        for (int k = 0; k < D; ++k) {
            md[k] /= T(n);
            n2 += md[k] * md[k];
        }
        M::boxplus(ref, md);
        norm = std::sqrt(n2);
    } while (norm > T(cfg.mean_tol) && ++it < cfg.mean_max_it);
    if (iterations) *iterations = it + 1;
    return it < cfg.mean_max_it;
}

// covSigmaPoints: 0.5 * sum (Vi - mean)(Vi - mean)^T
template <class T, class M>
inline void cov_sigma_points(const T* mean, const T* V, int n, T* C /*[D][D]*/) {
    constexpr int S = M::S, D = M::D;
    for (int i = 0; i < D * D; ++i) C[i] = T(0);
    for (int i = 0; i < n; ++i) {
        T d[D];
        M::boxminus(V + i * S, mean, d);
        for (int r = 0; r < D; ++r)
            for (int c = 0; c < D; ++c) C[r * D + c] += d[r] * d[c];
    }
    for (int i = 0; i < D * D; ++i) C[i] *= T(0.5);
}

// crossCovSigmaPoints: 0.5 * sum (Xi - meanX)(Zi - meanZ)^T
template <class T, class MX, class MZ>
inline void cross_cov_sigma_points(const T* meanX, const T* meanZ, const T* X, const T* Z, int n,
                                   T* C /*[MX::D][MZ::D]*/) {
    constexpr int DX = MX::D, DZ = MZ::D;
    for (int i = 0; i < DX * DZ; ++i) C[i] = T(0);
    for (int i = 0; i < n; ++i) {
        T dx[DX], dz[DZ];
        MX::boxminus(X + i * MX::S, meanX, dx);
        MZ::boxminus(Z + i * MZ::S, meanZ, dz);
        for (int r = 0; r < DX; ++r)
            for (int c = 0; c < DZ; ++c) C[r * DZ + c] += dx[r] * dz[c];
    }
    for (int i = 0; i < DX * DZ; ++i) C[i] *= T(0.5);
}

// ukf::predict(g, R).  Returns status bits; on ST_ERR_CHOLESKY the state is untouched.
template <class T, class M, class G>
inline uint32_t ukf_predict(T* mu, T* sigma, G g, const T* R, const Config& cfg) {
    constexpr int S = M::S, D = M::D, N = 2 * D + 1;
    T X[N * S];
    if (!generate_sigma_points<T, M>(mu, nullptr, sigma, X)) return ST_ERR_CHOLESKY;
    for (int i = 0; i < N; ++i) g(X + i * S);
    uint32_t st = ST_OK;
    T m[S];
    if (!mean_sigma_points<T, M>(X, N, cfg, m)) st |= ST_WARN_MEAN_NOCONV;
    T C[D * D];
    cov_sigma_points<T, M>(m, X, N, C);
    for (int s = 0; s < S; ++s) mu[s] = m[s];
    for (int i = 0; i < D * D; ++i) sigma[i] = C[i] + R[i];
    return st;
}

// ukf::applyDelta(delta): resample around mu+delta.
template <class T, class M>
inline bool ukf_apply_delta(T* mu, T* sigma, const T* delta) {
    constexpr int S = M::S, D = M::D, N = 2 * D + 1;
    T X[N * S];
    if (!generate_sigma_points<T, M>(mu, delta, sigma, X)) return false;
    for (int s = 0; s < S; ++s) mu[s] = X[s];
    cov_sigma_points<T, M>(mu, X, N, sigma);
    return true;
}

// ukf::update(z, h, Q, mtest).  MZ::D = m.  Q row-major m x m.
template <class T, class M, class MZ, class H>
inline uint32_t ukf_update(T* mu, T* sigma, const T* z, H h, const T* Q, const Config& cfg) {
    constexpr int S = M::S, D = M::D, N = 2 * D + 1, SZ = MZ::S, m = MZ::D;
    T X[N * S];
    if (!generate_sigma_points<T, M>(mu, nullptr, sigma, X)) return ST_ERR_CHOLESKY;
    T Z[N * SZ];
    for (int i = 0; i < N; ++i) h(X + i * S, Z + i * SZ);
    uint32_t st = ST_OK;
    T meanZ[SZ];
    if (!mean_sigma_points<T, MZ>(Z, N, cfg, meanZ)) st |= ST_WARN_MEAN_NOCONV;
    T Sm[m * m];
    cov_sigma_points<T, MZ>(meanZ, Z, N, Sm);
    for (int i = 0; i < m * m; ++i) Sm[i] += Q[i];
    T Cxz[D * m];
    cross_cov_sigma_points<T, M, MZ>(mu, meanZ, X, Z, N, Cxz);
    T Si[m * m];
    inverse_small(Sm, m, Si);
    T K[D * m];
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < m; ++c) {
            T a = T(0);
            for (int k = 0; k < m; ++k) a += Cxz[r * m + k] * Si[k * m + c];
            K[r * m + c] = a;
        }
    T innov[m];
    MZ::boxminus(z, meanZ, innov);
    T maha = T(0);
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < m; ++c) maha += innov[r] * Si[r * m + c] * innov[c];
    if (cfg.gate_chi2 >= 0.0 && !(maha <= T(cfg.gate_chi2))) return st | ST_REJECTED_GATE;
    // sigma_ -= K * S * K^T  (Eigen: (K*S)*K^T)
    T KS[D * m];
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < m; ++c) {
            T a = T(0);
            for (int k = 0; k < m; ++k) a += K[r * m + k] * Sm[k * m + c];
            KS[r * m + c] = a;
        }
    T sig2[D * D];
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
            T a = T(0);
            for (int k = 0; k < m; ++k) a += KS[r * m + k] * K[c * m + k];
            sig2[r * D + c] = sigma[r * D + c] - a;
        }
    T delta[D];
    for (int r = 0; r < D; ++r) {
        T a = T(0);
        for (int k = 0; k < m; ++k) a += K[r * m + k] * innov[k];
        delta[r] = a;
    }
    T mu2[S];
    for (int s = 0; s < S; ++s) mu2[s] = mu[s];
    if (!ukf_apply_delta<T, M>(mu2, sig2, delta)) return st | ST_ERR_CHOLESKY;
    for (int s = 0; s < S; ++s) mu[s] = mu2[s];
    for (int i = 0; i < D * D; ++i) sigma[i] = sig2[i];
    return st;
}

// ---------------------------------------------------------------------------------
// Time gate -- UnscentedKalmanFilter.hpp:83-125
// ---------------------------------------------------------------------------------
// predictionStep(delta_t): returns ST_OK when predictionStepImpl must run.
inline uint32_t gate_dt(double dt, double min_dt, double max_dt) {
    if (dt < 0.0) return ST_ERR_NEG_DT;
    if (dt <= min_dt) return ST_SKIPPED_SMALL_DT;
    if (dt > max_dt) return ST_ERR_DT_TOO_LARGE;
    return ST_OK;
}

// predictionStepFromSampleTime(ts): int64 microseconds (base::Time); *last == 0 is isNull().
inline uint32_t gate_timestamp(int64_t ts_us, int64_t* last_us, double min_dt, double max_dt, double* dt_out) {
    if (*last_us == 0) {
        *last_us = ts_us;
        *dt_out = 0.0;
        return ST_SKIPPED_FIRST_TS;
    }
    const double dt = double(ts_us - *last_us) / 1000000.0;  // base::Time::toSeconds()
    if (dt > min_dt) *last_us = ts_us;
    *dt_out = dt;
    return gate_dt(dt, min_dt, max_dt);
}

template <class T> inline bool all_finite(const T* v, int n) {
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(v[i])) return false;
    return true;
}

// rot * B * rot^T for a 3x3 diagonal block at (o,o) of a DxD matrix (Eigen: (rot*B)*rot^T).
template <class T, int D> inline void rotate_block(const T* rot, const T* src, T* dst, int o) {
    T tmp[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            T a = T(0);
            for (int k = 0; k < 3; ++k) a += rot[r * 3 + k] * src[(o + k) * D + (o + c)];
            tmp[r * 3 + c] = a;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            T a = T(0);
            for (int k = 0; k < 3; ++k) a += tmp[r * 3 + k] * rot[c * 3 + k];
            dst[(o + r) * D + (o + c)] = a;
        }
}

// ---------------------------------------------------------------------------------
// PoseUKF -- pose_with_velocity/PoseUKF.cpp
// ---------------------------------------------------------------------------------
// processModel (PoseUKF.cpp:75-83) / processModelWithAcceleration (:88-97); acc == nullptr
// selects the former.
template <class T> inline void pose_process(T* x, const T* acc, T dt) {
    using M = PoseManifold<T>;
    if (acc)
        for (int k = 0; k < 3; ++k) x[M::V + k] += dt * acc[k];  // velocity.boxplus(acc, dt)
    T rv[3];
    quat_rotate(x + M::Q, x + M::V, rv);                         // orientation * velocity
    for (int k = 0; k < 3; ++k) x[M::P + k] += dt * rv[k];       // position.boxplus(.., dt)
    T rw[3];
    quat_rotate(x + M::Q, x + M::W, rw);                         // orientation * angular_velocity
    so3_boxplus(x + M::Q, rw, dt);                               // orientation.boxplus(.., dt)
}

// PoseUKF::predictionStepImpl (PoseUKF.cpp:180-196).  acc_mu: 3 values, non-finite or nullptr ->
// constant-velocity branch.  Quirk kept: on the acceleration branch the process noise is the raw
// process_noise_cov with block(6,6) = 2*acc_cov, neither rotated nor scaled by delta (:190-192).
template <class T>
inline uint32_t pose_predict(T* mu, T* sigma, const T* process_noise_cov, const T* acc_mu, const T* acc_cov,
                             T delta, const Config& cfg) {
    using M = PoseManifold<T>;
    constexpr int D = M::D;
    T R[D * D];
    for (int i = 0; i < D * D; ++i) R[i] = process_noise_cov[i];
    if (acc_mu && all_finite(acc_mu, 3)) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[(6 + r) * D + (6 + c)] = T(2) * acc_cov[r * 3 + c];
        T a[3] = {acc_mu[0], acc_mu[1], acc_mu[2]};
        return ukf_predict<T, M>(mu, sigma, [&](T* x) { pose_process<T>(x, a, delta); }, R, cfg);
    }
    T rot[9];
    quat_to_matrix(mu + M::Q, rot);
    rotate_block<T, D>(rot, process_noise_cov, R, 0);
    rotate_block<T, D>(rot, process_noise_cov, R, 3);
    for (int i = 0; i < D * D; ++i) R[i] = delta * R[i];
    return ukf_predict<T, M>(mu, sigma, [&](T* x) { pose_process<T>(x, nullptr, delta); }, R, cfg);
}

// measurement dimension and sub-state selection for the 8 vector-valued Pose models
// (PoseUKF.cpp:7-26, 35-69).  Returns m, fills idx with stored-state indices.
inline int pose_measurement_select(int model, int idx[3]) {
    switch (model) {
        case MEAS_POS3: idx[0] = 0; idx[1] = 1; idx[2] = 2; return 3;
        case MEAS_POS_XY: idx[0] = 0; idx[1] = 1; return 2;
        case MEAS_POS_Z: idx[0] = 2; return 1;
        case MEAS_VEL3: idx[0] = 7; idx[1] = 8; idx[2] = 9; return 3;
        case MEAS_VEL_XY: idx[0] = 7; idx[1] = 8; return 2;
        case MEAS_VEL_Z: idx[0] = 9; return 1;
        case MEAS_XVEL_YAWVEL: idx[0] = 7; idx[1] = 12; return 2;
        case MEAS_ANGVEL3: idx[0] = 10; idx[1] = 11; idx[2] = 12; return 3;
        default: return 0;
    }
}

// PoseUKF::integrateMeasurement overloads (PoseUKF.cpp:112-173).  z: 3 values (first m used; for
// MEAS_ORIENT_SO3 the axis-angle vector, converted with SO3::exp as in :135).  Q: row-major 3x3,
// leading m x m block used.
template <class T>
inline uint32_t pose_update(T* mu, T* sigma, int model, const T* z, const T* Q3, const Config& cfg) {
    using M = PoseManifold<T>;
    if (model == MEAS_ORIENT_SO3) {
        T zq[4];
        so3_exp(z, T(1), zq);
        return ukf_update<T, M, SO3Manifold<T>>(
            mu, sigma, zq, [](const T* x, T* zz) { for (int k = 0; k < 4; ++k) zz[k] = x[M::Q + k]; }, Q3, cfg);
    }
    int idx[3] = {0, 0, 0};
    const int m = pose_measurement_select(model, idx);
    T Qm[9];
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < m; ++c) Qm[r * m + c] = Q3[r * 3 + c];
    auto h = [&](const T* x, T* zz) { for (int k = 0; k < m; ++k) zz[k] = x[idx[k]]; };
    if (m == 3) return ukf_update<T, M, VectManifold<T, 3>>(mu, sigma, z, h, Qm, cfg);
    if (m == 2) return ukf_update<T, M, VectManifold<T, 2>>(mu, sigma, z, h, Qm, cfg);
    if (m == 1) return ukf_update<T, M, VectManifold<T, 1>>(mu, sigma, z, h, Qm, cfg);
    return ST_INACTIVE;
}

// ---------------------------------------------------------------------------------
// OrientationUKF -- orientation_estimator/OrientationUKF.cpp
// ---------------------------------------------------------------------------------
template <class T> struct OrientParams {
    T gyro_bias_tau, acc_bias_tau;
    T earth_rotation[3];  // (EARTHW cos lat, 0, EARTHW sin lat)  OrientationUKF.cpp:47
};

// GravitationalModel.hpp:16
inline double earth_angular_velocity() { return (2.0 * M_PI) / 86164.0; }

// processModel (OrientationUKF.cpp:12-32)
template <class T>
inline void orient_process(T* x, const T* acc, const T* omega, const OrientParams<T>& p, T dt) {
    using M = OrientManifold<T>;
    T t[3], av[3];
    for (int k = 0; k < 3; ++k) t[k] = omega[k] - x[M::BG + k];
    quat_rotate(x + M::Q, t, av);
    for (int k = 0; k < 3; ++k) av[k] -= p.earth_rotation[k];
    so3_boxplus(x + M::Q, av, dt);
    T a[3];
    for (int k = 0; k < 3; ++k) t[k] = acc[k] - x[M::BA + k];
    quat_rotate(x + M::Q, t, a);  // uses the already updated orientation (:22)
    a[2] -= x[M::G];
    for (int k = 0; k < 3; ++k) x[M::V + k] += dt * a[k];
    for (int k = 0; k < 3; ++k) {
        const T gd = (T(-1.0) / p.gyro_bias_tau) * x[M::BG + k];
        x[M::BG + k] += dt * gd;
    }
    for (int k = 0; k < 3; ++k) {
        const T ad = (T(-1.0) / p.acc_bias_tau) * x[M::BA + k];
        x[M::BA + k] += dt * ad;
    }
}

// OrientationUKF::predictionStepImpl (OrientationUKF.cpp:79-89): orientation and velocity noise
// blocks rotated, whole matrix scaled by delta^2.
template <class T>
inline uint32_t orient_predict(T* mu, T* sigma, const T* process_noise_cov, const T* acc, const T* omega,
                               const OrientParams<T>& p, T delta, const Config& cfg) {
    using M = OrientManifold<T>;
    constexpr int D = M::D;
    T rot[9];
    quat_to_matrix(mu + M::Q, rot);
    T R[D * D];
    for (int i = 0; i < D * D; ++i) R[i] = process_noise_cov[i];
    rotate_block<T, D>(rot, process_noise_cov, R, 0);
    rotate_block<T, D>(rot, process_noise_cov, R, 3);
    const T d2 = T(std::pow(double(delta), 2.));
    for (int i = 0; i < D * D; ++i) R[i] = d2 * R[i];
    T a[3] = {acc[0], acc[1], acc[2]}, w[3] = {omega[0], omega[1], omega[2]};
    return ukf_predict<T, M>(mu, sigma, [&](T* x) { orient_process<T>(x, a, w, p, delta); }, R, cfg);
}

// integrateMeasurement(VelocityMeasurement) (OrientationUKF.cpp:65-72): h = q^-1 * v (:34-39),
// preceded by checkMeasurment (:67).
template <class T>
inline uint32_t orient_update(T* mu, T* sigma, const T* z, const T* Q3, const Config& cfg) {
    using M = OrientManifold<T>;
    if (!all_finite(z, 3) || !all_finite(Q3, 9)) return ST_ERR_NONFINITE_MEAS;
    auto h = [](const T* x, T* zz) {
        T qi[4];
        quat_inverse(x + M::Q, qi);
        quat_rotate(qi, x + M::V, zz);
    };
    return ukf_update<T, M, VectManifold<T, 3>>(mu, sigma, z, h, Q3, cfg);
}

// OrientationUKF::getRotationRate (OrientationUKF.cpp:74-77)
template <class T>
inline void orient_rotation_rate(const T* mu, const T* omega, const T* earth_rotation, T* out) {
    using M = OrientManifold<T>;
    T qi[4], e[3];
    quat_inverse(mu + M::Q, qi);
    quat_rotate(qi, earth_rotation, e);
    for (int k = 0; k < 3; ++k) out[k] = omega[k] - mu[M::BG + k] - e[k];
}

}  // namespace ukf_oracle
