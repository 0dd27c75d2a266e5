// ukf_device.hpp -- gfx950 device-side math for the batched UKF engine.
//
// SO(3) exp/log, quaternion algebra, compound-manifold boxplus/boxminus and the process /
// measurement models, written for register-resident fixed-size arrays (every loop is fully
// unrolled so that no array is indexed at run time and nothing spills to scratch).
//
// Behaviour follows the reference call sites (paths relative to /root/reference/src):
//   pose_with_velocity/PoseUKF.cpp:75-97        processModel / processModelWithAcceleration
//   pose_with_velocity/PoseUKF.cpp:7-69         measurement models
//   orientation_estimator/OrientationUKF.cpp:12-39   processModel / velocityMeasurementModel
// and the MTK semantics recalled in SURVEY.md Appendix A (SO3::exp/log with the
// cos_sinc_sqrt Taylor switch, right-multiplying boxplus, atan-based log).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define UKFB_DEV __device__ __forceinline__

namespace ukfb {

template <class T> struct Num;
template <> struct Num<double> {
    // sqrt(sqrt(DBL_EPSILON)) = 2^-13
    static constexpr double taylor_n_bound = 0.0001220703125;
    static constexpr double mtk_tol = 1e-11;
};
template <> struct Num<float> {
    // sqrt(sqrt(FLT_EPSILON)) = 2^-5.75
    static constexpr float taylor_n_bound = 0.018581361f;
    static constexpr float mtk_tol = 1e-5f;
};

UKFB_DEV double m_sqrt(double x) { return sqrt(x); }
UKFB_DEV float m_sqrt(float x) { return sqrtf(x); }
UKFB_DEV double m_atan(double x) { return atan(x); }
UKFB_DEV float m_atan(float x) { return atanf(x); }
UKFB_DEV void m_sincos(double x, double* s, double* c) { sincos(x, s, c); }
UKFB_DEV void m_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
UKFB_DEV bool m_finite(double x) { return isfinite(x); }
UKFB_DEV bool m_finite(float x) { return isfinite(x); }

enum { QX = 0, QY = 1, QZ = 2, QW = 3 };

// Wave-uniform votes straight from the compare mask (v_cmp into an SGPR pair + scalar test).  HIP's __any / __all go
// through an integer (v_cndmask 0/1, v_cmp_ne, s_cmp): two vector instructions more per vote, and the hot path votes
// about thirty times (every exp / log asks whether any lane needs its wide-angle path).
UKFB_DEV bool wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
UKFB_DEV bool wave_all(bool b) { return __builtin_amdgcn_ballot_w64(b) == __builtin_amdgcn_ballot_w64(true); }
// Lane masks straight from ONE compare (active lanes only, like a ballot).  Where the compared bool also steers selects or an
// exec-masked region, the compiler lowers ballot(b) by materialising b in a VGPR and comparing it again (v_cndmask 0/1 +
// v_cmp_ne: two 4-cycle instructions per vote, ~25 votes on the hot path); with the mask in hand the vote is a scalar test and
// the selects take the SGPR pair through inverse_ballot.  lanes_not_le: !(x <= bound), true for NaN as well.
UKFB_DEV unsigned long long lanes_not_le(double x, double bound) {
    unsigned long long m;
    asm("v_cmp_nge_f64_e64 %0, %1, %2" : "=s"(m) : "s"(bound), "v"(x));
    return m;
}
UKFB_DEV unsigned long long lanes_not_le(float x, float bound) {
    unsigned long long m;
    asm("v_cmp_nge_f32_e64 %0, %1, %2" : "=s"(m) : "s"(bound), "v"(x));
    return m;
}
UKFB_DEV unsigned long long lanes_gt(double x, double bound) {
    unsigned long long m;
    asm("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(m) : "s"(bound), "v"(x));
    return m;
}
UKFB_DEV unsigned long long lanes_gt(float x, float bound) {
    unsigned long long m;
    asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "s"(bound), "v"(x));
    return m;
}
UKFB_DEV bool lane_of(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// ---------------------------------------------------------------------------------------------
// Fast, division-free primitives for the tuned kernel (ukf_kernel16.hpp).  All are accurate to a
// few ulp of T, far inside the 1e-9 (f64) / 1e-4 (f32) parity budget, and replace the ocml
// sqrt / sincos / atan / IEEE-division expansions that dominated the first kernel's VALU time.
// ---------------------------------------------------------------------------------------------
// 1/sqrt(x) and 1/x from the hardware seeds.  v_rsq_f64 / v_rcp_f64 deliver ~2^-26..2^-27 (llvm refines
// them twice only to reach correct rounding), so ONE Newton step gives ~1e-15 relative, far inside the
// 1e-9 budget; the f32 seeds are already 1 ulp and are used as they are (budget 1e-4).
UKFB_DEV double fast_rsqrt(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * r), r, 1.0);
    return fma(0.5 * r, e, r);
}
// (a Newton step on the fp32 seeds does not move the fp32 engine's distance to the fp64 oracle: profiles/r03_f32_drift_attribution.txt)
UKFB_DEV float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
UKFB_DEV double fast_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
UKFB_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// -DUKFB_COUNTS (diagnostic build, tools/trip_counts.py): how often a wavefront takes the wide-angle paths of exp / log.
// One atomic per wavefront and call; the product build defines nothing and the macro vanishes.
#if defined(UKFB_COUNTS)
enum { DBG_EXP = 0, DBG_EXP_LARGE, DBG_EXP_BIG, DBG_LOGN, DBG_LOGN_BIG, DBG_LOG, DBG_LOG_BIG, DBG_N };
static __device__ unsigned long long ukfb_dbg[DBG_N];
#define UKFB_DBG_COUNT(i) do { if (threadIdx.x == 0) atomicAdd(&ukfb_dbg[i], 1ull); } while (0)
#else
#define UKFB_DBG_COUNT(i) do { } while (0)
#endif

// Near-minimax polynomials (Chebyshev fits computed with mpmath at 50 digits; max abs error quoted) for
//   cos(sqrt(y)), sin(sqrt(y))/sqrt(y) on 0 <= y <= 0.62   and   atan(sqrt(u))/sqrt(u) on 0 <= u <= 0.07.
// Three terms shorter than the Taylor series of the same accuracy; coefficients c0..cn, Horner form.
template <class T> struct Poly;
template <> struct Poly<double> {
    static constexpr int NC = 6, NS = 6, NA = 8;
    static constexpr double COS[NC + 1] = {0.999999999999999951, -0.499999999999992276, 0.0416666666664673017,
                                           -0.00138888888695903017, 0.000024801578404008521, -2.7555212612292537e-7,
                                           2.06292036185570243e-9};                                    // 4.9e-17
    static constexpr double SINC[NS + 1] = {0.999999999999999997, -0.166666666666666151, 0.00833333333332002843,
                                            -0.000198412698283910677, 2.75573132865827307e-6, -2.50507027741921191e-8,
                                            1.58939017348049994e-10};                                  // 3.3e-18
    static constexpr double ATAN[NA + 1] = {0.999999999999999988, -0.333333333333304925, 0.199999999989156351,
                                            -0.142857141260715088, 0.111110993100731619, -0.0909041735889935631,
                                            0.0768020126708874696, -0.0649096002386012782, 0.0446824258353296505};  // 1.2e-17
    static constexpr double Y_SMALL = 0.62, U_SMALL = 0.07;
};
template <> struct Poly<float> {
    static constexpr int NC = 4, NS = 3, NA = 3;
    static constexpr float COS[NC + 1] = {1.0f, -0.4999999961f, 0.04166661593f, -0.001388659487f, 2.437769412e-5f};  // 4.9e-11
    static constexpr float SINC[NS + 1] = {0.9999999969f, -0.1666665043f, 0.008332022568f, -0.0001950219494f};         // 3.1e-9
    static constexpr float ATAN[NA + 1] = {0.9999999814f, -0.3333247988f, 0.1993845199f, -0.1284461816f};              // 1.9e-8
    static constexpr float Y_SMALL = 0.62f, U_SMALL = 0.07f;
};
template <class T, int N> UKFB_DEV T horner(const T (&c)[N + 1], T x) {
    T r = c[N];
#pragma unroll
    for (int k = N - 1; k >= 0; --k) r = fma(r, x, c[k]);
    return r;
}
// fp32: the coefficients as instruction literals (v_fmaak_f32, 2 issue cycles).  Read through the array they arrive by scalar
// loads and every Horner step carries an SGPR operand, which issues at the 4-cycle rate on gfx950; in fp64 the
// SGPR pairs are the cheapest form (a 64-bit literal needs two s_mov per use).
#define UKFB_HORNER_LIT(ARR, N, x, r)                                   \
    do {                                                                \
        constexpr float c_[5] = {ARR[0], ARR[1], ARR[2], ARR[3], (N >= 4 ? ARR[N >= 4 ? 4 : 0] : 0.f)}; \
        r = c_[N];                                                      \
        if constexpr (N >= 4) r = fma(r, x, c_[3]);                     \
        r = fma(r, x, c_[2]);                                           \
        r = fma(r, x, c_[1]);                                           \
        r = fma(r, x, c_[0]);                                           \
    } while (0)
template <class T> UKFB_DEV void poly_cos_sinc(T y, T& c, T& s) {
    if constexpr (sizeof(T) == 4) {
        static_assert(Poly<float>::NC == 4 && Poly<float>::NS == 3, "literal Horner forms");
        UKFB_HORNER_LIT(Poly<float>::COS, 4, y, c);
        UKFB_HORNER_LIT(Poly<float>::SINC, 3, y, s);
    } else {
        c = horner<T, Poly<T>::NC>(Poly<T>::COS, y);
        s = horner<T, Poly<T>::NS>(Poly<T>::SINC, y);
    }
}
template <class T> UKFB_DEV T poly_atan_ratio(T u) {
    if constexpr (sizeof(T) == 4) {
        static_assert(Poly<float>::NA == 3, "literal Horner form");
        T r;
        UKFB_HORNER_LIT(Poly<float>::ATAN, 3, u, r);
        return r;
    } else {
        return horner<T, Poly<T>::NA>(Poly<T>::ATAN, u);
    }
}

template <class T> struct TwoPi;
template <> struct TwoPi<double> {
    static constexpr double hi = 6.283185307179586, lo = 2.4492935982947064e-16, inv = 0.15915494309189535;
};
template <> struct TwoPi<float> {
    static constexpr float hi = 6.28318548f, lo = -1.74845553e-07f, inv = 0.159154937f;
};
UKFB_DEV double m_rint(double x) { return __builtin_rint(x); }
UKFB_DEV float m_rint(float x) { return __builtin_rintf(x); }
UKFB_DEV double m_abs(double x) { return __builtin_fabs(x); }
UKFB_DEV float m_abs(float x) { return __builtin_fabsf(x); }

// (cos(sqrt(y)), sin(sqrt(y))/sqrt(y)) for y >= 0 with no sqrt / sincos / division on the common path:
// Taylor polynomial in y up to (pi/4)^2; one angle doubling up to (pi/2)^2; beyond that the angle is
// reduced modulo 2 pi (two-term Cody-Waite, wave-uniform branch) and doubled twice.  No ocml calls:
// their inlined large-argument paths cost ~100 live registers per call site.
template <class T> UKFB_DEV void cos_sinc_fast(T y, T& c, T& s) {
    const bool small = y <= Poly<T>::Y_SMALL;
    const bool any_large = wave_any(!small);   // wave-uniform: the doubling steps below are skipped on the common path
    const bool big = !(y <= T(4) * Poly<T>::Y_SMALL);
    T yy = y;
    T ratio = T(1);
    UKFB_DBG_COUNT(DBG_EXP);
    if (any_large) {
        UKFB_DBG_COUNT(DBG_EXP_LARGE);
        yy = small ? y : T(0.25) * y;
        if (wave_any(big)) {
            UKFB_DBG_COUNT(DBG_EXP_BIG);
            const T rs = fast_rsqrt(big ? y : T(1));         // 1/x
            const T x = y * rs;
            const T k = m_rint(x * TwoPi<T>::inv);
            T xr = fma(-k, TwoPi<T>::hi, x);
            xr = fma(-k, TwoPi<T>::lo, xr);                  // |xr| <= pi
            yy = big ? (xr * xr) * T(0.0625) : yy;           // (xr/4)^2 <= (pi/4)^2
            ratio = big ? xr * rs : T(1);                    // sin(x)/x = sinc(xr) * xr/x
        }
    }
    T c1, s1;
    poly_cos_sinc(yy, c1, s1);
    c = c1;
    s = s1;
    if (any_large) {
        const T c2 = fma(T(2) * c1, c1, T(-1)), s2 = s1 * c1;    // angle x2
        const T c4 = fma(T(2) * c2, c2, T(-1)), s4 = s2 * c2;    // angle x4
        c = small ? c1 : (big ? c4 : c2);
        s = small ? s1 : (big ? s4 * ratio : s2);
    }
}

// exp / log with the fast primitives (same maps as so3_exp / so3_log below)
template <class T> UKFB_DEV void so3_exp_fast(const T (&v)[3], T scale, T (&q)[4]) {
    const T s = scale * T(0.5);
    const T n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    T c, sc;
    cos_sinc_fast(s * s * n2, c, sc);
    const T mult = sc * s;
    q[0] = mult * v[0]; q[1] = mult * v[1]; q[2] = mult * v[2]; q[3] = c;
}
// 2 atan(|vec|/w)/|vec| * vec = (2/w) * [atan(t)/t] * vec with t^2 = |vec|^2 / w^2.  Beyond ~30 degrees
// (wave-uniform branch) the half angle phi = atan(|vec|/|w|) is halved three times with
// tan(phi/2) = t / (1 + sqrt(1 + t^2)), which brings it below atan(0.2); sign of w restores MTK's
// plus/minus periodicity.  No ocml calls.
template <class T> UKFB_DEV void so3_log_fast(const T (&q)[4], T (&r)[3]) {
    const T v2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const T w = q[3];
    const T rw = fast_rcp(w);
    const T u = v2 * rw * rw;
    T s = T(2) * rw * poly_atan_ratio(u);
    const unsigned long long bigm = lanes_not_le(u, Poly<T>::U_SMALL);
    UKFB_DBG_COUNT(DBG_LOG);
    if (bigm != 0ull) {
        UKFB_DBG_COUNT(DBG_LOG_BIG);
        const bool big = lane_of(bigm);
        const T n2 = fma(w, w, v2);
        const T n = n2 * fast_rsqrt(n2);
        const T r1 = fast_rcp(m_abs(w) + n);
        const T t1 = v2 * r1 * r1;                       // tan^2(phi/2) <= 1
        const T q2 = T(1) + t1;
        const T r2 = fast_rcp(T(1) + q2 * fast_rsqrt(q2));
        const T t2 = t1 * r2 * r2;                       // tan^2(phi/4) <= 0.172
        const T q3 = T(1) + t2;
        const T r3 = fast_rcp(T(1) + q3 * fast_rsqrt(q3));
        const T t3 = t2 * r3 * r3;                       // tan^2(phi/8) <= 0.0396
        const T sb = T(16) * (r1 * r2) * r3 * poly_atan_ratio(t3);
        s = big ? ((w < T(0)) ? -sb : sb) : s;
    }
    r[0] = s * q[0]; r[1] = s * q[1]; r[2] = s * q[2];
}


// The same map for a quaternion whose norm `nrm` is known (products of the mean's orientation with unit
// exponentials all have the mean's norm; MTK's atan(|vec| / w) is scale invariant, so the norm must be honoured,
// not assumed to be 1).  With the norm at hand the first half-angle step tan(phi/2) = |vec| / (w + |q|) costs one
// addition, so the polynomial covers rotation angles up to ~60 degrees (tan^2(phi/2) <= 0.07) instead of ~30:
//   log q = 4 atan(t1)/|vec| vec = 4 [atan(t1)/t1] / (w + |q|) vec.
// Beyond that (wave-uniform branch) two more half-angle steps; w < 0 goes through |w| and MTK's sign.
template <class T> UKFB_DEV void so3_log_fast_n(const T (&q)[4], T nrm, T (&r)[3]) {
    const T v2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const T w = q[3];
    const T r1 = fast_rcp(w + nrm);
    const T t1 = v2 * r1 * r1;                           // tan^2(phi/2)
    T s = T(4) * r1 * poly_atan_ratio(t1);
    const unsigned long long bigm = lanes_not_le(t1, Poly<T>::U_SMALL);   // also w + |q| <= 0 (angle >= pi) and NaN
    UKFB_DBG_COUNT(DBG_LOGN);
    if (bigm != 0ull) {
        UKFB_DBG_COUNT(DBG_LOGN_BIG);
        const bool big = lane_of(bigm);
        const T r1b = fast_rcp(m_abs(w) + nrm);
        const T t1b = v2 * r1b * r1b;                    // tan^2(phi/2) <= 1 with phi = atan(|vec| / |w|)
        const T q2 = T(1) + t1b;
        const T r2 = fast_rcp(T(1) + q2 * fast_rsqrt(q2));
        const T t2 = t1b * r2 * r2;                      // tan^2(phi/4) <= 0.172
        const T q3 = T(1) + t2;
        const T r3 = fast_rcp(T(1) + q3 * fast_rsqrt(q3));
        const T t3 = t2 * r3 * r3;                       // tan^2(phi/8) <= 0.0396
        const T sb = T(16) * (r1b * r2) * r3 * poly_atan_ratio(t3);
        s = big ? ((w < T(0)) ? -sb : sb) : s;
    }
    r[0] = s * q[0]; r[1] = s * q[1]; r[2] = s * q[2];
}

// Two logarithms at once (the + and - sigma point of a lane against the same reference): the same arithmetic as two calls of
// so3_log_fast_n, but ONE wave-uniform vote for the wide-angle path of both, so that the common path is straight-line code with
// two independent dependency chains for the scheduler to interleave (a branch between the two calls keeps them apart; the
// Horner chain of one logarithm alone leaves the fp64 pipeline waiting on itself).
template <class T> UKFB_DEV void so3_log_fast_n2(const T (&qa)[4], const T (&qb)[4], T nrm, T (&ra)[3], T (&rb)[3]) {
    const T v2a = qa[0] * qa[0] + qa[1] * qa[1] + qa[2] * qa[2];
    const T v2b = qb[0] * qb[0] + qb[1] * qb[1] + qb[2] * qb[2];
    const T r1a = fast_rcp(qa[3] + nrm), r1b = fast_rcp(qb[3] + nrm);
    const T t1a = v2a * r1a * r1a, t1b = v2b * r1b * r1b;
    T sa = T(4) * r1a * poly_atan_ratio(t1a);
    T sb = T(4) * r1b * poly_atan_ratio(t1b);
    const unsigned long long biga = lanes_not_le(t1a, Poly<T>::U_SMALL), bigb = lanes_not_le(t1b, Poly<T>::U_SMALL);
    UKFB_DBG_COUNT(DBG_LOGN);
    UKFB_DBG_COUNT(DBG_LOGN);
    if ((biga | bigb) != 0ull) {
        UKFB_DBG_COUNT(DBG_LOGN_BIG);
        UKFB_DBG_COUNT(DBG_LOGN_BIG);
        const auto wide = [&](const T v2, const T w) {
            const T r1w = fast_rcp(m_abs(w) + nrm);
            const T t1w = v2 * r1w * r1w;
            const T q2 = T(1) + t1w;
            const T r2 = fast_rcp(T(1) + q2 * fast_rsqrt(q2));
            const T t2 = t1w * r2 * r2;
            const T q3 = T(1) + t2;
            const T r3 = fast_rcp(T(1) + q3 * fast_rsqrt(q3));
            const T t3 = t2 * r3 * r3;
            const T sw = T(16) * (r1w * r2) * r3 * poly_atan_ratio(t3);
            return (w < T(0)) ? -sw : sw;
        };
        const T wa = wide(v2a, qa[3]), wb = wide(v2b, qb[3]);
        sa = lane_of(biga) ? wa : sa;
        sb = lane_of(bigb) ? wb : sb;
    }
    ra[0] = sa * qa[0]; ra[1] = sa * qa[1]; ra[2] = sa * qa[2];
    rb[0] = sb * qb[0]; rb[1] = sb * qb[1]; rb[2] = sb * qb[2];
}

// Rotation delta after a SMALL move of the reference: with d = log(conj(r) q) already known and r' = r exp(a),
//   log(conj(r') q) = log(exp(-a) exp(d)) = d - a - (a x d)/2 - c(|d|^2) (d (d.a) - a |d|^2) + (a (a.d) - d |a|^2)/12 + ...,
//   c(t) = 1/t - cot(sqrt(t)/2) / (2 sqrt(t)) = 1/12 + t/720 + t^2/30240 + ...   (inverse left Jacobian of SO(3)),
// i.e. one explicit Euler step of d'(s) = -Jl^-1(d(s)) a (exact in d) plus the leading second-order term in a.  For
// |a| <= 1e-6 and |d| <= 1.5 rad the remainder is below 4e-14 (checked against 40-digit arithmetic in
// tests/test_oracle_mpmath.py); the caller guarantees both bounds and takes the full logarithm otherwise.
// 34 operations instead of ~85 (quaternion product + logarithm).
template <class T> UKFB_DEV void so3_rebase_small(const T (&d)[3], const T (&a)[3], T a2, T (&r)[3]) {
    const T t = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const T da = d[0] * a[0] + d[1] * a[1] + d[2] * a[2];
    T c = T(691. / 1307674368000.);
    c = fma(c, t, T(1. / 47900160.));
    c = fma(c, t, T(1. / 1209600.));
    c = fma(c, t, T(1. / 30240.));
    c = fma(c, t, T(1. / 720.));
    c = fma(c, t, T(1. / 12.));
    // d (1 - c d.a - |a|^2 / 12) - a (1 - c |d|^2 - d.a / 12) - (a x d) / 2
    const T s1 = fma(T(-1. / 12.), a2, fma(-c, da, T(1))), s2 = fma(T(-1. / 12.), da, fma(-c, t, T(1)));
    const T x0 = a[1] * d[2] - a[2] * d[1], x1 = a[2] * d[0] - a[0] * d[2], x2 = a[0] * d[1] - a[1] * d[0];
    r[0] = fma(T(-0.5), x0, fma(d[0], s1, -(a[0] * s2)));
    r[1] = fma(T(-0.5), x1, fma(d[1], s1, -(a[1] * s2)));
    r[2] = fma(T(-0.5), x2, fma(d[2], s1, -(a[2] * s2)));
}

// MTK cos_sinc_sqrt: (cos(sqrt(x2)), sin(sqrt(x2))/sqrt(x2)), three Taylor pairs below eps^(1/4).
template <class T> UKFB_DEV void cos_sinc_sqrt(T x2, T& c, T& s) {
    T cosi = T(1), sinc = T(1);
    T term = T(-0.5) * x2;
    cosi += term; term *= T(1 / 3.); sinc += term; term *= T(-1 / 4.) * x2;
    cosi += term; term *= T(1 / 5.); sinc += term; term *= T(-1 / 6.) * x2;
    cosi += term; term *= T(1 / 7.); sinc += term;
    if (x2 >= Num<T>::taylor_n_bound) {
        const T x = m_sqrt(x2);
        T sn, cs;
        m_sincos(x, &sn, &cs);
        cosi = cs;
        sinc = sn / x;
    }
    c = cosi;
    s = sinc;
}

template <class T> UKFB_DEV void quat_mul(const T (&a)[4], const T (&b)[4], T (&r)[4]) {
    const T w = a[QW] * b[QW] - a[QX] * b[QX] - a[QY] * b[QY] - a[QZ] * b[QZ];
    const T x = a[QW] * b[QX] + a[QX] * b[QW] + a[QY] * b[QZ] - a[QZ] * b[QY];
    const T y = a[QW] * b[QY] + a[QY] * b[QW] + a[QZ] * b[QX] - a[QX] * b[QZ];
    const T z = a[QW] * b[QZ] + a[QZ] * b[QW] + a[QX] * b[QY] - a[QY] * b[QX];
    r[QX] = x; r[QY] = y; r[QZ] = z; r[QW] = w;
}

// Eigen _transformVector: v + w*uv + vec x uv, uv = 2 (vec x v)
template <class T> UKFB_DEV void quat_rotate(const T (&q)[4], const T (&v)[3], T (&r)[3]) {
    const T ux = q[QY] * v[2] - q[QZ] * v[1];   // uv / 2
    const T uy = q[QZ] * v[0] - q[QX] * v[2];
    const T uz = q[QX] * v[1] - q[QY] * v[0];
    const T cx = q[QY] * uz - q[QZ] * uy;
    const T cy = q[QZ] * ux - q[QX] * uz;
    const T cz = q[QX] * uy - q[QY] * ux;
    r[0] = fma(T(2), fma(q[QW], ux, cx), v[0]);
    r[1] = fma(T(2), fma(q[QW], uy, cy), v[1]);
    r[2] = fma(T(2), fma(q[QW], uz, cz), v[2]);
}

// Eigen toRotationMatrix, row-major
template <class T> UKFB_DEV void quat_to_matrix(const T (&q)[4], T (&R)[9]) {
    const T tx = T(2) * q[QX], ty = T(2) * q[QY], tz = T(2) * q[QZ];
    const T twx = tx * q[QW], twy = ty * q[QW], twz = tz * q[QW];
    const T txx = tx * q[QX], txy = ty * q[QX], txz = tz * q[QX];
    const T tyy = ty * q[QY], tyz = tz * q[QY], tzz = tz * q[QZ];
    R[0] = T(1) - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
    R[3] = txy + twz;          R[4] = T(1) - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = T(1) - (txx + tyy);
}

// MTK SO3::exp(v, scale)
template <class T> UKFB_DEV void so3_exp(const T (&v)[3], T scale, T (&q)[4]) {
    const T s = scale * T(0.5);
    const T n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    T c, sc;
    cos_sinc_sqrt(s * s * n2, c, sc);
    const T mult = sc * s;
    q[QX] = mult * v[0]; q[QY] = mult * v[1]; q[QZ] = mult * v[2]; q[QW] = c;
}

// MTK SO3::log(q): 2 atan(|vec| / w) / |vec| * vec  (plus/minus periodic)
template <class T> UKFB_DEV void so3_log(const T (&q)[4], T (&r)[3]) {
    T nv = m_sqrt(q[QX] * q[QX] + q[QY] * q[QY] + q[QZ] * q[QZ]);
    nv = nv < Num<T>::mtk_tol ? Num<T>::mtk_tol : nv;
    const T s = T(2) / nv * m_atan(nv / q[QW]);
    r[0] = s * q[QX]; r[1] = s * q[QY]; r[2] = s * q[QZ];
}

// q <- q * exp(v, scale)
template <class T> UKFB_DEV void so3_boxplus(T (&q)[4], const T (&v)[3], T scale) {
    T d[4], r[4];
    so3_exp_fast(v, scale, d);
    quat_mul(q, d, r);
    q[0] = r[0]; q[1] = r[1]; q[2] = r[2]; q[3] = r[3];
}

// log(other^* * q)
template <class T> UKFB_DEV void so3_boxminus(const T (&q)[4], const T (&other)[4], T (&r)[3]) {
    const T oc[4] = {-other[QX], -other[QY], -other[QZ], other[QW]};
    T d[4];
    quat_mul(oc, q, d);
    so3_log_fast(d, r);
}

// ---------------------------------------------------------------------------------------------
// Per-filter inputs of the process models (wave-resident scalars)
// ---------------------------------------------------------------------------------------------
template <class T> struct ProcIn {
    T dt;
    T a[3];       // Pose: acc.mu ; Orient: acceleration.mu
    T adt[3];     // Pose (tuned kernel): use_acc ? acc.mu * dt : 0, the velocity increment of every sigma point
    T w[3];       // Orient: rotation_rate.mu
    bool use_acc; // Pose: acc.mu.allFinite()   (PoseUKF.cpp:188)
    T ninv_tau_g, ninv_tau_a;  // Orient: -1/tau
    T earth[3];
};

// ---------------------------------------------------------------------------------------------
// PoseWithVelocity (PoseWithVelocity.hpp:18-23): p[0..2] q[3..6] v[7..9] w[10..12]
// ---------------------------------------------------------------------------------------------
template <class T> struct PoseM {
    template <class U> using rebind = PoseM<U>;   // the same manifold over another scalar (wide-arithmetic kernels)
    static constexpr int S = 13, D = 12, MODEL = 0;
    enum { P = 0, Q = 3, V = 7, W = 10 };

    static UKFB_DEV void boxplus(T (&x)[13], const T (&d)[12]) {
        x[0] += d[0]; x[1] += d[1]; x[2] += d[2];
        T q[4] = {x[3], x[4], x[5], x[6]};
        const T dv[3] = {d[3], d[4], d[5]};
        so3_boxplus(q, dv, T(1));
        x[3] = q[0]; x[4] = q[1]; x[5] = q[2]; x[6] = q[3];
#pragma unroll
        for (int k = 0; k < 6; ++k) x[7 + k] += d[6 + k];
    }
    static UKFB_DEV void boxminus(const T (&x)[13], const T (&y)[13], T (&d)[12]) {
        d[0] = x[0] - y[0]; d[1] = x[1] - y[1]; d[2] = x[2] - y[2];
        const T qx[4] = {x[3], x[4], x[5], x[6]}, qy[4] = {y[3], y[4], y[5], y[6]};
        T r[3];
        so3_boxminus(qx, qy, r);
        d[3] = r[0]; d[4] = r[1]; d[5] = r[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[6 + k] = x[7 + k] - y[7 + k];
    }
    // processModel / processModelWithAcceleration (PoseUKF.cpp:75-97)
    static UKFB_DEV void process(T (&x)[13], const ProcIn<T>& in) {
        if (in.use_acc) {
            x[7] += in.dt * in.a[0]; x[8] += in.dt * in.a[1]; x[9] += in.dt * in.a[2];
        }
        T q[4] = {x[3], x[4], x[5], x[6]};
        const T v[3] = {x[7], x[8], x[9]}, w[3] = {x[10], x[11], x[12]};
        T rv[3], rw[3];
        quat_rotate(q, v, rv);
        x[0] += in.dt * rv[0]; x[1] += in.dt * rv[1]; x[2] += in.dt * rv[2];
        quat_rotate(q, w, rw);
        so3_boxplus(q, rw, in.dt);
        x[3] = q[0]; x[4] = q[1]; x[5] = q[2]; x[6] = q[3];
    }
    static UKFB_DEV void orientation(const T (&x)[13], T (&q)[4]) {
        q[0] = x[3]; q[1] = x[4]; q[2] = x[5]; q[3] = x[6];
    }
    // measurement models (PoseUKF.cpp:7-69).  z[0..2] vector part (unused entries 0), for
    // MEAS_ORIENT_SO3 z[0..3] = orientation quaternion.  Returns nothing; `mid` in 0..8.
    static UKFB_DEV void measure(const T (&x)[13], int mid, T (&z)[4]) {
        // candidates: p(0..2) v(3..5) w(6..8)
        const T cand[9] = {x[0], x[1], x[2], x[7], x[8], x[9], x[10], x[11], x[12]};
        // selection table: first index and second/third per model
        //   POS3 0,1,2 | POS_XY 0,1 | POS_Z 2 | VEL3 3,4,5 | VEL_XY 3,4 | VEL_Z 5 | XVEL_YAWVEL 3,8 | ANGVEL3 6,7,8
        int i0 = -1, i1 = -1, i2 = -1;
        if (mid == 0) { i0 = 0; i1 = 1; i2 = 2; }
        else if (mid == 1) { i0 = 0; i1 = 1; }
        else if (mid == 2) { i0 = 2; }
        else if (mid == 4) { i0 = 3; i1 = 4; i2 = 5; }
        else if (mid == 5) { i0 = 3; i1 = 4; }
        else if (mid == 6) { i0 = 5; }
        else if (mid == 7) { i0 = 3; i1 = 8; }
        else if (mid == 8) { i0 = 6; i1 = 7; i2 = 8; }
        T z0 = T(0), z1 = T(0), z2 = T(0);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            z0 = (i0 == s) ? cand[s] : z0;
            z1 = (i1 == s) ? cand[s] : z1;
            z2 = (i2 == s) ? cand[s] : z2;
        }
        const bool so3 = (mid == 3);
        z[0] = so3 ? x[3] : z0;
        z[1] = so3 ? x[4] : z1;
        z[2] = so3 ? x[5] : z2;
        z[3] = so3 ? x[6] : T(0);
    }
    static UKFB_DEV int meas_dim(int mid) {
        return (mid == 2 || mid == 6) ? 1 : ((mid == 1 || mid == 5 || mid == 7) ? 2 : 3);
    }
    static UKFB_DEV bool meas_valid(int mid) { return mid >= 0 && mid <= 8; }
    static UKFB_DEV bool meas_is_so3(int mid) { return mid == 3; }
    static constexpr bool CHECK_MEAS_FINITE = false;  // PoseUKF never calls checkMeasurment
};

// ---------------------------------------------------------------------------------------------
// OrientationState (OrientationState.hpp:20-26): q[0..3] v[4..6] bg[7..9] ba[10..12] g[13]
// ---------------------------------------------------------------------------------------------
template <class T> struct OrientM {
    template <class U> using rebind = OrientM<U>;
    static constexpr int S = 14, D = 13, MODEL = 1;

    static UKFB_DEV void boxplus(T (&x)[14], const T (&d)[13]) {
        T q[4] = {x[0], x[1], x[2], x[3]};
        const T dv[3] = {d[0], d[1], d[2]};
        so3_boxplus(q, dv, T(1));
        x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; x[3] = q[3];
#pragma unroll
        for (int k = 0; k < 10; ++k) x[4 + k] += d[3 + k];
    }
    static UKFB_DEV void boxminus(const T (&x)[14], const T (&y)[14], T (&d)[13]) {
        const T qx[4] = {x[0], x[1], x[2], x[3]}, qy[4] = {y[0], y[1], y[2], y[3]};
        T r[3];
        so3_boxminus(qx, qy, r);
        d[0] = r[0]; d[1] = r[1]; d[2] = r[2];
#pragma unroll
        for (int k = 0; k < 10; ++k) d[3 + k] = x[4 + k] - y[4 + k];
    }
    // processModel (OrientationUKF.cpp:12-32)
    static UKFB_DEV void process(T (&x)[14], const ProcIn<T>& in) {
        T q[4] = {x[0], x[1], x[2], x[3]};
        const T t[3] = {in.w[0] - x[7], in.w[1] - x[8], in.w[2] - x[9]};
        T av[3];
        quat_rotate(q, t, av);
        av[0] -= in.earth[0]; av[1] -= in.earth[1]; av[2] -= in.earth[2];
        so3_boxplus(q, av, in.dt);
        x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; x[3] = q[3];
        const T u[3] = {in.a[0] - x[10], in.a[1] - x[11], in.a[2] - x[12]};
        T acc[3];
        quat_rotate(q, u, acc);  // already updated orientation (:22)
        acc[2] -= x[13];
        x[4] += in.dt * acc[0]; x[5] += in.dt * acc[1]; x[6] += in.dt * acc[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x[7 + k] += in.dt * (in.ninv_tau_g * x[7 + k]);
            x[10 + k] += in.dt * (in.ninv_tau_a * x[10 + k]);
        }
    }
    static UKFB_DEV void orientation(const T (&x)[14], T (&q)[4]) {
        q[0] = x[0]; q[1] = x[1]; q[2] = x[2]; q[3] = x[3];
    }
    // velocityMeasurementModel (OrientationUKF.cpp:34-39): q.inverse() * v
    static UKFB_DEV void measure(const T (&x)[14], int, T (&z)[4]) {
        const T n2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        const T rn = fast_rcp(n2);
        const T qi[4] = {-x[0] * rn, -x[1] * rn, -x[2] * rn, x[3] * rn};
        const T v[3] = {x[4], x[5], x[6]};
        T r[3];
        quat_rotate(qi, v, r);
        z[0] = r[0]; z[1] = r[1]; z[2] = r[2]; z[3] = T(0);
    }
    static UKFB_DEV int meas_dim(int) { return 3; }
    static UKFB_DEV bool meas_valid(int mid) { return mid == 9; }
    static UKFB_DEV bool meas_is_so3(int) { return false; }
    static constexpr bool CHECK_MEAS_FINITE = true;  // OrientationUKF.cpp:67 checkMeasurment
};

}  // namespace ukfb
