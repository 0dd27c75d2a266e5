// ukf_kernel.hpp -- the fused UKF predict / update / cycle kernel for gfx950 (MI355X).
//
// Work decomposition (MI355X-first):
//   * one workgroup = ONE wavefront of 64 lanes; no cross-wave communication, so the only
//     synchronisation is wave-local LDS ordering.
//   * the wavefront is split into 64/G groups of G lanes (G = 16, 32 or 64); each group owns one
//     filter for the whole launch.  G = 16 is one DPP row per filter and packs 4 filters per
//     wavefront; G = 64 is the literal "one wavefront per filter" layout.
//   * a filter's mean, covariance rows, sigma points and measurement sigma points live in VGPRs;
//     the Cholesky factor (column-major), the per-sigma-point tangent deltas and small
//     matrices live in that group's LDS slice.  HBM is touched once on entry (mean + packed lower
//     triangle + per-call inputs) and once on exit.
//   * phases, per group:  row-per-lane Cholesky -> lane-per-sigma-point spread (boxplus) ->
//     process / measurement model -> iterated manifold mean (boxminus + lane-column sums) ->
//     lane-per-entry covariance recombination -> gain / downdate -> apply_delta resampling.
//
// The arithmetic replaces ukfom::ukf<WState>::predict / update as reached from the reference at
// PoseUKF.cpp:114-172,192,195 and OrientationUKF.cpp:69,88 (algorithm: SURVEY.md Appendix A).
#pragma once

#include "ukf_device.hpp"

namespace ukfb {

enum : uint32_t {
    ST_OK = 0u,
    ST_SKIPPED_FIRST_TS = 1u << 0,
    ST_SKIPPED_SMALL_DT = 1u << 1,
    ST_ERR_NEG_DT = 1u << 2,
    ST_ERR_DT_TOO_LARGE = 1u << 3,
    ST_ERR_NONFINITE_MEAS = 1u << 4,
    ST_ERR_CHOLESKY = 1u << 5,
    ST_WARN_MEAN_NOCONV = 1u << 6,
    ST_UNINITIALISED = 1u << 7,
    ST_INACTIVE = 1u << 8,
    ST_REJECTED_GATE = 1u << 9,
};

#ifndef UKFB_MAX_MULTI_CYCLES
#define UKFB_MAX_MULTI_CYCLES 32   // (also in ukf_engine.hpp, for the host side)
#endif
template <class T> struct KArgs {
    int64_t n;                   // work items of this launch (= filters, or entries of fidx): the launch covers [item0, n)
    int64_t item0;               // direct launches of the tuned kernel: first work item (a batch run as two half launches)
    // Indirect launch (event streams): work item i acts on filter fidx[i]; the per-call inputs (ts, dt, meas, active,
    // z, Q) are indexed by i, the engine's per-filter state (everything else) by the filter.  Null: filter i.
    // A filter must not appear twice in one launch.
    const int32_t* fidx;
    // != 0 (indirect launches of the tuned kernel, model-class buckets): the per-call inputs are per-FILTER arrays like the
    // state (indexed by fidx[i], not by i), and a negative entry of fidx is padding: no filter, nothing loaded or stored for it
    int fidx_inputs;
    int status_accumulate;       // != 0: OR the new status word into the stored one instead of replacing it
    int noise_iso;               // != 0: the two 3x3 diagonal noise blocks that the models rotate are multiples of the identity
    int upd_short_ok;            // != 0: the host-side conditions of the short update factorisation hold (ukfb_config::full_update_check)
                                 // (batch-uniform noise only) -- R s I R^T = s I, the rotation is skipped (ukf_kernel16.hpp)
    T* mu;                       // [filters][S]
    T* cov;                      // [n][PK] packed lower triangle, row-major
    uint32_t* status;            // [n]
    const uint8_t* initialised;  // [n]
    // ---- predict
    const T* Rn;                 // process_noise_cov, D*D row-major; per filter if Rn_stride != 0
    int64_t Rn_stride;
    const T* Racc;               // Pose: Rn with block(6,6,3,3) = 2 acc.cov (same stride as Rn)
    const T* in_a;               // Pose: acc.mu [n][3] (may be null) ; Orient: acceleration.mu [n][3]
    const T* in_b;               // Orient: rotation_rate.mu [n][3]
    T acc_cov[9];                // Pose: acceleration.cov (batch-uniform)
    T ninv_tau_g, ninv_tau_a;    // Orient: -1/tau
    T earth[3];                  // Orient: earth rotation
    double dt_uniform;
    const double* dt;            // per filter, may be null
    const int64_t* ts;           // per filter timestamps (us), may be null
    int64_t* last_ts;            // [n]
    double min_dt, max_dt;
    // ---- update
    int meas_uniform;
    const int32_t* meas;         // per filter, may be null
    const T* z;                  // [n][3]
    const T* Q;                  // [n][9]; q_uniform != 0: ONE 3x3 for every filter, [9]
    const uint8_t* active;       // [n], may be null
    // ---- ukfom constants
    T mean_tol;
    int mean_max_it;
    T gate_chi2;                 // < 0: accept any
    // ---- multi-cycle launches (tuned kernel, fused cycle only): cyc_count consecutive cycles in one launch, the filter
    // state stays in LDS between them.  Cycle c reads input slot (cyc_first + c) % cyc_ring of z, Q ([cyc_ring][cyc_items]
    // [3], [..][9]) and of in_a (cyc_in & 1) / in_b (cyc_in & 2) when they are slotted ([cyc_ring][cyc_items][3]); otherwise
    // the latched inputs serve every cycle.  Single-cycle launches leave all of this zero.
    int cyc_count, cyc_first, cyc_ring, cyc_in;
    int64_t cyc_items;
    // per-cycle schedule (cyc_sched != 0): time step and measurement model of cycle c instead of dt_uniform / meas_uniform;
    // a negative model = prediction only in that cycle.  Launch-wide scalars: the kernel reads them with scalar loads.
    int q_uniform;               // the measurement covariance is batch-uniform: Q holds 9 scalars (single-cycle launches)
    int cyc_sched;
    int32_t cyc_model[UKFB_MAX_MULTI_CYCLES];
    double cyc_dt[UKFB_MAX_MULTI_CYCLES];
#if defined(UKFB_STAMPS) || defined(UKFB_COUNTS)
    unsigned long long* stamps;  // diagnostic builds: [grid][UKFB_MAX_STAMPS] s_memtime per phase marker (UKFB_STAMPS) / value + 1 per counter slot (UKFB_COUNTS)
#endif
};
#define UKFB_MAX_STAMPS 32

// LDS slice of one filter, in scalars of T.
template <class T, class M> struct Layout {
    static constexpr int VEC = 16 / int(sizeof(T));
    static constexpr int D = M::D, S = M::S, N = 2 * D + 1;
    static constexpr int LS = (D + VEC - 1) / VEC * VEC;  // column stride of the Cholesky factor
    static constexpr int DS = 16;                         // row stride of the delta table
    static constexpr int ZOFF = D;                        // measurement deltas behind the state deltas
    static constexpr int LC_OFF = 0;                      // D*LS : packed Sigma staging / L columns / small matrices
    static constexpr int DX_OFF = D * LS;                 // N*DS : [dx(0..D-1) | dz(0..2)] per sigma point
    static constexpr int MISC_OFF = DX_OFF + N * DS;      // 72   : MU MD ROT ZQ DEL
    static constexpr int PK = D * (D + 1) / 2;
    static constexpr int PKP = (PK + VEC - 1) / VEC * VEC;
    static constexpr int PKS_OFF = MISC_OFF + 72;         // PKP  : packed covariance of the current state
    static constexpr int DUM_OFF = PKS_OFF + PKP;         // 16   : sink for lane-predicated stores
    static constexpr int PF = DUM_OFF + 16;
    // small matrices that reuse the (dead) factor region between two Choleskys
    static constexpr int SMAT = 0, CXZ = 12, KMAT = 52, KEND = 92;
    static_assert(D + 3 <= DS, "delta row too small");
    static_assert(KEND <= D * LS, "small-matrix scratch exceeds the factor region");
    static_assert(PF % VEC == 0, "slice must keep 16-byte alignment");
};

template <class T, class M> constexpr int lds_bytes_per_filter() { return Layout<T, M>::PF * int(sizeof(T)); }

UKFB_DEV void wsync() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
}  // one wavefront per workgroup: no s_barrier wait across waves

template <int G> UKFB_DEV double gshfl(double v, int src) { return __shfl(v, src, G); }
template <int G> UKFB_DEV float gshfl(float v, int src) { return __shfl(v, src, G); }

// Row-per-lane right-looking Cholesky.  Lane l < D holds row l (entries 0..l) in a[]; column k of
// the factor is published to Lc (column-major, stride LS, zeros above the diagonal) as soon as it
// is final and read back (LDS broadcast) for the trailing update.  false: a pivot was <= 0
// (Eigen LLT: NumericalIssue).
template <class T, int D, int LS, int G> UKFB_DEV bool chol_rows_to_lds(T (&a)[D], T* Lc, int l, int dum) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const T akk = gshfl<G>(a[k], k);
        ok = ok && (akk > T(0));
        const T inv = fast_rsqrt(akk);
        const T d = akk * inv;
        const T lk = (l > k) ? a[k] * inv : ((l == k) ? d : T(0));
        Lc[(l < D) ? (k * LS + l) : dum] = lk;
        wsync();
#pragma unroll
        for (int c = k + 1; c < D; ++c) a[c] -= lk * Lc[k * LS + c];
    }
    return ok;
}

// Sigma point i of the set {mu+d0, mu+(d0+L col j), mu+(d0-L col j)} (ukfom generate_sigma_points).
template <class T, class M, int LS, bool HAS_D0>
UKFB_DEV void sigma_point(const T (&mu)[M::S], const T* Lc, const T (&d0)[M::D], int i, T (&x)[M::S]) {
    constexpr int D = M::D, S = M::S;
    const int j = (i > 0) ? ((i - 1) >> 1) : 0;
    const T sgn = (i == 0) ? T(0) : ((i & 1) ? T(1) : T(-1));
    T d[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const T lc = sgn * Lc[j * LS + c];
        d[c] = HAS_D0 ? (d0[c] + lc) : lc;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) x[s] = mu[s];
    M::boxplus(x, d);
}

// ukfom meanSigmaPoints on the state manifold: lanes hold the sigma points, deltas go through the
// group's LDS table, lanes 0..D-1 each sum one tangent component in sigma-point order.
template <class T, class M, int G, int RND, int DS>
UKFB_DEV bool mean_loop_state(const T (&X)[RND][M::S], T (&ref)[M::S], T* DX, T* MD, T* DUMP, int l, T tol, int max_it) {
    constexpr int D = M::D, S = M::S, N = 2 * D + 1;
    bool active = true, conv = true;
    int it = 0;
    const int lc = (l < D) ? l : (D - 1);
    for (;;) {
#pragma unroll
        for (int r = 0; r < RND; ++r) {
            const int i = l + G * r;
            T d[D];
            M::boxminus(X[r], ref, d);
            T* row = (i < N) ? (DX + i * DS) : DUMP;
#pragma unroll
            for (int c = 0; c < D; ++c) row[c] = d[c];
        }
        wsync();
        {
            T md = T(0);
#pragma unroll 5
            for (int i = 0; i < N; ++i) md += DX[i * DS + lc];
            MD[(l < D) ? l : 15] = md * (T(1) / T(N));   // MD has 16 slots; D <= 13
        }
        wsync();
        T mdv[D];
        T n2 = T(0);
#pragma unroll
        for (int c = 0; c < D; ++c) {
            mdv[c] = MD[c];
            n2 += mdv[c] * mdv[c];
        }
        T nref[S];
#pragma unroll
        for (int s = 0; s < S; ++s) nref[s] = ref[s];
        M::boxplus(nref, mdv);
#pragma unroll
        for (int s = 0; s < S; ++s) ref[s] = active ? nref[s] : ref[s];
        const bool more = n2 > tol * tol;
        const bool capped = more && (it + 1 >= max_it);
        it += (active && more) ? 1 : 0;
        conv = conv && !(active && capped);
        active = active && more && !capped;
        if (!wave_any(active)) break;
    }
    return conv;
}

// lower-triangle entry index -> (row, col)
UKFB_DEV void tri_decode(int e, int& r, int& c) {
    int rr = int((sqrtf(8.0f * float(e) + 1.0f) - 1.0f) * 0.5f);
    if (rr * (rr + 1) / 2 > e) --rr;
    if ((rr + 1) * (rr + 2) / 2 <= e) ++rr;
    r = rr;
    c = e - rr * (rr + 1) / 2;
}

// 0.5 * sum_i DX[i][r] DX[i][c] for this lane's packed entries (ukfom covSigmaPoints).
template <class T, int N, int DS, int EPL>
UKFB_DEV void cov_entries(const T* DX, const int (&er)[EPL], const int (&ec)[EPL], T (&P)[EPL]) {
#pragma unroll
    for (int t = 0; t < EPL; ++t) P[t] = T(0);
#pragma unroll 2
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int t = 0; t < EPL; ++t) P[t] += DX[i * DS + er[t]] * DX[i * DS + ec[t]];
    }
#pragma unroll
    for (int t = 0; t < EPL; ++t) P[t] *= T(0.5);
}

// Eigen fixed-size 3x3 inverse (cofactors / determinant)
template <class T> UKFB_DEV void inverse3(const T (&m)[9], T (&r)[9]) {
    const T c00 = m[4] * m[8] - m[5] * m[7];
    const T c10 = m[7] * m[2] - m[8] * m[1];
    const T c20 = m[1] * m[5] - m[2] * m[4];
    const T det = c00 * m[0] + c10 * m[3] + c20 * m[6];
    const T invdet = fast_rcp(det);
    r[0] = c00 * invdet;
    r[1] = c10 * invdet;
    r[2] = c20 * invdet;
    r[3] = (m[5] * m[6] - m[3] * m[8]) * invdet;
    r[4] = (m[0] * m[8] - m[2] * m[6]) * invdet;
    r[5] = (m[2] * m[3] - m[0] * m[5]) * invdet;
    r[6] = (m[3] * m[7] - m[4] * m[6]) * invdet;
    r[7] = (m[1] * m[6] - m[0] * m[7]) * invdet;
    r[8] = (m[0] * m[4] - m[1] * m[3]) * invdet;
}

// One entry of the shaped process noise R (PoseUKF.cpp:183-191, OrientationUKF.cpp:82-86).
template <class T, class M>
UKFB_DEV T process_noise_entry(const T* Rn, const T* ROT, const KArgs<T>& a, const ProcIn<T>& pin, int r, int c) {
    constexpr int D = M::D;
    const T rn = Rn[r * D + c];
    if (M::MODEL == 0 && pin.use_acc) {
        // acceleration branch: raw process_noise_cov, block(6,6,3,3) = 2 acc.cov, no rotation, no dt
        const bool vel = (r >= 6 && r < 9 && c >= 6 && c < 9);
        const int k = vel ? ((r - 6) * 3 + (c - 6)) : 0;
        T ac = T(0);
#pragma unroll
        for (int s = 0; s < 9; ++s) ac = (k == s) ? a.acc_cov[s] : ac;
        return vel ? T(2) * ac : rn;
    }
    const int o = (r < 3 && c < 3) ? 0 : ((r >= 3 && r < 6 && c >= 3 && c < 6) ? 3 : -1);
    T val = rn;
    if (o >= 0) {
        // (rot * B) * rot^T, entry (r-o, c-o)
        const int rr = r - o, cc = c - o;
        T acc = T(0);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            T tmp = T(0);
#pragma unroll
            for (int k = 0; k < 3; ++k) tmp += ROT[rr * 3 + k] * Rn[(o + k) * D + (o + m)];
            acc += tmp * ROT[cc * 3 + m];
        }
        val = acc;
    }
    const T scale = (M::MODEL == 0) ? pin.dt : pin.dt * pin.dt;
    return scale * val;
}

// minimum waves per SIMD the register allocator must leave room for.  fp64: 1, i.e. the whole 512-register file --
// at 2 the OrientationState instantiations of this (ablation) layout needed 12..64 bytes of scratch, and spilled code
// is not trusted with this toolchain (note in ukf_kernel16.hpp); tools/check_resources.py rejects scratch in ANY kernel.
template <class T> constexpr int min_waves_per_simd() { return sizeof(T) == 8 ? 1 : 2; }

template <class T, class M, int G, bool DO_PREDICT, bool DO_UPDATE>
__global__ void __launch_bounds__(64, min_waves_per_simd<T>()) ukf_kernel(const KArgs<T> a) {
    constexpr int S = M::S, D = M::D, N = 2 * D + 1, PK = D * (D + 1) / 2;
    using LY = Layout<T, M>;
    constexpr int LS = LY::LS, DS = LY::DS, ZO = LY::ZOFF;
    constexpr int FPW = 64 / G, RND = (N + G - 1) / G, EPL = (PK + G - 1) / G;
    static_assert(G >= 16 && S <= G, "group too small");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const int g = lane / G, l = lane % G;
    const int64_t f = int64_t(blockIdx.x) * FPW + g;
    const bool fvalid = f < a.n;
    const int64_t fi = fvalid ? f : (a.n - 1);                    // work item: index of the per-call inputs
    const int64_t fc = a.fidx ? int64_t(a.fidx[fi]) : fi;         // filter: index of the engine's state
    T* base = reinterpret_cast<T*>(smem_raw) + g * LY::PF;
    T* Lc = base + LY::LC_OFF;
    T* DX = base + LY::DX_OFF;
    T* MS = base + LY::MISC_OFF;
    T* MU = MS; T* MD = MS + 16; T* ROT = MS + 32; T* ZQ = MS + 44; T* DEL = MS + 56;
    T* PKS = base + LY::PKS_OFF;
    T* DUMP = base + LY::DUM_OFF;

    // this lane's packed covariance entries
    int er[EPL], ec[EPL];
    bool ev[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int e = l + G * t;
        ev[t] = e < PK;
        tri_decode(ev[t] ? e : (PK - 1), er[t], ec[t]);
    }

    uint32_t st = ST_OK;
    const bool live = fvalid && (a.initialised[fc] != 0);
    st |= (fvalid && !live) ? ST_UNINITIALISED : 0u;

    // ---- time gate (UnscentedKalmanFilter.hpp:83-125)
    bool do_p = false, p_error = false;
    bool noev = false;   // ts < 0: this filter has no sample in this call -> neither predicted nor updated
    T dtT = T(0);
    if constexpr (DO_PREDICT) {
        double dt;
        bool first = false;
        if (a.ts) {
            const int64_t last = a.last_ts[fc], ts = a.ts[fi];
            noev = ts < 0;                       // event streams: this filter has no sample in this call
            first = (last == 0) && !noev;
            dt = (first || noev) ? 0.0 : double(ts - last) / 1000000.0;
            if (live && l == 0 && !noev && (first || dt > a.min_dt)) a.last_ts[fc] = ts;
        } else {
            dt = a.dt ? a.dt[fi] : a.dt_uniform;
        }
        const bool neg = dt < 0.0, small = dt <= a.min_dt, large = dt > a.max_dt;
        const uint32_t code = first ? ST_SKIPPED_FIRST_TS
                                    : (neg ? ST_ERR_NEG_DT : (small ? ST_SKIPPED_SMALL_DT : (large ? ST_ERR_DT_TOO_LARGE : 0u)));
        st |= (live && !noev) ? code : 0u;
        p_error = live && !first && !noev && (neg || (!small && large));
        do_p = live && !noev && code == 0u;
        dtT = T(dt);
    }

    // ---- measurement selection
    bool do_u = false;
    int mid = -1;
    if constexpr (DO_UPDATE) {
        mid = a.meas ? a.meas[fi] : a.meas_uniform;
        const bool act = M::meas_valid(mid) && (a.active ? a.active[fi] != 0 : true);
        do_u = live && act && !p_error && !noev;
        st |= (live && !do_u) ? ST_INACTIVE : 0u;
    }

    // ---- load: packed covariance + mean -> LDS -> registers (mean replicated, row l on lane l)
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int e = l + G * t;
        const T v = a.cov[fc * PK + (ev[t] ? e : (PK - 1))];
        PKS[ev[t] ? e : (LY::DUM_OFF - LY::PKS_OFF)] = v;
    }
    {
        const T v = a.mu[fc * S + ((l < S) ? l : (S - 1))];
        MU[(l < S) ? l : (LY::DUM_OFF - LY::MISC_OFF)] = v;
    }
    wsync();
    T mu_r[S];
#pragma unroll
    for (int s = 0; s < S; ++s) mu_r[s] = MU[s];
    const int lrow = (l < D) ? l : (D - 1);   // row of the packed covariance this lane reads

    T zero_d[D];
#pragma unroll
    for (int c = 0; c < D; ++c) zero_d[c] = T(0);

    bool p_commit = false, u_commit = false;
    T Pn[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) Pn[t] = T(0);

    // =========================================================================== predict
    if constexpr (DO_PREDICT) {
        if (wave_any(do_p)) {
            ProcIn<T> pin;
            pin.dt = dtT;
            pin.ninv_tau_g = a.ninv_tau_g;
            pin.ninv_tau_a = a.ninv_tau_a;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pin.earth[k] = a.earth[k];
                pin.a[k] = a.in_a ? a.in_a[fc * 3 + k] : T(NAN);
                pin.w[k] = a.in_b ? a.in_b[fc * 3 + k] : T(0);
            }
            pin.use_acc = m_finite(pin.a[0]) && m_finite(pin.a[1]) && m_finite(pin.a[2]);

            {   // rotation matrix of the current mean (PoseUKF.cpp:182 / OrientationUKF.cpp:81)
                T q[4], rot[9];
                M::orientation(mu_r, q);
                quat_to_matrix(q, rot);
                T* dst = (l == 0) ? ROT : DUMP;
#pragma unroll
                for (int k = 0; k < 9; ++k) dst[k] = rot[k];
            }

            T arow[D];
#pragma unroll
            for (int j = 0; j < D; ++j) arow[j] = (j <= lrow) ? PKS[lrow * (lrow + 1) / 2 + j] : T(0);
            const bool ok = chol_rows_to_lds<T, D, LS, G>(arow, Lc, l, LY::DUM_OFF - LY::LC_OFF);

            T X[RND][S];
#pragma unroll
            for (int r = 0; r < RND; ++r) {
                const int i = min(l + G * r, N - 1);
                sigma_point<T, M, LS, false>(mu_r, Lc, zero_d, i, X[r]);
                M::process(X[r], pin);
            }
            T ref[S];
#pragma unroll
            for (int s = 0; s < S; ++s) ref[s] = gshfl<G>(X[0][s], 0);

            const bool conv = mean_loop_state<T, M, G, RND, DS>(X, ref, DX, MD, DUMP, l, a.mean_tol, a.mean_max_it);

#pragma unroll
            for (int r = 0; r < RND; ++r) {
                const int i = l + G * r;
                T d[D];
                M::boxminus(X[r], ref, d);
                T* row = (i < N) ? (DX + i * DS) : DUMP;
#pragma unroll
                for (int c = 0; c < D; ++c) row[c] = d[c];
            }
            wsync();
            cov_entries<T, N, DS, EPL>(DX, er, ec, Pn);
            const T* Rn = a.Rn + fc * a.Rn_stride;
#pragma unroll
            for (int t = 0; t < EPL; ++t) Pn[t] += process_noise_entry<T, M>(Rn, ROT, a, pin, er[t], ec[t]);
            wsync();

            p_commit = do_p && ok;
            st |= (do_p && !ok) ? ST_ERR_CHOLESKY : 0u;
            st |= (p_commit && !conv) ? ST_WARN_MEAN_NOCONV : 0u;
#pragma unroll
            for (int t = 0; t < EPL; ++t) PKS[(ev[t] && p_commit) ? (l + G * t) : (LY::DUM_OFF - LY::PKS_OFF)] = Pn[t];
#pragma unroll
            for (int s = 0; s < S; ++s) mu_r[s] = p_commit ? ref[s] : mu_r[s];
            wsync();
        }
    }

    // =========================================================================== update
    T mu2[S];
#pragma unroll
    for (int s = 0; s < S; ++s) mu2[s] = mu_r[s];
    T Pu[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) Pu[t] = T(0);

    if constexpr (DO_UPDATE) {
        if (wave_any(do_u)) {
            {
                const int zi = (l < 3) ? l : 0, qi = (l >= 3 && l < 12) ? (l - 3) : 0;
                const T zv = a.z[fi * 3 + zi], qv = a.Q[(a.q_uniform ? 0 : fi * 9) + qi];
                ZQ[(l < 12) ? l : (LY::DUM_OFF - LY::MISC_OFF - 44)] = (l < 3) ? zv : qv;
            }
            wsync();
            T zin[3], Qm[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) zin[k] = ZQ[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) Qm[k] = ZQ[3 + k];
            if (M::CHECK_MEAS_FINITE) {
                bool fin = true;
#pragma unroll
                for (int k = 0; k < 3; ++k) fin = fin && m_finite(zin[k]);
#pragma unroll
                for (int k = 0; k < 9; ++k) fin = fin && m_finite(Qm[k]);
                st |= (do_u && !fin) ? ST_ERR_NONFINITE_MEAS : 0u;
                do_u = do_u && fin;
            }
            const int midc = M::meas_valid(mid) ? mid : (M::MODEL == 0 ? 0 : 9);
            const int m = M::meas_dim(midc);
            const bool so3 = M::meas_is_so3(midc);
            // unused trailing dimensions are decoupled: Q = I there, z = h = 0
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const T pad = (r == c) ? T(1) : T(0);
                    Qm[r * 3 + c] = (r >= m || c >= m) ? pad : Qm[r * 3 + c];
                }
            T zval[4];
            {
                T qe[4];
                so3_exp_fast(zin, T(1), qe);  // RotationType(SO3::exp(mu)), PoseUKF.cpp:135
#pragma unroll
                for (int k = 0; k < 3; ++k) zval[k] = so3 ? qe[k] : ((k < m) ? zin[k] : T(0));
                zval[3] = so3 ? qe[3] : T(0);
            }

            T arow[D];
#pragma unroll
            for (int j = 0; j < D; ++j) arow[j] = (j <= lrow) ? PKS[lrow * (lrow + 1) / 2 + j] : T(0);
            const bool ok1 = chol_rows_to_lds<T, D, LS, G>(arow, Lc, l, LY::DUM_OFF - LY::LC_OFF);

            T X[RND][S], Z[RND][4];
#pragma unroll
            for (int r = 0; r < RND; ++r) {
                const int i = min(l + G * r, N - 1);
                sigma_point<T, M, LS, false>(mu_r, Lc, zero_d, i, X[r]);
                M::measure(X[r], midc, Z[r]);
            }
            // ---- mean of Z (ukfom meanSigmaPoints on the measurement space)
            T zref[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) zref[k] = gshfl<G>(Z[0][k], 0);
            bool zconv = true;
            const bool any_so3 = wave_any(so3);
            {
                bool active = true;
                int it = 0;
                for (;;) {
#pragma unroll
                    for (int r = 0; r < RND; ++r) {
                        const int i = l + G * r;
                        T dz[3], dq[3] = {T(0), T(0), T(0)};
                        if (any_so3) so3_boxminus(Z[r], zref, dq);   // wave-uniform
#pragma unroll
                        for (int k = 0; k < 3; ++k) dz[k] = so3 ? dq[k] : (Z[r][k] - zref[k]);
                        T* row = (i < N) ? (DX + i * DS + ZO) : DUMP;
#pragma unroll
                        for (int k = 0; k < 3; ++k) row[k] = dz[k];
                    }
                    wsync();
                    {
                        const int lz = (l < 3) ? l : 2;
                        T md = T(0);
#pragma unroll 5
                        for (int i = 0; i < N; ++i) md += DX[i * DS + ZO + lz];
                        MD[(l < 3) ? l : 15] = md * (T(1) / T(N));
                    }
                    wsync();
                    T mdv[3];
                    T n2 = T(0);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        mdv[k] = MD[k];
                        n2 += mdv[k] * mdv[k];
                    }
                    T nz[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) nz[k] = zref[k];
                    T nq[4] = {zref[0], zref[1], zref[2], zref[3]};
                    if (any_so3) so3_boxplus(nq, mdv, T(1));   // wave-uniform
#pragma unroll
                    for (int k = 0; k < 3; ++k) nz[k] = so3 ? nq[k] : (nz[k] + mdv[k]);
                    nz[3] = so3 ? nq[3] : nz[3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) zref[k] = active ? nz[k] : zref[k];
                    const bool more = n2 > a.mean_tol * a.mean_tol;
                    const bool capped = more && (it + 1 >= a.mean_max_it);
                    it += (active && more) ? 1 : 0;
                    zconv = zconv && !(active && capped);
                    active = active && more && !capped;
                    if (!wave_any(active)) break;
                }
            }
            // ---- final deltas: dz_i = Z_i - zbar, dx_i = X_i - mu
#pragma unroll
            for (int r = 0; r < RND; ++r) {
                const int i = l + G * r;
                const int ic = (i < N) ? i : (N - 1);
                T dz[3], dq[3] = {T(0), T(0), T(0)};
                if (any_so3) so3_boxminus(Z[r], zref, dq);
#pragma unroll
                for (int k = 0; k < 3; ++k) dz[k] = so3 ? dq[k] : (Z[r][k] - zref[k]);
                T* row = (i < N) ? (DX + i * DS) : DUMP;   // DUMP has 16 slots = DS
#pragma unroll
                for (int k = 0; k < 3; ++k) row[ZO + k] = dz[k];
                // (mu [+] d) [-] mu = d: the state deltas are the signed factor columns themselves
                const int jj = (ic > 0) ? ((ic - 1) >> 1) : 0;
                const T sg = (ic == 0) ? T(0) : ((ic & 1) ? T(1) : T(-1));
#pragma unroll
                for (int c = 0; c < D; ++c) row[c] = sg * Lc[jj * LS + c];
            }
            wsync();
            // ---- S = 0.5 sum dz dz^T + Q (9 entries), Cxz = 0.5 sum dx dz^T (3D entries)
            {
                constexpr int NE = 9 + 3 * D, TPL = (NE + G - 1) / G;
#pragma unroll
                for (int t = 0; t < TPL; ++t) {
                    const int q0 = l + G * t;
                    const int qi = (q0 < NE) ? q0 : (NE - 1);
                    {
                        const bool isS = qi < 9;
                        const int ra = isS ? (ZO + qi / 3) : ((qi - 9) / 3);
                        const int cb = ZO + (isS ? (qi % 3) : ((qi - 9) % 3));
                        T acc = T(0);
#pragma unroll 5
                        for (int i = 0; i < N; ++i) acc += DX[i * DS + ra] * DX[i * DS + cb];
                        acc *= T(0.5);
                        T qv = T(0);
#pragma unroll
                        for (int s = 0; s < 9; ++s) qv = (qi == s) ? Qm[s] : qv;
                        const int dst = isS ? (LY::SMAT + qi) : (LY::CXZ + (qi - 9));
                        Lc[(q0 < NE) ? dst : (LY::DUM_OFF - LY::LC_OFF)] = isS ? (acc + qv) : acc;
                    }
                }
            }
            wsync();
            // ---- gain, innovation, downdate (ukfom update)
            T Sm[9], Si[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) Sm[k] = Lc[LY::SMAT + k];
            inverse3(Sm, Si);
            T innov[3];
            {
                T iq[3] = {T(0), T(0), T(0)};
                if (any_so3) so3_boxminus(zval, zref, iq);
#pragma unroll
                for (int k = 0; k < 3; ++k) innov[k] = so3 ? iq[k] : (zval[k] - zref[k]);
            }
            T maha = T(0);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) maha += innov[r] * Si[r * 3 + c] * innov[c];
            const bool accept = (a.gate_chi2 < T(0)) || (maha <= a.gate_chi2);

            const int la = (l < D) ? l : (D - 1);
            T Kr[3], KSr[3];
            {
                T cx[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) cx[k] = Lc[LY::CXZ + la * 3 + k];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    T s = T(0);
#pragma unroll
                    for (int k = 0; k < 3; ++k) s += cx[k] * Si[k * 3 + c];
                    Kr[c] = s;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    T s = T(0);
#pragma unroll
                    for (int k = 0; k < 3; ++k) s += Kr[k] * Sm[k * 3 + c];
                    KSr[c] = s;
                }
                T del = T(0);
#pragma unroll
                for (int k = 0; k < 3; ++k) del += Kr[k] * innov[k];
                T* krow = (l < D) ? (Lc + LY::KMAT + l * 3) : DUMP;
#pragma unroll
                for (int k = 0; k < 3; ++k) krow[k] = Kr[k];
                DEL[(l < D) ? l : 15] = del;   // DEL has 16 slots
            }
            wsync();
            T arow2[D], dl[D];
#pragma unroll
            for (int b = 0; b < D; ++b) {
                T s = T(0);
#pragma unroll
                for (int k = 0; k < 3; ++k) s += KSr[k] * Lc[LY::KMAT + b * 3 + k];
                arow2[b] = ((b <= lrow) ? PKS[lrow * (lrow + 1) / 2 + b] : T(0)) - s;
                dl[b] = DEL[b];
            }
            wsync();
            const bool ok2 = chol_rows_to_lds<T, D, LS, G>(arow2, Lc, l, LY::DUM_OFF - LY::LC_OFF);

            // ---- applyDelta: resample around mu + delta
#pragma unroll
            for (int r = 0; r < RND; ++r) {
                const int i = min(l + G * r, N - 1);
                sigma_point<T, M, LS, true>(mu_r, Lc, dl, i, X[r]);
            }
#pragma unroll
            for (int s = 0; s < S; ++s) mu2[s] = gshfl<G>(X[0][s], 0);
#pragma unroll
            for (int r = 0; r < RND; ++r) {
                const int i = l + G * r;
                T d[D];
                M::boxminus(X[r], mu2, d);
                T* row = (i < N) ? (DX + i * DS) : DUMP;
#pragma unroll
                for (int c = 0; c < D; ++c) row[c] = d[c];
            }
            wsync();
            cov_entries<T, N, DS, EPL>(DX, er, ec, Pu);
            wsync();

            st |= (do_u && (!ok1 || (accept && !ok2))) ? ST_ERR_CHOLESKY : 0u;
            st |= (do_u && ok1 && !accept) ? ST_REJECTED_GATE : 0u;
            st |= (do_u && ok1 && !zconv) ? ST_WARN_MEAN_NOCONV : 0u;
            u_commit = do_u && ok1 && ok2 && accept;
        }
    }

    // =========================================================================== commit
    const bool changed = p_commit || u_commit;
    if (wave_any(changed)) {
#pragma unroll
        for (int t = 0; t < EPL; ++t) PKS[(ev[t] && u_commit) ? (l + G * t) : (LY::DUM_OFF - LY::PKS_OFF)] = Pu[t];
        {
            T* dst = (l == 0) ? MU : DUMP;
#pragma unroll
            for (int s = 0; s < S; ++s) dst[s] = u_commit ? mu2[s] : mu_r[s];
        }
        wsync();
        if (changed && fvalid) {
#pragma unroll
            for (int t = 0; t < EPL; ++t)
                if (ev[t]) a.cov[fc * PK + l + G * t] = PKS[l + G * t];
            if (l < S) a.mu[fc * S + l] = MU[l];
        }
    }
    if (fvalid && l == 0) a.status[fc] = a.status_accumulate ? (a.status[fc] | st) : st;
}

}  // namespace ukfb
