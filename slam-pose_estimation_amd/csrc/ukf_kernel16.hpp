// ukf_kernel16.hpp -- tuned fused UKF kernel: one DPP row (16 lanes) per filter, 4 filters per
// wavefront, one wavefront per workgroup (gfx950 / MI355X).
//
// Same arithmetic contract as ukf_kernel.hpp (ukfom predict / update as reached from
// PoseUKF.cpp:114-172,192,195 and OrientationUKF.cpp:69,88), re-organised after profiling the
// first kernel (profiles/r01_base_*): it was VALU-issue and LDS-latency bound, not HBM bound.
//
//  * lane l < D owns the sigma PAIR mu [+] (+L col l), mu [+] (-L col l): one L column read and one
//    SO(3) exp serve both points (exp(-v) = conj(exp(v))); lane D owns the centre point.
//  * prediction: only the first NL = 6 tangent components are nonlinear in the state (Pose: position, orientation;
//    Orient: orientation, velocity).  The affine rest needs no sigma-point sums: its mean is the propagated centre,
//    its deltas are the signed, scaled factor rows, so Sigma'[affine][affine] = s s Sigma + R in place and the cross
//    block is (affine factor rows) x (half differences W_l of the nonlinear deltas).  The delta table has 6 columns
//    and stores U_l = (d+ + d-)/2, d0/sqrt(2), W_l = (d+ - d-)/2: 0.5 sum d d^T = sum over rows of row row^T.
//  * manifold means over the NL components: fp32 by a 4-step DPP butterfly, fp64 (three instructions per butterfly
//    step) by two steps + broadcast FMAs of the quad sums, the 6-wide first iteration through an LDS transposition.
//    Euclidean components converge in the first iteration, later iterations only touch SO(3).  The SO(3) log takes
//    the known norm of its argument, which buys a half-angle step for one addition (ukf_device.hpp).
//  * Cholesky: rows in VGPRs; pivot and column entries travel by DPP row_newbcast fused into the FMA
//    (v_fmac_*_dpp), no LDS inside the factorisation; the columns are published after the last step.
//  * gain rows, delta and cross-term rows are exchanged the same way (row_newbcast), not through LDS.
//  * covariance: 16 work items of one register tile each (2x3 for D = 12, 3x3 for D = 13) over the delta table in
//    LDS -- the tiles of the nonlinear block are shared by two lanes (half the rows each), the cross block has one
//    lane per tile -- D + 1 iterations for every lane.
//  * the update exploits exact identities of the unscented transform instead of recomputing them:
//    (mu [+] d) [-] mu = d for the state deltas; linear (sub-state) measurements have S, Sigma_xz in closed
//    form; in applyDelta the Euclidean block of 0.5 sum (X_i [-] X_0)(..)^T equals L' L'^T = Sigma' entry for
//    entry, so only the rows/columns of the SO(3) component are re-sampled (through exp/log), and the 2 (RT + 3)
//    signed sigma points that have a rotation part are spread over lanes (one exp / log each).  Results differ
//    from the literal restatement by rounding only (tests/test_gpu_parity*.py hold both to 1e-9 / 1e-4).
//  * every per-filter stream is requested in the prologue, before the first dependent instruction.
//  * indirect launches (KArgs::fidx): work item i acts on filter fidx[i]; event streams launch each round over
//    exactly the filters that have a sample in it.
//  * lane constants of the prediction's covariance phase (which tile, operand and result offsets) come from a compile-time
//    table in the Pose kernels (CovTab) instead of being decoded by every wavefront.
//  * multi-cycle launches (MULTI): C fused cycles in one launch, the filter stays in LDS between them; bit-identical with C
//    single launches, the per-cycle inputs come from device rings.
//  * LDS per filter (Layout16): factor columns (stride 14) aliased by the delta table and the fp64 transposition,
//    packed-covariance staging, affine factor rows, 56 scalars of mean / rotation / z, Q / store sink.
//  * round 3: model-class buckets (KArgs::fidx_inputs: an indirect launch over a filter list grouped by update class, the
//    per-call inputs indexed by the filter, -1 entries = padding); split launches (KArgs::item0: a direct launch as two half
//    launches on two streams); the OrientationState fp64 slice trimmed to 11 workgroups per CU (Layout16::ORIENT64_TRIM,
//    SINK_IN_NSH, LAF_PAD: every LDS scalar a kernel reads is one it wrote -- tests/test_gpu_lds_leftovers.py).
//  * measured-and-rejected variants (the four-wavefront fp64 slice, the late minus point, Newton steps on the fp32 seeds, ...)
//    are NOT in this header: tools/variants/ holds them as patches that tools/build_variant.sh applies to a scratch copy.
//
// TOOLCHAIN NOTE (ROCm 7.2 hipcc, -O2/-O3): when this kernel needed VGPR spills / live-range
// splits, the compiler placed the copies at the join label of a divergent `if` BEFORE the
// `s_or_b64 exec` restore; entered through `s_cbranch_execz` (EXEC = 0) they save nothing and the
// reload returns stale registers.  The kernel is therefore written to need no spills (checked by
// the build: scratch = 0, AGPR = 0) and with as few exec-masked regions as possible: lane-predicated
// LDS stores select a dummy address instead of branching, rare math fall-backs are wave-uniform.
#pragma once

#include <type_traits>

#include "ukf_kernel.hpp"

namespace ukfb {

// ---------------------------------------------------------------------------------------------
// DPP row (16 lanes) all-reduce
// ---------------------------------------------------------------------------------------------
template <int CTRL> UKFB_DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> UKFB_DEV double dpp_mov(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, int(unsigned(b)), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, int(unsigned(b >> 32)), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32));
}
// value held by lane C of this lane's row (row_newbcast: one VALU move, no LDS round trip)
template <int C> UKFB_DEV float row_bcast(float v) { return dpp_mov<0x150 + C>(v); }
template <int C> UKFB_DEV double row_bcast(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + C, 0xF, 0xF, true); }
// acc += (lane C's src) * m as ONE instruction (the compiler keeps v_mov_dpp + v_fmac apart).  Inline asm is
// opaque to the hazard recognizer: callers put dpp_hazard_fence(src) between the producer of `src` and the
// first use (VALU write -> DPP read needs 2 wait states); volatile keeps that order.
template <int C> UKFB_DEV void fmac_bcast(float& acc, float src, float m) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(C));
}
template <int C> UKFB_DEV void fmac_bcast(double& acc, double src, double m) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(C));
}
// 1 / (lane C's v): in fp32 the reciprocal reads its operand through DPP, no broadcast move.
template <int C> UKFB_DEV float rcp_bcast(float v) {
    float r;
    asm volatile("v_rcp_f32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(C));
    return r;
}
// (fp64: the assembler accepts v_rcp_f64_dpp, the hardware does not broadcast for it -- every pivot came out wrong; the move stays)
template <int C> UKFB_DEV double rcp_bcast(double v) { return fast_rcp(row_bcast<C>(v)); }
// true on every lane of a 16-lane row iff the lane mask b holds on all 16 of them: the reduction is scalar (and-fold of the mask
// inside each 16-bit field, widened back to a lane mask); needs all 64 lanes active
UKFB_DEV bool row_all(unsigned long long b) {
    unsigned long long t = b & (b >> 8);
    t &= t >> 4;
    t &= t >> 2;
    t &= t >> 1;
    t &= 0x0001000100010001ull;
    return __builtin_amdgcn_inverse_ballot_w64((t << 16) - t);
}
UKFB_DEV void dpp_hazard_fence(float src) { asm volatile("s_nop 1" ::"v"(src)); }
UKFB_DEV void dpp_hazard_fence(double src) { asm volatile("s_nop 1" ::"v"(src)); }
// compile-time loop (DPP controls are immediates)
template <int B, int E, class F> UKFB_DEV void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
// xor-1, xor-2 inside quads, then mirrored halves: every lane of the row ends with the same bits.
UKFB_DEV float row_allreduce(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}
// fp64: a butterfly step costs three instructions (two 32-bit DPP moves and the add), but row_newbcast works on
// 64-bit operands, fused into the FMA.  Two butterfly steps leave the quad sums in every lane of a quad; the QUADS
// quad leaders are then added by broadcast: 6 + QUADS instructions instead of 12, the same bits on every lane.
// QUADS = 3 when lanes 12..15 are known to contribute nothing.
template <int QUADS = 4> UKFB_DEV double row_allreduce(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    dpp_hazard_fence(v);
    double t = row_bcast<0>(v);
    fmac_bcast<4>(t, v, 1.0);
    fmac_bcast<8>(t, v, 1.0);
    if constexpr (QUADS > 3) fmac_bcast<12>(t, v, 1.0);
    return t;
}

// K row sums at once (DPP butterflies).  An LDS transposition (lane c adds up component c) was measured for
// fp64, where a butterfly costs 12 VALU: it pays for the 12-wide mean of the prediction only; for K <= 6 the
// two extra LDS round trips cost more than the saved instructions (Orient fp64 -2.5 %).
template <class T, int K, int QUADS = 4> UKFB_DEV void row_allreduce_n(T (&v)[K]) {
#pragma unroll
    for (int c = 0; c < K; ++c) {
        if constexpr (sizeof(T) == 8) v[c] = row_allreduce<QUADS>(v[c]);
        else v[c] = row_allreduce(v[c]);
    }
}

// ---------------------------------------------------------------------------------------------
// manifold traits for the tuned kernel
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long nib(int i, int v) { return (unsigned long long)(v & 15) << (4 * i); }

template <class M> struct MT;
template <class T> struct MT<PoseM<T>> {
    static constexpr int Q = 3;    // stored offset of the quaternion
    static constexpr int RT = 3;   // tangent offset of the rotation
    static constexpr int TR = 2, TC = 3;  // covariance tile
    // stored index of measurement component k for model id 0..8 (PoseUKF.cpp:7-69), 15 = unused
    static constexpr unsigned long long SEL0 = nib(0, 0) | nib(1, 0) | nib(2, 2) | nib(3, 15) | nib(4, 7) | nib(5, 7) |
                                               nib(6, 9) | nib(7, 7) | nib(8, 10);
    static constexpr unsigned long long SEL1 = nib(0, 1) | nib(1, 1) | nib(2, 15) | nib(3, 15) | nib(4, 8) | nib(5, 8) |
                                               nib(6, 15) | nib(7, 12) | nib(8, 11);
    static constexpr unsigned long long SEL2 = nib(0, 2) | nib(1, 15) | nib(2, 15) | nib(3, 15) | nib(4, 9) | nib(5, 15) |
                                               nib(6, 15) | nib(7, 15) | nib(8, 12);
    static constexpr bool HAS_EUCLID_MEAS = true;
    static constexpr int ZCOLS = 6;   // tangent columns that can move an orientation-dependent measurement (p, q)
    // Prediction (PoseUKF.cpp:75-97): position and orientation (tangent 0..5) are nonlinear in the state; velocity and
    // angular velocity (6..11) are affine with unit scale (v + acc dt, omega).  Covariance work items of the 16 lanes
    // (first tangent row / column of the lane's TR x TC tile, one nibble per lane): lanes 0..9 = the five tiles of the
    // nonlinear 6x6 block, two lanes per tile (each sums half of the sigma points); lanes 10..15 = the cross block.
    static constexpr unsigned long long TILE_R = nib(0, 0) | nib(1, 0) | nib(2, 2) | nib(3, 2) | nib(4, 2) | nib(5, 2) | nib(6, 4) |
                                                 nib(7, 4) | nib(8, 4) | nib(9, 4) | nib(10, 6) | nib(11, 6) | nib(12, 8) |
                                                 nib(13, 8) | nib(14, 10) | nib(15, 10);
    static constexpr unsigned long long TILE_C = nib(0, 0) | nib(1, 0) | nib(2, 0) | nib(3, 0) | nib(4, 3) | nib(5, 3) | nib(6, 0) |
                                                 nib(7, 0) | nib(8, 3) | nib(9, 3) | nib(10, 0) | nib(11, 3) | nib(12, 0) |
                                                 nib(13, 3) | nib(14, 0) | nib(15, 3);
    static constexpr int WORK_LANES = 16;
    UKFB_DEV static T aff_scale(int, const ProcIn<T>&) { return T(1); }
    // orientation-dependent measurement (model id 3, PoseUKF.cpp:28-33): the quaternion itself.
    // qp / qm / q0: orientation of the +column, -column and centre sigma point.
    UKFB_DEV static void gen_measure(const T (&qp)[4], const T (&qm)[4], const T (&q0)[4], const T*, const T*, T,
                                     T (&zp)[4], T (&zm)[4], T (&z0)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { zp[k] = qp[k]; zm[k] = qm[k]; z0[k] = q0[k]; }
    }
};
template <class T> struct MT<OrientM<T>> {
    static constexpr int Q = 0, RT = 0, TR = 3, TC = 3;
    static constexpr unsigned long long SEL0 = 0, SEL1 = 0, SEL2 = 0;
    static constexpr bool HAS_EUCLID_MEAS = false;
    static constexpr int ZCOLS = 6;   // body velocity q^-1 v reads the orientation (0..2) and the velocity (3..5)
    // Prediction (OrientationUKF.cpp:12-32): orientation and velocity (tangent 0..5) are nonlinear; the biases (6..11)
    // and gravity (12) are affine with diagonal scale (1 - dt / tau), 1.  3x3 tiles: lanes 0..5 = the three tiles of the
    // nonlinear block, two lanes per tile; lanes 6..11 = the cross block; lanes 12..15 repeat lane 0 and store nothing.
    static constexpr unsigned long long TILE_R = nib(0, 0) | nib(1, 0) | nib(2, 3) | nib(3, 3) | nib(4, 3) | nib(5, 3) | nib(6, 6) |
                                                 nib(7, 6) | nib(8, 9) | nib(9, 9) | nib(10, 12) | nib(11, 12);
    static constexpr unsigned long long TILE_C = nib(0, 0) | nib(1, 0) | nib(2, 0) | nib(3, 0) | nib(4, 3) | nib(5, 3) | nib(6, 0) |
                                                 nib(7, 3) | nib(8, 0) | nib(9, 3) | nib(10, 0) | nib(11, 3);
    static constexpr int WORK_LANES = 12;
    UKFB_DEV static T aff_scale(int c, const ProcIn<T>& in) {
        return (c < 9) ? fma(in.dt, in.ninv_tau_g, T(1)) : ((c < 12) ? fma(in.dt, in.ninv_tau_a, T(1)) : T(1));
    }
    // velocityMeasurementModel (OrientationUKF.cpp:34-39): q.inverse() * v for the three sigma points;
    // the velocity (stored 4..6, tangent 3..5) comes straight from the mean staging and the factor column
    // (rn = 1 / |q|^2, passed in: the three sigma-point orientations of a lane are q0 times unit exponentials -- one reciprocal of
    // |q0|^2 serves all of them, to the rounding of the exponential's norm)
    UKFB_DEV static void body_vel(const T (&q)[4], const T (&v)[3], T rn, T (&z)[4]) {
        const T qi[4] = {-q[0] * rn, -q[1] * rn, -q[2] * rn, q[3] * rn};
        T r[3];
        quat_rotate(qi, v, r);
        z[0] = r[0]; z[1] = r[1]; z[2] = r[2]; z[3] = T(0);
    }
    UKFB_DEV static void gen_measure(const T (&qp)[4], const T (&qm)[4], const T (&q0)[4], const T* MUS, const T* colp,
                                     T w, T (&zp)[4], T (&zm)[4], T (&z0)[4]) {
        const T v0[3] = {MUS[4], MUS[5], MUS[6]};
        const T cv[3] = {colp[3] * w, colp[4] * w, colp[5] * w};
        const T vp[3] = {v0[0] + cv[0], v0[1] + cv[1], v0[2] + cv[2]};
        const T vm[3] = {v0[0] - cv[0], v0[1] - cv[1], v0[2] - cv[2]};
        const T rn = fast_rcp(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
        body_vel(qp, vp, rn, zp);
        body_vel(qm, vm, rn, zm);
        body_vel(q0, v0, rn, z0);
    }
};

// | |q|^2 - 1 | up to which an isotropic noise block is not rotated (fp32 engines; 1e-9 in fp64)
constexpr float ISO_TOL_F32 = 1e-4f;
template <class T, class M> struct Layout16 {
    static constexpr int VEC = 16 / int(sizeof(T));
    static constexpr int D = M::D, S = M::S, N = 2 * D + 1, PK = D * (D + 1) / 2;
    // column stride of the factor = row stride of the deltas.  Lane l addresses row / column l, so the stride
    // in dwords decides the bank conflicts of every lane-strided access (measured on gfx950: 32 banks for b32,
    // 64 for b64/b128 accesses - tools/lds_probe.hip).
    // 14 scalars = 14 / 28 dwords: lanes 0..15 fall on distinct banks in both precisions (a stride of 12 or 16
    // scalars repeats after 8 resp. 2 lanes; measured +3..10 % on the fused cycle).
    static constexpr int LS = 14;
    static constexpr int al(int x) { return (x + VEC - 1) / VEC * VEC; }
    static constexpr int PKP = al(PK);
    // Prediction tables.  Only the NL = 6 leading tangent components are nonlinear in the state for both models
    // (MT<M>), so the delta table holds 6 columns.  It stores the sigma-point deltas as half sums and half differences,
    //   rows 0..D-1: U_l = (delta+_l + delta-_l) / 2,  row D: delta_0 / sqrt(2),  rows D+1..2D: W_l = (delta+_l - delta-_l) / 2,
    // because 1/2 sum_i delta_i delta_i^T = sum over these rows of row row^T, and the W rows are at the same time the
    // left factor of the cross block (with the affine rows of the factor, LAF).  Row stride ST (6 / 7 scalars):
    // lane-strided b64 / b32 accesses of lanes 0..15 fall on distinct banks, and D - NL <= ST.  Row N of the table and
    // row D of LAF are all-zero, so every lane runs the same trip count D + 1 (the table has 2 (D + 1) rows).
    static constexpr int NL = 6, ST = (D == 12) ? 6 : 7, TRIP = D + 1;
    static constexpr int LC = 0;                            // D*LS : unscaled factor columns
    // delta table rows 0..N-1 alias the factor (dead once every lane holds its column); its zero row lies behind it
    static constexpr int TNL = (D * LS > N * ST) ? (D * LS - N * ST) : 0;
    // OrientationState fp64 (round 3): the same two trims bring its slice from 14 656 B to 14 016 B = 11 instead of 10
    // one-wave workgroups per CU (11 LDS allocation granules of 1 280 B; its 146 VGPRs allow 12)
    static constexpr bool ORIENT64_TRIM = (M::MODEL == 1 && sizeof(T) == 8);
    static constexpr bool LAF_ROW_D = !(M::MODEL == 0 || ORIENT64_TRIM);
    // fp32 Pose: the small regions packed to 8-byte instead of 16-byte boundaries bring the slice to 400 floats = 6400 B =
    // 5 allocation granules: 24 workgroups per CU = 6 wavefronts per SIMD (77 VGPRs allow it)
    static constexpr int alm(int x) { return (M::MODEL == 0 && sizeof(T) == 4) ? (x + 1) / 2 * 2 : al(x); }
    static constexpr int UEND = al(TNL + (N + 1) * ST);     // end of the factor / delta-table region
    static constexpr int PKS = UEND;                        // PKP scalars, packed lower triangle (survives the prediction)
    static constexpr int LAF = PKS + PKP;                   // (D+1)*ST : row l = scaled affine rows of column l of the factor.  Where !LAF_ROW_D its zero row D
                                                            // (read by the cross lanes in their last trip) is not stored: it aliases the
                                                            // mean staging behind it - finite values that meet the table's exact-zero row
    static constexpr int LAF_SZ = al((D + (LAF_ROW_D ? 1 : 0)) * ST);
    // Where the zero row D is not stored the cross lanes' last trip reads the ST scalars behind row D - 1: the mean staging
    // (finite) -- and, if D * ST is not a multiple of the alignment, the PAD scalars in between, which nothing else ever writes.
    // Uninitialised LDS times the table's exact zero is NaN if the leftover happens to be NaN or Inf: the pad is zeroed with
    // the factor rows (found by tests/fuzz_parity.py as a box-dependent Cholesky failure of one OrientationState filter).
    static constexpr int LAF_PAD = LAF_ROW_D ? 0 : (LAF_SZ - D * ST);
    static constexpr int MISC = LAF + LAF_SZ;
    static constexpr int MUS = MISC;                        // S  : mean staging
    static constexpr int ROT = MUS + alm(S);                // 9  : rotation matrix of the mean
    static constexpr int ZQ = ROT + alm(9);                 // 12 : z (3) + Q (9)
    static constexpr int NSH = ZQ + 12;                     // 21 : shaped process noise of the nonlinear block
    // SINK_IN_NSH: the store sink IS the shaped-noise table.  Safe because while that table is live (from its fill to the last
    // fetch of one of its entries) every sink store is a single scalar and goes to the table's spare slot NSH_SINK; all other
    // sink stores (up to S scalars from the sink's base) happen while the table is dead.
    static constexpr bool SINK_IN_NSH = ORIENT64_TRIM;
    static constexpr int NSH_SINK = SINK_IN_NSH ? (NSH + 21) : (NSH + alm(21));   // where stores inside the table's live window send their idle lanes
    static constexpr int DUM = SINK_IN_NSH ? NSH : (NSH + alm(21));   // S  : sink for lane-predicated stores (longest: a mean / a covariance row)
    static constexpr int PF_RAW = SINK_IN_NSH ? (NSH + alm(21)) : (DUM + alm(S));
    static_assert(!SINK_IN_NSH || (alm(21) > 21 && alm(21) >= S), "the sink fits the noise table, which has a spare slot");
    // LDS index (from the slice base) of covariance entry (r, c), c <= r
    __host__ __device__ static constexpr int cv(int r, int c) { return PKS + r * (r + 1) / 2 + c; }
    // Workgroups per CU follow the LDS allocation granule of 1280 B (measured, tools/lds_granule.hip; the occupancy API
    // assumes 512 B): the fp64 Pose slice must stay <= 12800 B per workgroup for 12 workgroups = 3 wavefronts per SIMD
    // (it was 13120 B = 11 workgroups per CU until round 2).
    // the four slices of a wavefront must not start on the same LDS bank (measured: a slice stride that is
    // a multiple of 32 dwords costs 25-50 %: every broadcast read becomes a 4-way conflict)
    static constexpr int PF = PF_RAW + (((PF_RAW * int(sizeof(T)) / 4) % 32 == 0) ? 2 * VEC : 0);
    static_assert(TNL + N * ST >= D * LS && D - NL <= ST && NL <= ST, "prediction tables");
    static_assert(PF % 2 == 0 && LS >= D && LS <= 16 && S <= 16 && D + 1 <= 16, "scratch layout");
};

template <class T, class M> constexpr int lds_bytes_per_filter16() { return Layout16<T, M>::PF * int(sizeof(T)); }

// ---------------------------------------------------------------------------------------------
// Lane constants of the prediction's covariance phase.  Which tile a lane owns, where its operands sit in the LDS
// slice and where its results go depends on the lane index alone -- not on the filter, not on the launch -- yet every
// wavefront used to re-derive it (nibble decodes, triangular indices, predicates, address selects: ~80 of the ~130
// non-FMA VALU instructions of the phase).  The values are tabulated at compile time instead: one row of byte offsets
// per lane, fetched with two / four vector loads from constant memory.
//   rd[l]  operands: row pointer, column pointer (LDS bytes from the slice base), the weight of the neighbour's half sum
//          (bit pattern of T(1) for the two lanes that share a tile of the nonlinear block, T(0) for a cross lane),
//          byte offsets of the lane's plain-noise entries (tile origin, affine entries)
//   wr[l]  results: LDS byte offset of every tile entry and affine entry this lane stores (entries it does not own --
//          beyond the diagonal, outside the matrix, the second half lane of a shared tile -- point at the store sink),
//          then the offsets the affine entries are read from.  Row 16 = every store to the sink: the row of a lane
//          whose filter does not commit its prediction, so that the stores need no predicate of their own.
// ---------------------------------------------------------------------------------------------
template <class T, class M, class TS = T> struct CovTab {
    using LY = Layout16<T, M>;
    static constexpr int D = M::D, NL = LY::NL, ST = LY::ST, TRIP = LY::TRIP, TR = MT<M>::TR, TC = MT<M>::TC;
    static constexpr int NAB = (D - NL) * (D - NL + 1) / 2, AEL = (NAB + 15) / 16;
    static constexpr int NRD = 8, NWR = 16, SZ = int(sizeof(T)), SZG = int(sizeof(TS));   // LDS scalars / HBM scalars
    static constexpr int RD_PR = 0, RD_PC = 1, RD_WS = 2, RD_NZ = 4, RD_ANZ = 5;   // rd row (RD_WS: one or two dwords)
    static constexpr int WR_TILE = 0, WR_AFF = TR * TC, WR_AFF_RD = TR * TC + AEL;  // wr row
    static_assert(RD_ANZ + AEL <= NRD && WR_AFF_RD + AEL <= NWR, "covariance lane tables");
    struct Tabs {
        uint32_t rd[16][NRD];
        uint32_t wr[17][NWR];
    };
    static constexpr int tri_r(int e) {
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        return r;
    }
    static constexpr Tabs make() {
        Tabs t{};
        const uint32_t sink = uint32_t(LY::DUM * SZ);
        for (int l = 0; l < 16; ++l) {
            const int lw = (l < MT<M>::WORK_LANES) ? l : 0;
            const int R0 = int((MT<M>::TILE_R >> (4 * lw)) & 15ull), C0 = int((MT<M>::TILE_C >> (4 * lw)) & 15ull);
            const bool is_cross = R0 >= NL;
            const int half = is_cross ? 0 : (lw & 1);
            t.rd[l][RD_PR] = uint32_t((is_cross ? (LY::LAF + (R0 - NL)) : (LY::TNL + half * (TRIP * ST) + R0)) * SZ);
            t.rd[l][RD_PC] = uint32_t((LY::TNL + ((is_cross || half) ? (TRIP * ST) : 0) + C0) * SZ);
            if (SZ == 8) {
                t.rd[l][RD_WS] = 0u;
                t.rd[l][RD_WS + 1] = is_cross ? 0u : 0x3FF00000u;
            } else {
                t.rd[l][RD_WS] = is_cross ? 0u : 0x3F800000u;
            }
            const int rc0 = R0 < D ? R0 : D - 1, cc0 = C0 < D ? C0 : D - 1;
            t.rd[l][RD_NZ] = uint32_t((rc0 * D + cc0) * SZG);
            const bool writer = (l < MT<M>::WORK_LANES) && (is_cross || half == 0);
            for (int i2 = 0; i2 < TR; ++i2)
                for (int j2 = 0; j2 < TC; ++j2) {
                    const int r = R0 + i2, c = C0 + j2;
                    const bool w = writer && r < D && c <= r;
                    t.wr[l][WR_TILE + i2 * TC + j2] = w ? uint32_t(LY::cv(r, c) * SZ) : sink;
                }
            for (int k = 0; k < AEL; ++k) {
                const int e = l + 16 * k;
                const bool v = e < NAB;
                const int rr = v ? tri_r(e) : 0, cc = v ? (e - rr * (rr + 1) / 2) : 0;
                const int ar = NL + rr, ac = NL + cc;
                t.rd[l][RD_ANZ + k] = uint32_t((ar * D + ac) * SZG);
                t.wr[l][WR_AFF + k] = v ? uint32_t(LY::cv(ar, ac) * SZ) : sink;
                t.wr[l][WR_AFF_RD + k] = uint32_t(LY::cv(ar, ac) * SZ);
            }
        }
        for (int k = 0; k < NWR; ++k) t.wr[16][k] = sink;
        for (int k = 0; k < AEL; ++k) t.wr[16][WR_AFF_RD + k] = uint32_t(LY::cv(NL, NL) * SZ);
        return t;
    }
    static constexpr Tabs tabs = make();
};

// The same idea for the OrientationState kernels (round 4), whose 3 x 3 tiles on a 13 x 13 matrix made the decoded form expensive (tiles
// that hang over the edge, rows of unequal length, scale classes of the affine block: ~215 integer / select instructions per
// wavefront, most of them at the 4-cycle rate).  Round 2 tried 32-bit tables -- six 16-byte loads per lane -- and lost 1.4 %; here
// every offset is 16 bits (LDS offsets inside one filter's slice, byte offsets inside one 13 x 13 noise table): 48 bytes per lane,
// three loads.
//   rd[l]   PR, PC (LDS bytes), FLAGS (bit 0: tile of the nonlinear block, i.e. the neighbour lane's half sum is added; bits 2..9:
//           scale class 0 = gyro-bias, 1 = acc-bias, 2 = one of the row / column of the lane's two affine entries), NZ (byte offset
//           of the tile origin in the noise table: the nine entries follow by immediate offsets, past the table's end for tiles
//           that hang over -- the engine pads its noise allocations, ukf_batch.hip), ANZ0, ANZ1 (the affine entries' noise)
//   wr[l]   nine tile store offsets (entries the lane does not own -- beyond the diagonal, outside the matrix, the second half
//           lane of a shared tile, lanes >= WORK_LANES: the noise table's spare slot), two affine store offsets (sink for an
//           entry past the triangle), two affine read offsets.  Row 16: every store to its sink (a filter that does not commit).
template <class T, class M, class TS = T> struct OCovTab {
    using LY = Layout16<T, M>;
    static constexpr int D = M::D, NL = LY::NL, ST = LY::ST, TRIP = LY::TRIP, TR = MT<M>::TR, TC = MT<M>::TC;
    static constexpr int NAB = (D - NL) * (D - NL + 1) / 2, AEL = (NAB + 15) / 16, SZ = int(sizeof(T)), SZG = int(sizeof(TS));
    static constexpr int RD_PR = 0, RD_PC = 1, RD_FLAGS = 2, RD_NZ = 3, RD_ANZ = 4, NRD = 8;
    static constexpr int WR_TILE = 0, WR_AFF = TR * TC, WR_AFF_RD = TR * TC + AEL, NWR = 16;
    static_assert(AEL == 2 && TR * TC + 2 * AEL <= NWR && RD_ANZ + AEL <= NRD, "OrientationState lane tables");
    struct Tabs {
        uint16_t rd[16][NRD];
        uint16_t wr[17][NWR];
    };
    static constexpr int tri_r(int e) {
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        return r;
    }
    static constexpr int scale_class(int c) { return c < 9 ? 0 : (c < 12 ? 1 : 2); }   // MT<OrientM>::aff_scale
    static constexpr Tabs make() {
        Tabs t{};
        const uint16_t tile_sink = uint16_t(LY::NSH_SINK * SZ), aff_sink = uint16_t(LY::DUM * SZ);
        for (int l = 0; l < 16; ++l) {
            const int lw = (l < MT<M>::WORK_LANES) ? l : 0;
            const int R0 = int((MT<M>::TILE_R >> (4 * lw)) & 15ull), C0 = int((MT<M>::TILE_C >> (4 * lw)) & 15ull);
            const bool is_cross = R0 >= NL;
            const int half = is_cross ? 0 : (lw & 1);
            t.rd[l][RD_PR] = uint16_t((is_cross ? (LY::LAF + (R0 - NL)) : (LY::TNL + half * (TRIP * ST) + R0)) * SZ);
            t.rd[l][RD_PC] = uint16_t((LY::TNL + ((is_cross || half) ? (TRIP * ST) : 0) + C0) * SZ);
            t.rd[l][RD_NZ] = uint16_t((R0 * D + C0) * SZG);
            unsigned flags = is_cross ? 0u : 1u;
            const bool writer = (l < MT<M>::WORK_LANES) && (is_cross || half == 0);
            for (int i2 = 0; i2 < TR; ++i2)
                for (int j2 = 0; j2 < TC; ++j2) {
                    const int r = R0 + i2, c = C0 + j2;
                    t.wr[l][WR_TILE + i2 * TC + j2] = (writer && r < D && c <= r) ? uint16_t(LY::cv(r, c) * SZ) : tile_sink;
                }
            for (int k = 0; k < AEL; ++k) {
                const int e = l + 16 * k;
                const bool v = e < NAB;
                const int rr = v ? tri_r(e) : 0, cc = v ? (e - rr * (rr + 1) / 2) : 0;
                const int ar = NL + rr, ac = NL + cc;
                t.rd[l][RD_ANZ + k] = uint16_t((ar * D + ac) * SZG);
                t.wr[l][WR_AFF + k] = v ? uint16_t(LY::cv(ar, ac) * SZ) : aff_sink;
                t.wr[l][WR_AFF_RD + k] = uint16_t(LY::cv(ar, ac) * SZ);
                flags |= unsigned(scale_class(ar)) << (2 + 4 * k);
                flags |= unsigned(scale_class(ac)) << (4 + 4 * k);
            }
            t.rd[l][RD_FLAGS] = uint16_t(flags);
        }
        for (int k = 0; k < NWR; ++k) t.wr[16][k] = (k < WR_AFF) ? tile_sink : aff_sink;
        for (int k = 0; k < AEL; ++k) t.wr[16][WR_AFF_RD + k] = uint16_t(LY::cv(NL, NL) * SZ);
        return t;
    }
    static constexpr Tabs tabs = make();
};

// Lane constants of the update's assembly (every kernel): LDS byte offset of the lane's covariance row and of its three cross
// entries with the rotation columns (symmetric position (max, min)) -- one 16-byte load per lane instead of ~20 integer
// instructions of triangular indexing per wavefront.
template <class T, class M> struct AsmTab {
    using LY = Layout16<T, M>;
    struct Tabs { uint32_t off[16][4]; };
    static constexpr Tabs make() {
        Tabs t{};
        constexpr int RT = MT<M>::RT, SZ = int(sizeof(T));
        for (int l = 0; l < 16; ++l) {
            const int lr = l < M::D ? l : 0;
            t.off[l][0] = uint32_t(LY::cv(lr, 0) * SZ);
            for (int k = 0; k < 3; ++k) {
                const int hi = lr > RT + k ? lr : RT + k, lo = lr > RT + k ? RT + k : lr;
                t.off[l][1 + k] = uint32_t(LY::cv(hi, lo) * SZ);
            }
        }
        return t;
    }
    static constexpr Tabs tabs = make();
};

// ---------------------------------------------------------------------------------------------
// Cholesky: lane l < D holds row l (entries 0..l) in a[].  The factorisation itself runs on DPP row
// broadcasts (lane c's A[c][k] for the trailing update, lane k's pivot); column k is published UNSCALED
// (v_l = a_l[k], pivot included) with one LDS write for the consumers.  L[c][k] = Lc[k*LS+c]*rs_k
// for c >= k (entries above the diagonal are garbage and must be masked by the consumer).
// Returns this lane's rs_l (lane l < D); ok = all pivots > 0.
// ---------------------------------------------------------------------------------------------
// scheduling fence: keeps the machine scheduler from hoisting the next phase's loads / ALU work
// across this point (it otherwise trades ~2x the registers for ILP and ends up spilling)
UKFB_DEV void sfence() { __builtin_amdgcn_sched_barrier(0); }
// Issue priority of the wavefront (s_setprio): the factorisations are serial dependency chains; a wavefront inside one should
// not queue behind the throughput-bound phases of its SIMD neighbours.
// Measured (same-box A/B, 300 steps): priority 1 inside the two factorisations +0.8 % fp64, +2.5..3 % fp32; raising the mean
// iteration or the whole update as well brought nothing more.
#define UKFB_PRIO(p) __builtin_amdgcn_s_setprio(p)
// Phase markers (diagnostic builds only; the product build defines none of these macros and they vanish):
//   -DUKFB_PHASE_MARKS  an assembly comment between two scheduling barriers, read by tools/isa_phases.py
//   -DUKFB_STAMPS       lane 0 stores s_memtime to KArgs::stamps[workgroup][marker] (tools/phase_stamps.py:
//                       time per phase of a wavefront's life, the dynamic counterpart of the static listing)
//   -DUKFB_COUNTS       lane 0 stores value + 1 to KArgs::stamps[workgroup][slot] (tools/trip_counts.py: trip counts of the
//                       manifold-mean iteration per wavefront, which path the final deltas took)
#if defined(UKFB_COUNTS)
#define UKFB_COUNT_VAL(slot, v) do { if (threadIdx.x == 0) a.stamps[size_t(blockIdx.x) * UKFB_MAX_STAMPS + (slot)] = (unsigned long long)(v) + 1ull; } while (0)
#else
#define UKFB_COUNT_VAL(slot, v) do { } while (0)
#endif
#if defined(UKFB_STAMPS)
enum { UKFB_STAMP_BASE = __COUNTER__ };
#define UKFB_MARK(name)                                                                                          \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                              \
        if (threadIdx.x == 0) a.stamps[size_t(blockIdx.x) * UKFB_MAX_STAMPS + (__COUNTER__ - UKFB_STAMP_BASE - 1)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
#elif defined(UKFB_PHASE_MARKS)
#define UKFB_MARK(name)                                         \
    do {                                                        \
        __builtin_amdgcn_sched_barrier(0);                      \
        asm volatile("; @@PHASE " name);                        \
        __builtin_amdgcn_sched_barrier(0);                      \
    } while (0)
#else
#define UKFB_MARK(name) do { } while (0)
#endif
// Keeps a loaded value live in a VGPR so that "cond ? loaded : other" stays a v_cndmask; without it the
// compiler sinks the LDS load into an exec-masked branch (s_and_saveexec / s_cbranch_execz per element).
UKFB_DEV void keep(float& x) { asm volatile("" : "+v"(x)); }
UKFB_DEV void keep(double& x) { asm volatile("" : "+v"(x)); }

// KS < D factorises the first KS columns only (consumers that need no more; pivots KS.. are then not checked).
// PUB <= KS publishes the first PUB columns only (the update's applyDelta reads no others; all KS pivots are still
// computed and checked).  The columns are published after the last step: column k is final once step k has run, the
// row stays in registers anyway, and the compiler pairs the stores (ds_write2) when they stand together.
template <class T, int D, int LS, int KS = D, int PUB = KS> UKFB_DEV T chol16(T (&a)[D], T* Lc, int l, bool& ok) {
    static_assert(PUB <= KS && KS <= D, "chol16 column counts");
    bool good = true;
    // lanes >= D carry a copy of row D-1 (load_row clamps) and store the same values to the same addresses
    const int lw = (l < D) ? l : (D - 1);
    static_for<0, KS>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        dpp_hazard_fence(a[k]);   // written by the previous step's fused FMA, read through DPP from here on
        if constexpr (k >= PUB) {   // not published: checked as it comes (the published pivots are checked once, below)
            const T akk = row_bcast<k>(a[k]);
            good = good && (akk > T(0));
        }
        const T nt = -(a[k] * rcp_bcast<k>(a[k]));   // trailing update needs 1/pivot only; 1/sqrt is taken once, at the end
        if constexpr (k + 1 < KS) {
            static_for<k + 1, KS>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                fmac_bcast<c>(a[c], a[k], nt);   // a[c] -= t * A[c][k], lane c holds A[c][k]
            });
        }
    });
    // rows above the pivot publish an exact zero, so consumers can read whole columns unmasked
#pragma unroll
    for (int k = 0; k < PUB; ++k) Lc[k * LS + lw] = (l >= k) ? a[k] : T(0);
    wsync();
    const int lc = (l < PUB) ? l : (PUB - 1);
    const T pv = Lc[lc * LS + lc];
    // pivot k is final once step k - 1 has run and is what column k publishes on its diagonal: lane l < PUB checks its own,
    // one compare for all of them instead of one broadcast + compare per step (lanes >= PUB of every row: a constant mask)
    constexpr unsigned long long NO_PIVOT = ((0xFFFFull << PUB) & 0xFFFFull) * 0x0001000100010001ull;
    ok = good && row_all(NO_PIVOT | lanes_gt(pv, T(0)));
    return fast_rsqrt(pv);   // this lane's column scale 1/sqrt(pivot_l) (lanes >= PUB: the last published one)
}

// Row l of a packed lower-triangular matrix for chol16: one base address and immediate offsets, no selects.  The
// entries beyond the diagonal are whatever follows the row in the packed array (finite; chol16 updates them along
// with the rest -- they cannot be kept at zero, steps k <= l touch them -- and masks them when it publishes a column).
template <class T, int D> UKFB_DEV void load_row(const T* PKS, int l, T (&row)[D]) {
    const int lr = (l < D) ? l : (D - 1);
    const T* p = PKS + lr * (lr + 1) / 2;
    // the last row is complete, a shorter row l reads at most index l (l + 1) / 2 + D - 1 < D (D + 1) / 2
#pragma unroll
    for (int j = 0; j < D; ++j) row[j] = p[j];
}

// scaled column l of the factor (the stored column already has zeros above the diagonal);
// lanes without a column (l >= D) get zeros through w = 0
template <class T, int D, int LS> UKFB_DEV void load_column(const T* Lc, int l, T rs, T (&col)[D]) {
    const int lc = (l < D) ? l : (D - 1);
    const T w = (l < D) ? rs : T(0);
#pragma unroll
    for (int c = 0; c < D; ++c) col[c] = Lc[lc * LS + c] * w;
}

// fast boxminus of the SO(3) component only: log(conj(y) * x)
template <class T> UKFB_DEV void rot_minus(const T (&qx)[4], const T (&qy)[4], T (&r)[3]) {
    const T oc[4] = {-qy[0], -qy[1], -qy[2], qy[3]};
    T d[4];
    quat_mul(oc, qx, d);
    so3_log_fast(d, r);
}

// log(conj(y) * xa), log(conj(y) * xb) with the norm of the products known: both at once (so3_log_fast_n2)
template <class T> UKFB_DEV void rot_minus_n2(const T (&qxa)[4], const T (&qxb)[4], const T (&qy)[4], T nrm, T (&ra)[3], T (&rb)[3]) {
    const T oc[4] = {-qy[0], -qy[1], -qy[2], qy[3]};
    T da[4], db[4];
    quat_mul(oc, qxa, da);
    quat_mul(oc, qxb, db);
    // fp32 (issue-bound, registers to spare at 6 wavefronts per SIMD): the paired form, +0.7 ... 1.2 % on configs 3 / 4.  fp64: the
    // ten registers it holds more through the phase cost more than the interleaving gains (-0.6 % on the headline, same-box A/B).
    if constexpr (sizeof(T) == 4) {
        so3_log_fast_n2(da, db, nrm, ra, rb);
    } else {
        so3_log_fast_n(da, nrm, ra);
        so3_log_fast_n(db, nrm, rb);
    }
}

// the same with the norm of conj(y) * x known (see so3_log_fast_n)
template <class T> UKFB_DEV void rot_minus_n(const T (&qx)[4], const T (&qy)[4], T nrm, T (&r)[3]) {
    const T oc[4] = {-qy[0], -qy[1], -qy[2], qy[3]};
    T d[4];
    quat_mul(oc, qx, d);
    so3_log_fast_n(d, nrm, r);
}

// process models with the fast exp (same statements as PoseM/OrientM::process)
template <class T> UKFB_DEV void process_fast(PoseM<T>*, T (&x)[13], const ProcIn<T>& in) {
    x[7] += in.adt[0];
    x[8] += in.adt[1];
    x[9] += in.adt[2];
    T q[4] = {x[3], x[4], x[5], x[6]};
    const T v[3] = {x[7], x[8], x[9]}, w[3] = {x[10], x[11], x[12]};
    T rv[3], rw[3], e[4], r[4];
    quat_rotate(q, v, rv);
    x[0] += in.dt * rv[0]; x[1] += in.dt * rv[1]; x[2] += in.dt * rv[2];
    quat_rotate(q, w, rw);
    so3_exp_fast(rw, in.dt, e);
    quat_mul(q, e, r);
    x[3] = r[0]; x[4] = r[1]; x[5] = r[2]; x[6] = r[3];
}
template <class T> UKFB_DEV void process_fast(OrientM<T>*, T (&x)[14], const ProcIn<T>& in) {
    T q[4] = {x[0], x[1], x[2], x[3]};
    const T t[3] = {in.w[0] - x[7], in.w[1] - x[8], in.w[2] - x[9]};
    T av[3], e[4], r[4];
    quat_rotate(q, t, av);
    av[0] -= in.earth[0]; av[1] -= in.earth[1]; av[2] -= in.earth[2];
    so3_exp_fast(av, in.dt, e);
    quat_mul(q, e, r);
    x[0] = r[0]; x[1] = r[1]; x[2] = r[2]; x[3] = r[3];
    const T u[3] = {in.a[0] - x[10], in.a[1] - x[11], in.a[2] - x[12]};
    T acc[3];
    quat_rotate(r, u, acc);
    acc[2] -= x[13];
    x[4] += in.dt * acc[0]; x[5] += in.dt * acc[1]; x[6] += in.dt * acc[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x[7 + k] += in.dt * (in.ninv_tau_g * x[7 + k]);
        x[10 + k] += in.dt * (in.ninv_tau_a * x[10 + k]);
    }
}

// q * e and q * conj(e) for e = (u, c): A = c q and B = q * (u, 0) serve both, A + B and A - B (24 operations instead of 32)
template <class T> UKFB_DEV void quat_mul_pm(const T (&q)[4], const T (&e)[4], T (&rp)[4], T (&rm)[4]) {
    const T c = e[3];
    const T bx = q[QW] * e[0] + q[QY] * e[2] - q[QZ] * e[1];
    const T by = q[QW] * e[1] + q[QZ] * e[0] - q[QX] * e[2];
    const T bz = q[QW] * e[2] + q[QX] * e[1] - q[QY] * e[0];
    const T bw = -(q[QX] * e[0] + q[QY] * e[1] + q[QZ] * e[2]);
    rp[QX] = fma(c, q[QX], bx); rm[QX] = fma(c, q[QX], -bx);
    rp[QY] = fma(c, q[QY], by); rm[QY] = fma(c, q[QY], -by);
    rp[QZ] = fma(c, q[QZ], bz); rm[QZ] = fma(c, q[QZ], -bz);
    rp[QW] = fma(c, q[QW], bw); rm[QW] = fma(c, q[QW], -bw);
}

// sigma pair mu [+] (+col), mu [+] (-col): one exp serves both points, exp(-v) = conj(exp(v))
template <class T, class M>
UKFB_DEV void sigma_pair(const T (&mu)[M::S], const T (&col)[M::D], T (&xp)[M::S], T (&xm)[M::S]) {
    constexpr int Q = MT<M>::Q, RT = MT<M>::RT, S = M::S;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s < Q) { xp[s] = mu[s] + col[s]; xm[s] = mu[s] - col[s]; }
        else if (s >= Q + 4) { xp[s] = mu[s] + col[s - 1]; xm[s] = mu[s] - col[s - 1]; }
    }
    const T q[4] = {mu[Q], mu[Q + 1], mu[Q + 2], mu[Q + 3]};
    const T v[3] = {col[RT], col[RT + 1], col[RT + 2]};
    T ep[4], rp[4], rm[4];
    so3_exp_fast(v, T(1), ep);
    quat_mul_pm(q, ep, rp, rm);
#pragma unroll
    for (int k = 0; k < 4; ++k) { xp[Q + k] = rp[k]; xm[Q + k] = rm[k]; }
}

// One entry of the shaped process noise R without exec-masked regions (cf. process_noise_entry).
template <class T, class M, class TS>
UKFB_DEV T process_noise_entry16(const TS* Rn, const TS* Racc, const T* ROT, const ProcIn<T>& pin, int r, int c) {
    constexpr int D = M::D;
    T vacc = T(0);
    if (M::MODEL == 0) {
        // acceleration branch (PoseUKF.cpp:190-191): raw noise with block(6,6,3,3) = 2 acc.cov, prepared by
        // the host (ukf_batch.hip: rebuild_racc) whenever the noise or acc.cov changes
        vacc = T(Racc[r * D + c]);
    }
    const T rn = T(Rn[r * D + c]);
    const int o = (r < 3 && c < 3) ? 0 : ((r >= 3 && r < 6 && c >= 3 && c < 6) ? 3 : -1);
    const int oo = o < 0 ? 0 : o;
    const int rr = (o < 0) ? 0 : (r - oo), cc = (o < 0) ? 0 : (c - oo);
    T acc = T(0);
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        T tmp = T(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) tmp += ROT[rr * 3 + k] * T(Rn[(oo + k) * D + (oo + m)]);
        acc += tmp * ROT[cc * 3 + m];
    }
    const T val = (o >= 0) ? acc : rn;
    const T scale = (M::MODEL == 0) ? pin.dt : pin.dt * pin.dt;
    const T vcv = scale * val;
    return (M::MODEL == 0 && pin.use_acc) ? vacc : vcv;
}

// The same for an entry OUTSIDE the two rotated 3x3 diagonal blocks (cross and affine blocks): no rotation.
template <class T, class M, class TS> UKFB_DEV T plain_noise_entry16(const TS* Rn, const TS* Racc, const ProcIn<T>& pin, int r, int c) {
    constexpr int D = M::D;
    const T rn = T(Rn[r * D + c]);
    if (M::MODEL == 0) {
        const T vacc = T(Racc[r * D + c]);
        return pin.use_acc ? vacc : pin.dt * rn;
    }
    return (pin.dt * pin.dt) * rn;
}

// (row, column) of packed lower-triangle entries first .. first + 15, one nibble per entry (entries >= 28 read as 0)
constexpr unsigned long long tri_rows(int first) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; ++i) {
        const int e = first + i;
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        t |= (unsigned long long)((e < 28 ? r : 0) & 15) << (4 * i);
    }
    return t;
}
constexpr unsigned long long tri_cols(int first) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; ++i) {
        const int e = first + i;
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        t |= (unsigned long long)((e < 28 ? e - r * (r + 1) / 2 : 0) & 15) << (4 * i);
    }
    return t;
}

// minimum waves per SIMD the register allocator must leave room for (LDS admits 3 in fp64, 6 in fp32), per instantiation
template <class T> constexpr int min_waves16() { return sizeof(T) == 8 ? 2 : 5; }
template <class T, class M, bool MULTI> constexpr int waves16() {
    if (sizeof(T) == 4) return MULTI ? 4 : min_waves16<T>();
    if (M::MODEL != 0 && MULTI) return 3;   // OrientationState fp64, multi-cycle: three wavefronts per SIMD (192 VGPRs = two before round 3's loop hygiene)
    return min_waves16<T>();
}

// Element idx of an array entered through a scalar base: the byte offset is formed in 32 bits, so that the load / store takes
// the base as its scalar operand and the offset as its 32-bit vector operand (base[idx] would widen idx first and add in 64 bits)
// Workgroup b of a launch runs on XCD b % 8 (round-robin dispatch).  The groups of four filters are
// renumbered so that every XCD works through ONE contiguous eighth of the per-filter arrays (its own pages, its own L2
// lines) instead of every eighth group of all of them.  Scalar arithmetic; a bijection for any grid size.  Same-box A/B:
// fp32 +0.7 % (1 M filters), +0.9 % (131 072), config 4 +1.3 %; the fp64 configurations unchanged.
__host__ UKFB_DEV unsigned group_of_block(unsigned b, unsigned nb) {
    const unsigned q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

template <class P> UKFB_DEV P* at32(P* base, uint32_t idx) {
    using B = std::conditional_t<std::is_const_v<P>, const unsigned char, unsigned char>;
    return reinterpret_cast<P*>(reinterpret_cast<B*>(base) + idx * uint32_t(sizeof(P)));
}

// MULTI (fused cycle only): KArgs::cyc_count consecutive cycles in one launch.  The filter is staged in LDS once, every
// cycle reads its own input slot (requested one cycle ahead), the state goes back to HBM after the last cycle; the status
// word is the OR over the cycles.  MULTI = false compiles the single-cycle kernel exactly as before (trip count 1).
// INDIRECT (fused cycle only): work item i acts on filter fidx[i] (event rounds).  Direct launches -- everything else --
// have the four filters of a workgroup side by side in every per-filter array: their addresses are a SCALAR base per
// workgroup (64-bit arithmetic on the scalar unit) plus a small 32-bit lane offset, instead of a 64-bit multiply-add per
// lane and stream (8 v_mad_u64_u32, 2 v_mul_lo_u32, 7 v_lshl_add_u64 per wavefront; 45 instructions fewer in all).  Measured:
// +0.7 % fp64, nothing in fp32 -- the prologue is not where a wavefront's time goes.
// PLAINL ("plain launch": fused cycles, direct; also as a multi-cycle launch without a schedule): what the host knows about a launch of the common fixed-rate case becomes the kernel's TYPE --
// one time step and one measurement model for the whole launch, no per-filter timestamps / time steps / model ids / activity
// flags, the accept-any gate of the reference (PoseUKF.cpp:116), a fresh status word; for PoseWithVelocity the model is one
// of the three full 3-vector selections (position, velocity, angular velocity).  The other measurement paths, the gate
// arithmetic on per-filter streams and their loads are not in this kernel at all: 136 instead of 140 VGPRs (fp64 Pose), much less
// code: +3.0 % on the fp64 headline, +2.8 % fp32, +2.5 % config 4, +2 % OrientationState fp64 (same-box A/B, DESIGN.md
// section 4.10).  The launcher picks it when all of that holds
// (ukf_launch.inc.hpp); the arithmetic of a filter is the general kernel's, bit for bit (tests/test_gpu_plain_kernel.py).
// TS (storage type, default T): the type of every per-filter array and noise table in HBM.  TS = float with T = double is the
// WIDE-ARITHMETIC mode of the fp32 engines (ukfb_config::wide_arithmetic): fp32 state and inputs in HBM, every instruction of
// the cycle in fp64 (values widen on load and narrow on commit; the LDS slice is the fp64 one).  Why that and not a cheaper mix:
// tests/study_f32_mixed.py / profiles/r04_f32_mixed_ab.txt -- over the bench's run length any stage left in fp32 (the SO(3)
// maps or the factorisations / recombinations) keeps the OrientationState mean at 5e-4 ... 9e-4 from the fp64 algorithm.
// PLAIN is a level: 0 = the general kernel; 1 = STREAMS ONLY (round 4): no per-filter timestamps / time steps / activity flags,
// accept-any gate, fresh status word -- but per-filter model ids are allowed (model-class buckets of a mixed stream, any uniform
// model); 2 = the plain launch described above (level 1 + one full-3-vector model for the launch).  Levels 1 and 2 also exist
// for prediction-only and update-only launches (callers that keep the reference's two calls, UnscentedKalmanFilter.hpp:107-125
// then PoseUKF.cpp:112-173) and level 1 for indirect launches over a bucketed filter list.
template <class T, class M, bool DO_PREDICT, bool DO_UPDATE, bool MULTI = false, bool INDIRECT = false, int PLAIN = 0, class TS = T>
__global__ void __launch_bounds__(64, (waves16<T, M, MULTI>())) ukf_kernel16(const KArgs<TS> a) {
    constexpr bool STREAMS_ONLY = PLAIN >= 1, PLAINL = PLAIN == 2;
    static_assert(!MULTI || (DO_PREDICT && DO_UPDATE), "multi-cycle launches run the fused cycle");
    static_assert(!INDIRECT || (DO_PREDICT && DO_UPDATE && !MULTI), "indirect launches run the single fused cycle");
    static_assert(!PLAINL || !INDIRECT, "a plain launch is direct (one model for every filter)");
    static_assert(PLAIN >= 0 && PLAIN <= 2, "plain level");
    constexpr int S = M::S, D = M::D, N = 2 * D + 1, PK = D * (D + 1) / 2;
    using LY = Layout16<T, M>;
    constexpr int LS = LY::LS, Q = MT<M>::Q, RT = MT<M>::RT, TR = MT<M>::TR, TC = MT<M>::TC;
    constexpr int G = 16, FPW = 4, EPL = (PK + G - 1) / G;
    static_assert(D + 1 <= G && S <= G, "a filter must fit one DPP row");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if constexpr (!INDIRECT) __builtin_assume(a.fidx == nullptr);
    // Optional per-filter streams.  A multi-cycle launch has none of them BY TYPE (a compile-time null, not an assumption about
    // a kernel argument: the loads of an argument inside the cycle loop are not the load the assumption was made about, and
    // the timestamp path stayed alive in the multi-cycle kernels -- registers around the whole loop, and a wavefront per SIMD
    // in the OrientationState fp64 kernel).
    const int64_t* const a_ts = (MULTI || STREAMS_ONLY) ? nullptr : a.ts;
    const double* const a_dt = (MULTI || STREAMS_ONLY) ? nullptr : a.dt;
    const uint8_t* const a_active = (MULTI || STREAMS_ONLY) ? nullptr : a.active;
    const int32_t* const a_meas = PLAINL ? nullptr : a.meas;
    const T gate_chi2_c = STREAMS_ONLY ? T(-1) : T(a.gate_chi2);
    const bool status_accumulate_c = STREAMS_ONLY ? false : (a.status_accumulate != 0);
    if constexpr (MULTI) {
        // multi-cycle launches are direct launches with one dt per cycle for every filter (checked by the host): no
        // per-filter timestamps, time steps, activity flags or filter index list to keep alive
        __builtin_assume(a.fidx == nullptr); __builtin_assume(a.ts == nullptr); __builtin_assume(a.dt == nullptr);
        __builtin_assume(a.active == nullptr);   // (per-filter model ids are allowed: a ring like z and Q, negative = none)
    }
    const int lane = threadIdx.x;
    // (not const: a multi-cycle launch re-derives everything that follows from the lane index in every cycle, see below)
    int g = lane >> 4, l = lane & 15;
    // Indices of this row's work item (fi: per-call inputs) and filter (fc: engine state).  Direct launches: relative to the
    // workgroup's first work item wg0 -- 0..3, 32 bits -- and every per-filter array below is entered at wg0 (scalar);
    // indirect launches: absolute, 64 bits, the arrays as they are.
    using IDX = std::conditional_t<INDIRECT, int64_t, uint32_t>;
    const int64_t wg0 = INDIRECT ? int64_t(0) : a.item0 + int64_t(group_of_block(blockIdx.x, gridDim.x)) * FPW;
    const int64_t n_here = a.n - wg0;                                   // work items from wg0 on (scalar)
    const int n_wg = int(n_here < FPW ? n_here : int64_t(FPW));          // ... of this workgroup, direct launches
    IDX f = INDIRECT ? IDX(int64_t(blockIdx.x) * FPW + g) : IDX(g);
    bool fvalid = INDIRECT ? (int64_t(f) < a.n) : (g < n_wg);
    IDX fi = fvalid ? f : (INDIRECT ? IDX(a.n - 1) : IDX(n_wg - 1));
    IDX fc = INDIRECT ? IDX(a.fidx[fi]) : fi;
    if constexpr (INDIRECT) {
        // model-class buckets (ukfb_cycle_dev with per-filter model ids): the list is ordered by class and padded to whole
        // wavefronts with -1; z, Q and the model ids are the caller's per-filter arrays
        if (a.fidx_inputs) {
            fvalid = fvalid && (fc >= 0);
            fc = (fc >= 0) ? fc : IDX(0);
            fi = fc;
        }
    }
    if constexpr (!INDIRECT) __builtin_assume(fi < 4u && fc < 4u);
    const auto at_wg = [&](auto* p, int64_t stride) { return p ? p + wg0 * stride : p; };   // scalar pointer arithmetic
    // element idx of a per-filter array (idx includes the row's fi / fc)
    const auto at = [](auto* base, IDX idx) {
        if constexpr (INDIRECT) return base + idx;
        else return at32(base, idx);
    };
    // (the phases far from the prologue -- process noise, commit -- form their bases again from an opaque copy of the
    // workgroup index: kept alive from here, fifteen pointers cost more registers than the arithmetic they save)
    const auto wg0_again = [&]() {
        unsigned b = blockIdx.x;
        asm volatile("" : "+s"(b));
        return INDIRECT ? int64_t(0) : a.item0 + int64_t(group_of_block(b, gridDim.x)) * FPW;
    };
    TS* const mu_p = at_wg(a.mu, S);
    TS* const cov_p = at_wg(a.cov, PK);
    uint32_t* const status_p = at_wg(a.status, 1);
    const uint8_t* const init_p = at_wg(a.initialised, 1);
    int64_t* const last_ts_p = at_wg(a.last_ts, 1);
    const TS* const in_a_p = at_wg(a.in_a, 3);
    const TS* const in_b_p = at_wg(a.in_b, 3);
    const int64_t* const ts_p = at_wg(a_ts, 1);
    const double* const dt_p = at_wg(a_dt, 1);
    const int32_t* const meas_p = at_wg(a_meas, 1);
    const uint8_t* const active_p = at_wg(a_active, 1);
    const TS* const z_p = at_wg(a.z, 3);
    const IDX q_stride = (!MULTI && a.q_uniform) ? 0 : 9;     // batch-uniform measurement covariance: every filter reads the same 9 scalars
    const TS* const Q_p = at_wg(a.Q, int64_t(q_stride));
    T* base = reinterpret_cast<T*>(smem_raw) + g * LY::PF;
    T* Lc = base + LY::LC;
    T* TAB = base + LY::TNL;
    T* PKS = base + LY::PKS;
    T* LAF = base + LY::LAF;
    T* MUS = base + LY::MUS;
    T* ROT = base + LY::ROT;
    T* ZQ = base + LY::ZQ;
    T* NSH = base + LY::NSH;
    T* DUMP = base + LY::DUM;
    bool has_pair = l < D;       // lane owns the sigma pair of column l
    bool has_ctr = l == D;       // lane owns the centre point
    double dt_uniform_c = a.dt_uniform;   // launch-wide time step and measurement model (scalars)
    int meas_uniform_c = a.meas_uniform;

    UKFB_MARK("prologue");
    // ---- prologue: EVERY per-filter stream is requested before the first dependent instruction, so the kernel
    // pays the HBM latency once.  Optional streams (null pointer) are read from a substitute address that is
    // always valid and the value is dropped by a select - a branch here would serialise the round trips.
    const uint8_t init_b = *at(init_p, fc);
    int64_t last_l = 0, ts_l = 0;
    double dt_l = 0.0;
    if constexpr (DO_PREDICT) {
        // (optional streams: direct launches have fi == fc, the substitute is a scalar choice of the base)
        const int64_t* tsp = INDIRECT ? (ts_p ? at(ts_p, fi) : at(static_cast<const int64_t*>(last_ts_p), fc))
                                      : at(ts_p ? ts_p : static_cast<const int64_t*>(last_ts_p), fi);
        const double* dtp = INDIRECT ? (dt_p ? at(dt_p, fi) : reinterpret_cast<const double*>(at(static_cast<const int64_t*>(last_ts_p), fc)))
                                     : at(dt_p ? dt_p : reinterpret_cast<const double*>(last_ts_p), fi);
        last_l = *at(last_ts_p, fc);
        ts_l = *tsp;
        dt_l = *dtp;
    }
    int32_t mid_l = 0;
    uint8_t act_b = 1;
    if constexpr (DO_UPDATE) {
        const int32_t* mp = INDIRECT ? (meas_p ? at(meas_p, fi) : reinterpret_cast<const int32_t*>(at(static_cast<const uint32_t*>(status_p), fc)))
                                     : at(meas_p ? meas_p : reinterpret_cast<const int32_t*>(status_p), fi);
        const uint8_t* ap = INDIRECT ? (active_p ? at(active_p, fi) : at(init_p, fc)) : at(active_p ? active_p : init_p, fi);
        mid_l = *mp;
        act_b = *ap;
    }
    T cov_l[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int e = l + G * t;
        cov_l[t] = T(*at(static_cast<const TS*>(cov_p), fc * PK + IDX((e < PK) ? e : (PK - 1))));
    }
    const T mu_l = T(*at(static_cast<const TS*>(mu_p), fc * S + IDX((l < S) ? l : (S - 1))));
    ProcIn<T> pin;
    // per-call inputs of one input slot (single-cycle launches: slot 0 = the arrays themselves)
    const auto load_inputs = [&](int slot, T (&ia)[3], T (&ib)[3], T& zq, int32_t& mid_slot) {
        const int64_t so = MULTI ? int64_t(slot) * a.cyc_items : 0;   // scalar: the slot's offset goes to the scalar base
        if constexpr (DO_UPDATE) {
            const int32_t* mp = at(meas_p ? meas_p + so : reinterpret_cast<const int32_t*>(status_p), fi);   // (MULTI: direct, fi == fc)
            mid_slot = *mp;
        }
        if constexpr (DO_PREDICT) {
            // latched inputs: the engine always passes both arrays (its own or the bound ones)
            const TS* pa = at(in_a_p + ((MULTI && (a.cyc_in & 1)) ? so * 3 : 0), fc * 3);
            const TS* pb = at(in_b_p + ((MULTI && (a.cyc_in & 2)) ? so * 3 : 0), fc * 3);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ia[k] = T(pa[k]);
                ib[k] = T(pb[k]);
            }
        }
        if constexpr (DO_UPDATE) {
            const TS* zp = (l < 3) ? at(z_p + so * 3, fi * 3 + IDX(l)) : at(Q_p + so * 9, fi * 9 + IDX((l < 12) ? (l - 3) : 0));
            zq = T(*zp);
        }
    };
    T zq_l = T(0);
    int slot = MULTI ? a.cyc_first : 0;
    // fp64 (3 wavefronts per SIMD, latency-bound): a multi-cycle launch requests the inputs of the next cycle before this
    // cycle's arithmetic, into registers -- the first cycle's here, with the state (OrientationState fp64, same-box: 805 M
    // filter-cycles/s with the prefetch, 787 M without).  The fp32 multi-cycle kernels (6 per SIMD,
    // issue-bound) load every cycle's inputs at the head of that cycle: nothing is carried around the cycle loop.
    constexpr bool PREFETCH = MULTI && sizeof(T) == 8;
    if constexpr (MULTI) {
        if constexpr (PREFETCH) load_inputs(slot, pin.a, pin.w, zq_l, mid_l);
    } else {
        if constexpr (DO_PREDICT) {
            const TS* pa = at(in_a_p, fc * 3);
            const TS* pb = at(in_b_p, fc * 3);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pin.a[k] = T(pa[k]);
                pin.w[k] = T(pb[k]);
            }
        }
        if constexpr (DO_UPDATE) {
            const TS* zp = (l < 3) ? at(z_p, fi * 3 + IDX(l)) : at(Q_p, fi * q_stride + IDX((l < 12) ? (l - 3) : 0));
            zq_l = T(*zp);
        }
    }

    UKFB_MARK("stage");
    // ---- stage the filter in LDS: packed covariance, mean, measurement
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int e = l + G * t;
        PKS[(e < PK) ? e : (LY::DUM - LY::PKS)] = cov_l[t];
    }
    MUS[(l < S) ? l : (LY::DUM - LY::MUS)] = mu_l;
    const bool live = fvalid && (init_b != 0);
    uint32_t st_all = ST_OK;   // OR over the cycles of the launch
    bool changed = false;      // some cycle committed: the state goes back to HBM
    const int ncyc = MULTI ? a.cyc_count : 1;
    // ---- cycles of this launch (one, unless MULTI).  The body is not indented: it is the single-cycle kernel.
    int cyc = 0;
    do {   // (a do-while: with MULTI = false its condition is a constant and no loop exists at all)
    if constexpr (MULTI) {
        // Everything derived from the lane index is loop-invariant, and the compiler would hoist all of it -- addresses,
        // predicates, decoded tables: a hundred registers held across the whole cycle, spills in every MULTI kernel.  The
        // lane index is therefore passed through an opaque move in every cycle and its derivatives are formed again.
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        g = lane_c >> 4;
        l = lane_c & 15;
        f = IDX(g);
        fvalid = g < n_wg;
        fi = fvalid ? f : IDX(n_wg - 1);
        fc = fi;   // (multi-cycle launches are direct: no filter index list)
        base = reinterpret_cast<T*>(smem_raw) + g * LY::PF;
        Lc = base + LY::LC;
        TAB = base + LY::TNL;
        PKS = base + LY::PKS;
        LAF = base + LY::LAF;
        MUS = base + LY::MUS;
        ROT = base + LY::ROT;
        ZQ = base + LY::ZQ;
        NSH = base + LY::NSH;
        DUMP = base + LY::DUM;
        has_pair = l < D;
        has_ctr = l == D;
        // the same for the two launch-wide scalars whose derivatives (time gate, selection tables of the measurement
        // model) are evaluated with vector instructions: opaque scalar moves
        // (a schedule: this cycle's own dt and model; a plain multi-cycle launch has none)
        dt_uniform_c = (!PLAINL && a.cyc_sched) ? a.cyc_dt[cyc] : a.dt_uniform;
        meas_uniform_c = (!PLAINL && a.cyc_sched) ? a.cyc_model[cyc] : a.meas_uniform;
        asm volatile("" : "+s"(dt_uniform_c));
        asm volatile("" : "+s"(meas_uniform_c));
    }
    T nx_a[3] = {T(0), T(0), T(0)}, nx_w[3] = {T(0), T(0), T(0)}, nx_zq = T(0);
    int32_t nx_mid = 0;
    if constexpr (MULTI) {
        if constexpr (PREFETCH) {   // (the last cycle re-reads its own slot)
            const int nslot = (slot + 1 >= a.cyc_ring) ? 0 : (slot + 1);
            slot = (cyc + 1 < ncyc) ? nslot : slot;
            load_inputs(slot, nx_a, nx_w, nx_zq, nx_mid);
        } else {
            const int nslot = (slot + 1 >= a.cyc_ring) ? 0 : (slot + 1);
            slot = (cyc > 0) ? nslot : slot;
            load_inputs(slot, pin.a, pin.w, zq_l, mid_l);
        }
    }
    if constexpr (DO_UPDATE) ZQ[(l < 12) ? l : (LY::DUM - LY::ZQ)] = zq_l;

    uint32_t st = ST_OK;
    st |= (fvalid && !live) ? ST_UNINITIALISED : 0u;

    UKFB_MARK("gate");
    // ---- time gate (UnscentedKalmanFilter.hpp:83-125)
    bool do_p = false, p_error = false, ts_store = false;
    bool noev = false;   // ts < 0: this filter has no sample in this call -> neither predicted nor updated
    T dtT = T(0);
    if constexpr (DO_PREDICT) {
        const bool use_ts = a_ts != nullptr;
        noev = use_ts && (ts_l < 0);
        const bool first = use_ts && (last_l == 0) && !noev;
        double dt;
        if (!use_ts && !a_dt) {
            // one dt for the whole launch (kernel argument): the gate is scalar arithmetic, one select per lane
            dt = dt_uniform_c;
            const bool neg = dt < 0.0, small = dt <= a.min_dt, large = dt > a.max_dt;
            const uint32_t code = neg ? ST_ERR_NEG_DT : (small ? ST_SKIPPED_SMALL_DT : (large ? ST_ERR_DT_TOO_LARGE : 0u));
            st |= live ? code : 0u;
            p_error = live && (neg || (!small && large));
            do_p = live && code == 0u;
        } else {
            dt = a_dt ? dt_l : dt_uniform_c;
            if (use_ts)   // the IEEE division of base::Time::toSeconds only where it is needed
                dt = (first || noev) ? 0.0 : double(ts_l - last_l) / 1000000.0;
            ts_store = use_ts && live && l == 0 && !noev && (first || dt > a.min_dt);
            const bool neg = dt < 0.0, small = dt <= a.min_dt, large = dt > a.max_dt;
            const uint32_t code = first ? ST_SKIPPED_FIRST_TS
                                        : (neg ? ST_ERR_NEG_DT : (small ? ST_SKIPPED_SMALL_DT : (large ? ST_ERR_DT_TOO_LARGE : 0u)));
            st |= (live && !noev) ? code : 0u;
            p_error = live && !first && !noev && (neg || (!small && large));
            do_p = live && !noev && code == 0u;
        }
        dtT = T(dt);
        pin.dt = dtT;
        pin.ninv_tau_g = T(a.ninv_tau_g);
        pin.ninv_tau_a = T(a.ninv_tau_a);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pin.earth[k] = T(a.earth[k]);
        }
        pin.use_acc = m_finite(pin.a[0]) && m_finite(pin.a[1]) && m_finite(pin.a[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) pin.adt[k] = pin.use_acc ? pin.dt * pin.a[k] : T(0);
    }
    bool do_u = false;
    int mid = -1;
    if constexpr (DO_UPDATE) {
        mid = a_meas ? mid_l : meas_uniform_c;
        const bool act = PLAINL ? true : (M::meas_valid(mid) && (a_active ? act_b != 0 : true));   // (plain: the launcher checked the model id)
        do_u = live && act && !p_error && !noev;
        // (a scheduled prediction-only cycle of a multi-cycle launch is a plain predictionStep: no INACTIVE mark)
        const bool predict_only = MULTI && !PLAINL && a.cyc_sched != 0 && meas_uniform_c < 0;
        st |= (live && !do_u && !predict_only) ? ST_INACTIVE : 0u;
    }
    wsync();
    if constexpr (DO_PREDICT) {
        if (ts_store) *at(last_ts_p, fc) = ts_l;   // after every load of the prologue has been issued
    }

    bool p_commit = false, u_commit = false;

    // =========================================================================== predict
    if constexpr (DO_PREDICT) {
        if (wave_any(do_p)) {
            UKFB_MARK("p_chol");
            constexpr int NL = LY::NL, ST = LY::ST, TRIP = LY::TRIP;
            // stored index of a nonlinear Euclidean tangent component t (t outside [RT, RT + 3))
            constexpr auto st_of = [](int t) constexpr { return t < MT<M>::RT ? t : t + 1; };
            T xp[S], xm[S], ref[S];
            bool ok;
            bool noise_plain = false;   // wave-uniform: the process noise needs no rotation (see below)
            bool need_rot = false;      // wave-uniform: the rotated noise blocks are evaluated, they need the mean's rotation matrix
            T qn2 = T(1);   // |q|^2 of the mean's orientation = norm of every conj(a) * b between sigma-point orientations
            {
                T mu_r[S];
#pragma unroll
                for (int s = 0; s < S; ++s) mu_r[s] = MUS[s];
                qn2 = mu_r[Q] * mu_r[Q] + mu_r[Q + 1] * mu_r[Q + 1] + mu_r[Q + 2] * mu_r[Q + 2] + mu_r[Q + 3] * mu_r[Q + 3];
                // rotation matrix of the current mean (PoseUKF.cpp:182 / OrientationUKF.cpp:81); the Pose acceleration branch does
                // not rotate its noise (wave-uniform skip).  An isotropic block is its own rotation (R s I R^T = s I R R^T): with
                // the launch-wide flag set by the host and unit orientation quaternions (|q|^2 within 1e-9 of 1 on every lane,
                // 1e-4 in fp32 where the norm drifts; R R^T = I up to that) neither the rotation matrix nor the rotated entries
                // are evaluated, the shaped-noise table is filled with the plain entries instead
                // (OrientationState kernels only: in the Pose kernels the two extra branches cost the acceleration-branch
                // headline 0.6 % through code placement alone, same-box A/B)
                if constexpr (M::MODEL != 0)
                    noise_plain = a.noise_iso != 0 && wave_all(!do_p || m_abs(qn2 - T(1)) <= (sizeof(T) == 8 ? T(1e-9) : T(ISO_TOL_F32)));
                need_rot = !noise_plain && (M::MODEL != 0 || !wave_all(pin.use_acc));
                if (need_rot) {
                    T q[4], rot[9];
                    M::orientation(mu_r, q);
                    quat_to_matrix(q, rot);
                    T* dst = (l == 0) ? ROT : DUMP;
#pragma unroll
                    for (int k = 0; k < 9; ++k) dst[k] = rot[k];
                }
                T rs;
                {
                    T arow[D];
                    UKFB_MARK("p_chol_row");
                    load_row<T, D>(PKS, l, arow);
                    UKFB_MARK("p_chol_fact");
                    UKFB_PRIO(1);
                    rs = chol16<T, D, LS>(arow, Lc, l, ok);
                    UKFB_PRIO(0);
                    wsync();
                }
                UKFB_MARK("p_sigma");
                T col[D];
                load_column<T, D, LS>(Lc, l, rs, col);
                {   // affine rows of this lane's column, scaled by the model's diagonal factor, for the cross block;
                    // lanes without a column hold zeros: they fill LAF's zero row, or go to the sink where that row is
                    // not stored (Layout16::LAF_ROW_D)
                    T* lrow = LY::LAF_ROW_D ? (LAF + ((l < D) ? l : D) * ST) : ((l < D) ? (LAF + l * ST) : DUMP);
#pragma unroll
                    for (int c = NL; c < D; ++c) lrow[c - NL] = (M::MODEL == 0) ? col[c] : col[c] * MT<M>::aff_scale(c, pin);
                    if constexpr (LY::LAF_PAD > 0) {   // see Layout16::LAF_PAD
                        T* padp = (l == D) ? (LAF + D * ST) : DUMP;
#pragma unroll
                        for (int k = 0; k < LY::LAF_PAD; ++k) padp[k] = T(0);
                    }
                }
                sigma_pair<T, M>(mu_r, col, xp, xm);
            }
            UKFB_MARK("p_process");
            const bool pc = do_p && ok;            // this filter's predict will be committed
            sfence();
            process_fast((M*)nullptr, xp, pin);    // lanes >= D carry the centre point (their column is zero)
            sfence();   // the two process models run one after the other (interleaved by the scheduler: same registers, no gain)
            process_fast((M*)nullptr, xm, pin);
            sfence();
            UKFB_MARK("p_mean1");
            // Propagated centre point (lane D): every lane starts the mean from it.  The affine components
            // (tangent >= NL) of the sigma points are centre +- scale * L[c][l] exactly, so their mean IS the centre
            // (ukfom's iteration finds a correction at rounding level) and their deltas are the signed factor rows.
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (s < NL + 1) ref[s] = row_bcast<D>(xp[s]);   // stored 0..NL: the nonlinear components incl. the quaternion
            // (lanes >= D hold the propagated CENTRE as both of their points -- their column is zero, the process model ran on the same
            // bits twice.  Against the centre itself, the reference of the FIRST iteration, their deltas are exactly zero in the
            // Euclidean components and log(conj(q) q), zero to ~1e-17, in the rotation: that sum needs no per-lane weights.  The later
            // iterations do -- their reference has moved, and the centre counts once, not 2 (16 - D) times.)
            const bool has_p = has_pair || has_ctr, has_m = has_pair;
            const T wp = has_p ? T(1) : T(0), wm = has_m ? T(1) : T(0);

            // ---- ukfom meanSigmaPoints, first iteration over the nonlinear components (reference = centre)
            sfence();
            T n2 = T(0);
            {
                T loc[NL];
                {
                    const T qr[4] = {ref[Q], ref[Q + 1], ref[Q + 2], ref[Q + 3]};
                    const T qp[4] = {xp[Q], xp[Q + 1], xp[Q + 2], xp[Q + 3]};
                    const T qm[4] = {xm[Q], xm[Q + 1], xm[Q + 2], xm[Q + 3]};
                    T rp[3], rm[3];
                    rot_minus_n2(qp, qm, qr, qn2, rp, rm);
#pragma unroll
                    for (int k = 0; k < 3; ++k) loc[RT + k] = rp[k] + rm[k];
                }
#pragma unroll
                for (int t = 0; t < NL; ++t)
                    if (t < RT || t >= RT + 3) loc[t] = (xm[st_of(t)] - ref[st_of(t)]) + (xp[st_of(t)] - ref[st_of(t)]);
                T md[NL];
                if constexpr (sizeof(T) == 8) {
                    // fp64 has no DPP butterfly (12 VALU per value); transpose through the (free) factor region:
                    // lane c sums component c over the 16 lanes and publishes the mean
                    constexpr int TBS = 18;   // row stride: b128 rows of lanes 0..NL-1 fall on distinct banks
                    static_assert(NL * TBS + NL <= D * LS, "transposition buffer must fit the factor region");
                    T* TB = Lc;
#pragma unroll
                    for (int c = 0; c < NL; ++c) TB[c * TBS + l] = loc[c];
                    wsync();
                    const int cl = (l < NL) ? l : (NL - 1);
                    T part[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) part[j] = TB[cl * TBS + j];
#pragma unroll
                    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                        for (int j = 0; j < w; ++j) part[j] += part[j + w];
                    TB[NL * TBS + cl] = part[0] * (T(1) / T(N));
                    wsync();
#pragma unroll
                    for (int c = 0; c < NL; ++c) {
                        md[c] = TB[NL * TBS + c];
                        n2 += md[c] * md[c];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NL; ++c) {
                        md[c] = row_allreduce(loc[c]) * (T(1) / T(N));
                        n2 += md[c] * md[c];
                    }
                }
#if defined(UKFB_COUNTS)
                {   // size of the first mean delta's rotation part, largest of the wavefront's committing filters: bin b = |.| in (1e-(8-b), 1e-(7-b)], bin 0 = up to 1e-7
                    const T a2d = md[RT] * md[RT] + md[RT + 1] * md[RT + 1] + md[RT + 2] * md[RT + 2];
                    const double thr[7] = {1e-14, 1e-12, 1e-10, 1e-8, 1e-6, 1e-4, 1e-2};
                    int bin = 0;
#pragma unroll
                    for (int k = 0; k < 7; ++k) bin += wave_any(pc && double(a2d) > thr[k]) ? 1 : 0;
                    UKFB_COUNT_VAL(3, bin);
                }
#endif
                // reference [+] mean delta
                {
                    const T qr[4] = {ref[Q], ref[Q + 1], ref[Q + 2], ref[Q + 3]};
                    const T v[3] = {md[RT], md[RT + 1], md[RT + 2]};
                    T e[4], r[4];
                    so3_exp_fast(v, T(1), e);
                    quat_mul(qr, e, r);
#pragma unroll
                    for (int t = 0; t < NL; ++t)
                        if (t < RT || t >= RT + 3) ref[st_of(t)] += md[t];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ref[Q + k] = r[k];
                }
            }
            UKFB_MARK("p_delta_e");
            // Euclidean part of the mean is final: write those delta columns now and drop the registers.
            wsync();  // every lane is done with the transposition buffer (it aliases the table)
            // rows 0..D (lane l <= D): U_l, the centre lane's row scaled by sqrt(1/2) (its U is delta_0);
            // rows D+1..N: W_l, where the centre lane's W = 0 is the table's zero row
            T* const rowu = has_p ? (TAB + l * ST) : DUMP;
            T* const roww = has_p ? (TAB + (D + 1 + l) * ST) : DUMP;
            const T fu = has_ctr ? T(0.70710678118654752440) : T(1);
            {
#pragma unroll
                for (int t = 0; t < NL; ++t)
                    if (t < RT || t >= RT + 3) {
                        rowu[t] = fu * (T(0.5) * (xp[st_of(t)] + xm[st_of(t)]) - ref[st_of(t)]);
                        roww[t] = T(0.5) * (xp[st_of(t)] - xm[st_of(t)]);
                    }
            }
            {   // mean staging: the nonlinear Euclidean part from lane 0 (the quaternion follows after the loop), the
                // affine part straight from the centre lane, which holds it
                T* dst = (pc && l == 0) ? MUS : DUMP;
                T* dstc = (pc && has_ctr) ? MUS : DUMP;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (s >= NL + 1) dstc[s] = xp[s];
                    else if (s < Q || s >= Q + 4) dst[s] = ref[s];
                }
            }
            sfence();
            UKFB_MARK("p_mean_it");
            // ---- remaining iterations: SO(3) component only
            T qr[4] = {ref[Q], ref[Q + 1], ref[Q + 2], ref[Q + 3]};
            const T qp[4] = {xp[Q], xp[Q + 1], xp[Q + 2], xp[Q + 3]};
            const T qm[4] = {xm[Q], xm[Q + 1], xm[Q + 2], xm[Q + 3]};
            bool conv = true;
            // the last trip's rotation deltas and the move it applied to the reference: the final deltas follow from
            // them by so3_rebase_small instead of a third round of logarithms (see p_delta_r)
            T rpl[3] = {T(0), T(0), T(0)}, rml[3] = {T(0), T(0), T(0)}, al[3] = {T(0), T(0), T(0)};
            bool have_last = false;   // wave-uniform
            {
                bool active = pc && n2 > T(a.mean_tol) * T(a.mean_tol);   // (rows that commit nothing never keep the wavefront iterating)
                int it = 0;
                if (active && ++it >= a.mean_max_it) { active = false; conv = false; }
#if defined(UKFB_COUNTS)
                int wave_trips = 0;
#endif
                while (wave_any(active)) {
#if defined(UKFB_COUNTS)
                    ++wave_trips;
#endif
                    T rp[3], rm[3], mr[3];
                    rot_minus_n2(qp, qm, qr, qn2, rp, rm);
                    T m2 = T(0);
#pragma unroll
                    for (int k = 0; k < 3; ++k) mr[k] = fma(wm, rm[k], wp * rp[k]);
                    row_allreduce_n<T, 3>(mr);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        mr[k] *= (T(1) / T(N));
                        m2 += mr[k] * mr[k];
                    }
                    T e[4], nq[4];
                    so3_exp_fast(mr, T(1), e);
                    quat_mul(qr, e, nq);
                    have_last = true;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        rpl[k] = rp[k];
                        rml[k] = rm[k];
                        al[k] = active ? mr[k] : T(0);   // a filter that had converged earlier keeps its reference
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) qr[k] = active ? nq[k] : qr[k];
                    const bool more = m2 > T(a.mean_tol) * T(a.mean_tol);
                    const bool capped = more && (it + 1 >= a.mean_max_it);
                    it += (active && more) ? 1 : 0;
                    conv = conv && !(active && capped);
                    active = active && more && !capped;
                }
#if defined(UKFB_COUNTS)
                UKFB_COUNT_VAL(0, wave_trips);                      // trips of the wavefront (max over its four filters)
                UKFB_COUNT_VAL(1, it);                              // iterations of row 0's filter (incl. the first, p_mean1)
#endif
            }
            UKFB_MARK("p_delta_r");
            // lane constants of the covariance phase (CovTab, Pose kernels), requested here so that they have arrived when it starts
            uint32_t ctr[CovTab<T, M, TS>::NRD];
            if constexpr (M::MODEL == 0) {
                const uint32_t* rrow = CovTab<T, M, TS>::tabs.rd[l];
#pragma unroll
                for (int k = 0; k < CovTab<T, M, TS>::NRD; ++k) ctr[k] = rrow[k];
            }
            uint32_t otr[4] = {0u, 0u, 0u, 0u};   // OrientationState: the lane's rd row, eight 16-bit values
            if constexpr (M::MODEL != 0) {
                const uint32_t* rrow = reinterpret_cast<const uint32_t*>(OCovTab<T, M, TS>::tabs.rd[l]);
#pragma unroll
                for (int k = 0; k < 4; ++k) otr[k] = rrow[k];
            }
            {   // rotation deltas to the final mean; quaternion of the mean.  The loop's last trip took the logarithms
                // against the reference BEFORE its (sub-tolerance) move al: re-base them to first order in al, exact in
                // the delta (remainder < 4e-14 under the bounds tested here); anything else takes the logarithms again.
                T rp[3], rm[3];
                bool rebase = have_last;
                const T a2 = al[0] * al[0] + al[1] * al[1] + al[2] * al[2];
                if (rebase) {
                    const T tp = rpl[0] * rpl[0] + rpl[1] * rpl[1] + rpl[2] * rpl[2];
                    const T tm = rml[0] * rml[0] + rml[1] * rml[1] + rml[2] * rml[2];
                    // (a row whose prediction is not committed -- uninitialised, gated out, failed factorisation: NaN deltas --
                    // has no say in what its wave-mates do)
                    rebase = wave_all(!pc || (tp <= T(2.25) && tm <= T(2.25) && a2 <= T(1e-12)));
                }
                UKFB_COUNT_VAL(2, rebase ? 1 : 0);                  // final deltas: re-based (1) or a third round of logarithms (0)
                if (rebase) {
                    so3_rebase_small(rpl, al, a2, rp);
                    so3_rebase_small(rml, al, a2, rm);
                } else {
                    rot_minus_n2(qp, qm, qr, qn2, rp, rm);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    rowu[RT + k] = (fu * T(0.5)) * (rp[k] + rm[k]);
                    roww[RT + k] = T(0.5) * (rp[k] - rm[k]);
                }
                T* dst = (pc && l == 0) ? MUS : DUMP;
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[Q + k] = qr[k];
            }
            wsync();
            UKFB_MARK("p_cov");
            // ---- covariance.  With d = sigma-point deltas (NL columns), U / W as above, A = scaled affine factor rows:
            //   nonlinear block   Sigma'[a][b] = 1/2 sum_i d_i[a] d_i[b] = sum over the table rows t of t[a] t[b]   (a, b < NL)
            //   cross block       Sigma'[c][a] = sum_l A_l[c] W_l[a]                           (c >= NL > a)
            //   affine block      Sigma'[c][e] = s_c s_e Sigma[c][e]                            (c, e >= NL; L L^T = Sigma)
            // each plus the shaped process noise.  Lane -> one TR x TC tile (MT<M>::TILE_R / TILE_C): a tile of the
            // nonlinear block is shared by two lanes (rows 0..D and D+1..N of the table, N = zero row), a tile of the
            // cross block belongs to one lane (rows 0..D of A and W, D = zero row): TRIP = D + 1 iterations for all.
            // Which tile a lane owns, where its operands and results live is a function of the lane index alone.  The Pose
            // kernels read it from CovTab (byte offsets, one row per lane) instead of decoding it in every wavefront:
            // -60 VALU instructions per wavefront, +1.5 % on the fp32 engine, which is issue-bound (same-box A/B; the
            // latency-bound fp64 engine +0.4 %).  The OrientationState kernels keep the decoded form: their 3 x 3 tiles need
            // twice the table loads, and with them config 4 measured -1.4 %.
            constexpr bool LANE_TABLES = (M::MODEL == 0);
            if constexpr (LANE_TABLES) {
                using CT = CovTab<T, M, TS>;
                static_assert(10 + TR <= D && 3 + TC <= NL, "Pose tile table: every tile entry is a valid (row, column)");
                constexpr int AEL = CT::AEL;   // affine block entries per lane
                unsigned char* const wbase = reinterpret_cast<unsigned char*>(base);
                const T* pr = reinterpret_cast<const T*>(wbase + ctr[CT::RD_PR]);
                const T* pc_ = reinterpret_cast<const T*>(wbase + ctr[CT::RD_PC]);   // cross: the W rows
                // results: the row of the lane, or the all-sink row when this filter's prediction is not committed
                uint32_t ctw[CT::NWR];
                {
                    const uint32_t* wrow = CT::tabs.wr[pc ? l : 16];
#pragma unroll
                    for (int k = 0; k < CT::WR_AFF_RD + AEL; ++k) ctw[k] = wrow[k];
                }
                const int64_t wgn = wg0_again() * a.Rn_stride;
                const TS* Rn = at(a.Rn + wgn, fc * IDX(a.Rn_stride));
                const TS* Ra = at(a.Racc + wgn, fc * IDX(a.Rn_stride));
                // element imm behind byte offset off of the (HBM) table p
                const auto at_off = [](const TS* p, uint32_t off, int imm) {
                    return T(reinterpret_cast<const TS*>(reinterpret_cast<const unsigned char*>(p) + off)[imm]);
                };
                // plain (un-rotated) noise entry (cf. plain_noise_entry16)
                const auto plain_off = [&](uint32_t off, int imm) {
                    const T rn = at_off(Rn, off, imm), vacc = at_off(Ra, off, imm);
                    return pin.use_acc ? vacc : pin.dt * rn;
                };
                const bool all_acc = wave_all(pin.use_acc);
                // shaped process noise of this lane's tile: requested before the accumulation loop, consumed after it
                T acc[TR][TC], nz[TR][TC], anz[AEL];
#pragma unroll
                for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < TC; ++j2) acc[i2][j2] = nz[i2][j2] = T(0);
#pragma unroll
                for (int t = 0; t < AEL; ++t) anz[t] = T(0);
                if (all_acc) {   // wave-uniform: plain table reads (acceleration branch: raw noise), all in flight together
#pragma unroll
                    for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                        for (int j2 = 0; j2 < TC; ++j2) nz[i2][j2] = at_off(Ra, ctr[CT::RD_NZ], i2 * D + j2);
#pragma unroll
                    for (int t = 0; t < AEL; ++t) anz[t] = at_off(Ra, ctr[CT::RD_ANZ + t], 0);
                } else {
                    // Rotated noise (PoseUKF.cpp:184-185): only the two 3x3 diagonal blocks of the nonlinear 6x6 block are
                    // rotated.  Its 21 entries are evaluated ONCE per filter, at most two per lane, and parked in LDS.
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bool v = l + G * t < NL * (NL + 1) / 2;
                        const int r = v ? int((tri_rows(G * t) >> (4 * l)) & 15ull) : 0, c = v ? int((tri_cols(G * t) >> (4 * l)) & 15ull) : 0;
                        NSH[v ? (l + G * t) : (LY::NSH_SINK - LY::NSH)] = process_noise_entry16<T, M>(Rn, Ra, ROT, pin, r, c);
                    }
                    const int lw = (l < MT<M>::WORK_LANES) ? l : 0;
                    const int R0 = int((MT<M>::TILE_R >> (4 * lw)) & 15ull), C0 = int((MT<M>::TILE_C >> (4 * lw)) & 15ull);
                    const bool is_cross = R0 >= NL;
#pragma unroll
                    for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                        for (int j2 = 0; j2 < TC; ++j2) {
                            const int rn_ = is_cross ? 0 : (R0 + i2), cn_ = is_cross ? 0 : (C0 + j2);   // a nonlinear tile: rows / columns < NL
                            T shaped = NSH[rn_ * (rn_ + 1) / 2 + cn_];
                            keep(shaped);
                            const T plain = plain_off(ctr[CT::RD_NZ], i2 * D + j2);
                            nz[i2][j2] = is_cross ? plain : shaped;
                        }
#pragma unroll
                    for (int t = 0; t < AEL; ++t) anz[t] = plain_off(ctr[CT::RD_ANZ + t], 0);
                }
#pragma unroll
                for (int i = 0; i < TRIP; ++i) {
                    T vr[TR], vc[TC];
#pragma unroll
                    for (int k = 0; k < TR; ++k) vr[k] = pr[i * ST + k];
#pragma unroll
                    for (int k = 0; k < TC; ++k) vc[k] = pc_[i * ST + k];
#pragma unroll
                    for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                        for (int j2 = 0; j2 < TC; ++j2) acc[i2][j2] = fma(vr[i2], vc[j2], acc[i2][j2]);
                }
                p_commit = pc;
                st |= (do_p && !ok) ? ST_ERR_CHOLESKY : 0u;
                st |= (p_commit && !conv) ? ST_WARN_MEAN_NOCONV : 0u;
                // the two halves of a nonlinear tile sit on neighbouring lanes (xor 1); a cross lane keeps its own sum
                // (weight 0: its neighbour's accumulators are finite sums of the same filter)
                T wsum;
                if constexpr (sizeof(T) == 8)
                    wsum = __builtin_bit_cast(T, (unsigned long long)ctr[CT::RD_WS] | ((unsigned long long)ctr[CT::RD_WS + 1] << 32));
                else
                    wsum = __builtin_bit_cast(T, ctr[CT::RD_WS]);
#pragma unroll
                for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < TC; ++j2) {
                        const T tot = fma(wsum, dpp_mov<0xB1>(acc[i2][j2]), acc[i2][j2]);   // quad_perm [1,0,3,2]
                        *reinterpret_cast<T*>(wbase + ctw[CT::WR_TILE + i2 * TC + j2]) = tot + nz[i2][j2];
                    }
                // affine block in place (unit scale): the old entries are still staged
#pragma unroll
                for (int t = 0; t < AEL; ++t) {
                    T old = *reinterpret_cast<const T*>(wbase + ctw[CT::WR_AFF_RD + t]);
                    keep(old);
                    *reinterpret_cast<T*>(wbase + ctw[CT::WR_AFF + t]) = old + anz[t];
                }
            } else {
                // OrientationState: 3 x 3 tiles, lane constants from OCovTab (16-bit offsets, three loads per lane)
                using OT = OCovTab<T, M, TS>;
                static_assert(M::MODEL == 1, "the decoded form of this phase is gone: every model has lane tables");
                constexpr int AEL = OT::AEL;
                unsigned char* const wbase = reinterpret_cast<unsigned char*>(base);
                const auto lo16 = [](uint32_t x) { return x & 0xFFFFu; };
                const auto hi16 = [](uint32_t x) { return x >> 16; };
                const T* pr = reinterpret_cast<const T*>(wbase + lo16(otr[0]));
                const T* pc_ = reinterpret_cast<const T*>(wbase + hi16(otr[0]));   // cross: the W rows
                const uint32_t flags = lo16(otr[1]);
                // results: the row of the lane, or the all-sink row when this filter's prediction is not committed
                uint32_t otw[OT::NWR / 2];
                {
                    const uint32_t* wrow = reinterpret_cast<const uint32_t*>(OT::tabs.wr[pc ? l : 16]);
#pragma unroll
                    for (int k = 0; k < (OT::WR_AFF_RD + AEL + 1) / 2; ++k) otw[k] = wrow[k];
                }
                const auto wr_off = [&](int k) { return (k & 1) ? hi16(otw[k >> 1]) : lo16(otw[k >> 1]); };
                const int64_t wgn = wg0_again() * a.Rn_stride;
                const TS* Rn = at(a.Rn + wgn, fc * IDX(a.Rn_stride));
                const TS* Ra = at(a.Racc + wgn, fc * IDX(a.Rn_stride));
                // element imm behind byte offset off of the (HBM) noise table
                const auto rn_off = [&](uint32_t off, int imm) {
                    return T(reinterpret_cast<const TS*>(reinterpret_cast<const unsigned char*>(Rn) + off)[imm]);
                };
                T acc[TR][TC];
#pragma unroll
                for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < TC; ++j2) acc[i2][j2] = T(0);
                // Rotated noise (OrientationUKF.cpp:84-85): only the two 3x3 diagonal blocks of the nonlinear 6x6 block are rotated.
                // Its 21 entries are evaluated ONCE per filter, at most two per lane, and parked in LDS; the tiles pick them up after
                // the accumulation loop.  Isotropic blocks (noise_plain, the default and every BASELINE configuration) need none of it.
                if (!noise_plain) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bool v = l + G * t < NL * (NL + 1) / 2;
                        const int r = v ? int((tri_rows(G * t) >> (4 * l)) & 15ull) : 0, c = v ? int((tri_cols(G * t) >> (4 * l)) & 15ull) : 0;
                        NSH[v ? (l + G * t) : (LY::NSH_SINK - LY::NSH)] = process_noise_entry16<T, M>(Rn, Ra, ROT, pin, r, c);
                    }
                }
#pragma unroll
                for (int i = 0; i < TRIP; ++i) {
                    T vr[TR], vc[TC];
#pragma unroll
                    for (int k = 0; k < TR; ++k) vr[k] = pr[i * ST + k];
#pragma unroll
                    for (int k = 0; k < TC; ++k) vc[k] = pc_[i * ST + k];
#pragma unroll
                    for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                        for (int j2 = 0; j2 < TC; ++j2) acc[i2][j2] = fma(vr[i2], vc[j2], acc[i2][j2]);
                }
                p_commit = pc;
                st |= (do_p && !ok) ? ST_ERR_CHOLESKY : 0u;
                st |= (p_commit && !conv) ? ST_WARN_MEAN_NOCONV : 0u;
                {
                    // The noise is picked up late: ALL global loads of this lane's entries are issued here, together, before anything
                    // waits for one of them (issued one by one between the stores below they cost a full L2 round trip each: eleven
                    // in a row were 20 % of the wavefront's life).  A tile that hangs over the matrix reads past its table (padded).
                    const T dt2 = pin.dt * pin.dt;
                    T pl[TR][TC], apl[AEL];
#pragma unroll
                    for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                        for (int j2 = 0; j2 < TC; ++j2) pl[i2][j2] = dt2 * rn_off(hi16(otr[1]), i2 * D + j2);
                    apl[0] = dt2 * rn_off(lo16(otr[2]), 0);
                    apl[1] = dt2 * rn_off(hi16(otr[2]), 0);
                    sfence();
                    // the two halves of a nonlinear tile sit on neighbouring lanes (xor 1); a cross lane keeps its own sum
                    // (weight 0: its neighbour's accumulators are finite sums of the same filter)
                    const bool nonlin = (flags & 1u) != 0u;
                    const T wsum = nonlin ? T(1) : T(0);
                    // (isotropic noise blocks: every entry is the plain table value that is already in flight, the shaped-noise table
                    // is neither filled nor read -- a second copy of the loop under a wave-uniform branch, because a branch per entry
                    // makes the register allocator spill)
                    auto store_tiles = [&](auto plain_c) {
                        constexpr bool PLAIN_NOISE = decltype(plain_c)::value;
                        int R0 = 0, C0 = 0;
                        if constexpr (!PLAIN_NOISE) {   // where the shaped entries sit: the tile origin, decoded on this path only
                            const int lw = (l < MT<M>::WORK_LANES) ? l : 0;
                            R0 = int((MT<M>::TILE_R >> (4 * lw)) & 15ull);
                            C0 = int((MT<M>::TILE_C >> (4 * lw)) & 15ull);
                        }
#pragma unroll
                        for (int i2 = 0; i2 < TR; ++i2)
#pragma unroll
                            for (int j2 = 0; j2 < TC; ++j2) {
                                T nv = pl[i2][j2];
                                if constexpr (!PLAIN_NOISE) {
                                    const int rn_ = nonlin ? (R0 + i2) : 0, cn_ = nonlin ? (C0 + j2) : 0;   // a nonlinear tile: rows / columns < NL
                                    T shaped = NSH[rn_ * (rn_ + 1) / 2 + cn_];
                                    keep(shaped);
                                    nv = nonlin ? shaped : nv;
                                }
                                const T tot = fma(wsum, dpp_mov<0xB1>(acc[i2][j2]), acc[i2][j2]);   // quad_perm [1,0,3,2]
                                // (entries the lane does not own: the noise table's spare slot -- the table may still be read)
                                *reinterpret_cast<T*>(wbase + wr_off(OT::WR_TILE + i2 * TC + j2)) = tot + nv;
                            }
                    };
                    if (noise_plain) store_tiles(std::true_type{});
                    else store_tiles(std::false_type{});
                    // affine block in place: the old entries are still staged; scale classes of row and column from the flags
                    const T sg = MT<M>::aff_scale(NL, pin), sa = MT<M>::aff_scale(NL + 3, pin);
                    const auto scale_of = [&](int shift) {
                        const uint32_t cls = (flags >> shift) & 3u;
                        return (cls == 0u) ? sg : ((cls == 1u) ? sa : T(1));
                    };
#pragma unroll
                    for (int t = 0; t < AEL; ++t) {
                        T old = *reinterpret_cast<const T*>(wbase + wr_off(OT::WR_AFF_RD + t));
                        keep(old);
                        const T ss = scale_of(2 + 4 * t) * scale_of(4 + 4 * t);
                        *reinterpret_cast<T*>(wbase + wr_off(OT::WR_AFF + t)) = fma(ss, old, apl[t]);
                    }
                }
            }
            UKFB_MARK("p_end");
            // a gated / failed predict leaves the staged state as it was: every store to MUS / PKS above is
            // predicated on this filter's commit
            wsync();
        }
    }

    // =========================================================================== update
    if constexpr (DO_UPDATE) {
        if (wave_any(do_u)) {
            UKFB_MARK("u_stats");
            T zin[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) zin[k] = ZQ[k];
            if (M::CHECK_MEAS_FINITE) {
                bool fin = true;
#pragma unroll
                for (int k = 0; k < 12; ++k) fin = fin && m_finite(ZQ[k]);
                st |= (do_u && !fin) ? ST_ERR_NONFINITE_MEAS : 0u;
                do_u = do_u && fin;
            }
            const int midc = M::meas_valid(mid) ? mid : (M::MODEL == 0 ? 0 : 9);
            // (plain launches: a full 3-vector Euclidean selection for PoseWithVelocity, the body velocity for OrientationState)
            const int m = PLAINL ? 3 : M::meas_dim(midc);
            const bool so3 = PLAINL ? false : M::meas_is_so3(midc);
            const bool need_q = so3 || (M::MODEL == 1);
            // Measurement statistics: S (innovation covariance), cx (row l of Sigma_xz), innovation.
            bool ok1 = true, zconv = true;
            T Sm[9], cx[3], innov[3];
            if constexpr (MULTI && sizeof(T) == 8 && M::MODEL != 0) {
                // Inside the cycle loop a variable that is only assigned under a wave-uniform branch carries its (unused) value of
                // the previous cycle around the loop: a register pair each, for the whole cycle.  Fifteen fp64 values here --
                // with them the OrientationState fp64 multi-cycle kernel needed 192 VGPRs (two wavefronts per SIMD, slower than
                // single launches), without them 155 (three; +15 %, same-box A/B).  The other multi-cycle kernels have the
                // registers to spare and skip the fifteen moves per cycle (-0.5 ... -1.3 % with them).
#pragma unroll
                for (int k = 0; k < 9; ++k) Sm[k] = T(0);
#pragma unroll
                for (int k = 0; k < 3; ++k) cx[k] = innov[k] = T(0);
            }
            const int la = has_pair ? l : (D - 1);
            if (MT<M>::HAS_EUCLID_MEAS) {
                // Sub-state selections (PoseUKF.cpp:7-26,35-69) are LINEAR in the tangent, and the unscented
                // transform of a linear map is exact: with Sigma = L L^T the sigma-point sums collapse to
                //   zbar = mu[sel],  S = Sigma[sel][sel] + Q,  Sigma_xz = Sigma[:, sel]
                // (the same identity applyDelta uses), so neither the factorisation nor the spread is needed.
                // One model id for the whole launch (a kernel argument, i.e. scalar) that selects a full 3-vector --
                // position, velocity or angular velocity: three consecutive tangent components from a scalar base,
                // so every index below is scalar arithmetic and nothing needs a select.
                const int mu_id = meas_uniform_c;
                const bool uni3 = PLAINL || ((a_meas == nullptr) && (mu_id == 0 || mu_id == 4 || mu_id == 8));
                if (uni3) {
                    const int tb = (mu_id == 0) ? 0 : ((mu_id == 4) ? 6 : 9);   // first tangent index
                    const int sb = tb + ((tb >= Q) ? 1 : 0);                       // first stored index
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        innov[k] = zin[k] - MUS[sb + k];
                        const int tk = tb + k;
                        const int hi = la > tk ? la : tk, lo = la > tk ? tk : la;
                        cx[k] = base[LY::cv(hi, lo)];
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c) {
                            const T sp = base[LY::cv(tb + r, tb + c)];
                            Sm[r * 3 + c] = sp + ZQ[3 + r * 3 + c];
                            if (c < r) Sm[c * 3 + r] = sp + ZQ[3 + c * 3 + r];
                        }
                } else {
                const unsigned long long sel[3] = {MT<M>::SEL0, MT<M>::SEL1, MT<M>::SEL2};
                int ti[3];
                bool used[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int sk = int((sel[k] >> (4 * midc)) & 15ull);
                    used[k] = sk != 15;
                    const int si = used[k] ? sk : 0;
                    ti[k] = (si < Q) ? si : (si - 1);
                    T m0 = MUS[si];
                    keep(m0);
                    innov[k] = used[k] ? (zin[k] - m0) : T(0);
                    const int hi = la > ti[k] ? la : ti[k], lo = la > ti[k] ? ti[k] : la;
                    T sx = base[LY::cv(hi, lo)];
                    keep(sx);
                    cx[k] = used[k] ? sx : T(0);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int hi = ti[r] > ti[c] ? ti[r] : ti[c], lo = ti[r] > ti[c] ? ti[c] : ti[r];
                        const T pad = (r == c) ? T(1) : T(0);
                        T sp = base[LY::cv(hi, lo)], sq = ZQ[3 + r * 3 + c];
                        keep(sp);
                        keep(sq);
                        Sm[r * 3 + c] = (used[r] && used[c]) ? (sp + sq) : pad;
                    }
                }
            }
            if (wave_any(need_q)) {
                // Orientation-dependent models (PoseUKF.cpp:28-33, OrientationUKF.cpp:34-39): full sigma-point
                // path of ukfom::update.  Wave-uniform branch; results are selected per filter below.
                T zval[4] = {(0 < m) ? zin[0] : T(0), (1 < m) ? zin[1] : T(0), (2 < m) ? zin[2] : T(0), T(0)};
                if (wave_any(so3)) {
                    T qe[4];
                    so3_exp_fast(zin, T(1), qe);  // RotationType(SO3::exp(mu)), PoseUKF.cpp:135
#pragma unroll
                    for (int k = 0; k < 4; ++k) zval[k] = so3 ? qe[k] : zval[k];
                }
                bool okg;
                T rs;
                {
                    T arow[D];
                    load_row<T, D>(PKS, l, arow);
                    // Only the first ZCOLS columns of the factor move the measurement.  An indefinite Sigma whose
                    // first ZCOLS pivots are positive is caught by the complete factorisation of Sigma' below
                    // (Sigma' <= Sigma), with the same status bit.
                    UKFB_PRIO(1);
                    rs = chol16<T, D, LS, MT<M>::ZCOLS>(arow, Lc, l, okg);
                    UKFB_PRIO(0);
                    wsync();
                }
                T zp[4], zm[4], z0[4];
                {
                    const bool zcol = l < MT<M>::ZCOLS;
                    const T w = zcol ? rs : T(0);
                    const T* colp = Lc + (zcol ? l : (MT<M>::ZCOLS - 1)) * LS;
                    const T q0[4] = {MUS[Q], MUS[Q + 1], MUS[Q + 2], MUS[Q + 3]};
                    const T cr[3] = {colp[RT] * w, colp[RT + 1] * w, colp[RT + 2] * w};
                    T e[4], qp[4], qm[4];
                    so3_exp_fast(cr, T(1), e);
                    quat_mul_pm(q0, e, qp, qm);
                    MT<M>::gen_measure(qp, qm, q0, MUS, colp, w, zp, zm, z0);
                }
                sfence();
                // ---- mean of Z.  Euclidean: one pass is exact.  SO(3): iterate on the manifold.
                T zref[4] = {z0[0], z0[1], z0[2], z0[3]};
                bool zc = true;
                if (wave_any(so3 && do_u)) {
                    bool active = so3;
                    int it = 0;
                    while (wave_any(active)) {
                        T rp[3], rm[3], r0v[3], mr[3];
                        rot_minus(zp, zref, rp);
                        rot_minus(zm, zref, rm);
                        rot_minus(z0, zref, r0v);
                        T m2 = T(0);
#pragma unroll
                        for (int k = 0; k < 3; ++k) mr[k] = has_pair ? (rp[k] + rm[k]) : T(0);
                        row_allreduce_n<T, 3>(mr);
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            mr[k] = (mr[k] + r0v[k]) * (T(1) / T(N));
                            m2 += mr[k] * mr[k];
                        }
                        T e[4], nq[4];
                        so3_exp_fast(mr, T(1), e);
                        quat_mul(zref, e, nq);
#pragma unroll
                        for (int k = 0; k < 4; ++k) zref[k] = active ? nq[k] : zref[k];
                        const bool more = m2 > T(a.mean_tol) * T(a.mean_tol);
                        const bool capped = more && (it + 1 >= a.mean_max_it);
                        it += (active && more) ? 1 : 0;
                        zc = zc && !(active && capped);
                        active = active && more && !capped;
                    }
                }
                {
                    T zr[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) zr[k] = has_pair ? ((zp[k] - z0[k]) + (zm[k] - z0[k])) : T(0);
                    row_allreduce_n<T, 3>(zr);
#pragma unroll
                    for (int k = 0; k < 3; ++k) zr[k] = z0[k] + zr[k] * (T(1) / T(N));
#pragma unroll
                    for (int k = 0; k < 3; ++k) zref[k] = so3 ? zref[k] : zr[k];
                }
                sfence();
                // ---- deltas to the measurement mean, S, innovation
                T dzp[3], dzm[3], dz0[3], inn[3];
                {
                    T a3[3] = {T(0), T(0), T(0)}, b3[3] = {T(0), T(0), T(0)}, c3[3] = {T(0), T(0), T(0)},
                      d3[3] = {T(0), T(0), T(0)};
                    if (wave_any(so3)) {
                        rot_minus(zp, zref, a3);
                        rot_minus(zm, zref, b3);
                        rot_minus(z0, zref, c3);
                        rot_minus(zval, zref, d3);
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        dzp[k] = so3 ? a3[k] : (zp[k] - zref[k]);
                        dzm[k] = so3 ? b3[k] : (zm[k] - zref[k]);
                        dz0[k] = so3 ? c3[k] : (z0[k] - zref[k]);
                        inn[k] = so3 ? d3[k] : (zval[k] - zref[k]);
                    }
                }
                T Sg[9];
                {
                    // measurement covariance from its LDS staging; unused trailing dimensions are decoupled
                    // (Q = I there, z = h = 0), which leaves the leading m x m inverse bit-equal
                    T Qm[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const T pad = (r == c) ? T(1) : T(0);
                            T qv = ZQ[3 + r * 3 + c];
                            keep(qv);
                            Qm[r * 3 + c] = (r >= m || c >= m) ? pad : qv;
                        }
                    T u6[6];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c)
                            u6[r * (r + 1) / 2 + c] = has_pair ? fma(dzp[r], dzp[c], dzm[r] * dzm[c]) : T(0);
                    row_allreduce_n<T, 6>(u6);
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c)
                            u6[r * (r + 1) / 2 + c] = T(0.5) * (u6[r * (r + 1) / 2 + c] + dz0[r] * dz0[c]);
                    Sg[0] = u6[0] + Qm[0];
                    Sg[3] = u6[1] + Qm[3]; Sg[1] = u6[1] + Qm[1];
                    Sg[4] = u6[2] + Qm[4];
                    Sg[6] = u6[3] + Qm[6]; Sg[2] = u6[3] + Qm[2];
                    Sg[7] = u6[4] + Qm[7]; Sg[5] = u6[4] + Qm[5];
                    Sg[8] = u6[5] + Qm[8];
                }
                sfence();
                // ---- cross covariance: Cxz[a][c] = sum_l L[a][l] * 0.5 (dz+_l - dz-_l)[c]
                // (columns >= ZCOLS do not move the measurement: dz+ == dz- bit for bit, their W row is zero)
                T cg[3] = {T(0), T(0), T(0)};
                {
                    T w[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        w[k] = T(0.5) * rs * (dzp[k] - dzm[k]);
                        dpp_hazard_fence(w[k]);
                    }
                    static_for<0, MT<M>::ZCOLS>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const T v = Lc[j * LS + la];   // zero for j > la
                        fmac_bcast<j>(cg[0], w[0], v);
                        fmac_bcast<j>(cg[1], w[1], v);
                        fmac_bcast<j>(cg[2], w[2], v);
                    });
                }
                ok1 = need_q ? okg : ok1;
                zconv = need_q ? zc : zconv;
#pragma unroll
                for (int k = 0; k < 9; ++k) Sm[k] = (need_q || !MT<M>::HAS_EUCLID_MEAS) ? Sg[k] : Sm[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    cx[k] = (need_q || !MT<M>::HAS_EUCLID_MEAS) ? cg[k] : cx[k];
                    innov[k] = (need_q || !MT<M>::HAS_EUCLID_MEAS) ? inn[k] : innov[k];
                }
            }
            sfence();
            UKFB_MARK("u_gain");
            // lane constants of the assembly phase, requested here so that they have arrived when it starts
            uint32_t asmo[4];
            {
                const uint32_t* arow = AsmTab<T, M>::tabs.off[l];
#pragma unroll
                for (int k = 0; k < 4; ++k) asmo[k] = arow[k];
            }
            T Kr[3], KSr[3];
            bool accept;
            {
                T Si[9];
                inverse3(Sm, Si);
                T maha = T(0);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) maha += innov[r] * Si[r * 3 + c] * innov[c];
                accept = (gate_chi2_c < T(0)) || (maha <= gate_chi2_c);
#pragma unroll
                for (int c = 0; c < 3; ++c) Kr[c] = cx[0] * Si[c] + cx[1] * Si[3 + c] + cx[2] * Si[6 + c];
#pragma unroll
                for (int c = 0; c < 3; ++c) KSr[c] = Kr[0] * Sm[c] + Kr[1] * Sm[3 + c] + Kr[2] * Sm[6 + c];
            }
            const T del = Kr[0] * innov[0] + Kr[1] * innov[1] + Kr[2] * innov[2];
            sfence();
            UKFB_MARK("u_downdate_chol");
            // ---- Sigma' = Sigma - (K S) K^T, row l on lane l; delta on every lane
            T srow2[D], drot[3];
            bool ok2;
            T rs2;
            {
                T arow2[D];
                load_row<T, D>(PKS, l, arow2);
                const T nks[3] = {-KSr[0], -KSr[1], -KSr[2]};
                dpp_hazard_fence(Kr[0]);
                dpp_hazard_fence(Kr[1]);
                dpp_hazard_fence(Kr[2]);
                static_for<0, D>([&](auto bc) {   // lane b holds row b of the gain and delta[b]
                    constexpr int b = decltype(bc)::value;
                    fmac_bcast<b>(arow2[b], Kr[0], nks[0]);
                    fmac_bcast<b>(arow2[b], Kr[1], nks[1]);
                    fmac_bcast<b>(arow2[b], Kr[2], nks[2]);
                    srow2[b] = arow2[b];
                    if constexpr (b >= RT && b < RT + 3) drot[b - RT] = row_bcast<b>(del);   // the rotation part of delta, on every lane
                });
                // applyDelta reads the first RT + 3 columns of this factor; the remaining steps exist to find out whether Sigma' is
                // positive definite at all.  In a fused cycle of the plain kind that is already established when (ukfb_config::
                // full_update_check) the filter's prediction was committed in THIS launch -- its input covariance factorised, the
                // predicted one a Gram matrix plus a noise the host found positive semidefinite -- and the sample's measurement
                // covariance is positive definite: Sigma' = Sigma - K S K^T is then the Schur complement of a positive definite joint
                // covariance.  Wave-uniform: one filter that does not qualify sends its wavefront through the complete factorisation
                // (every update-only launch and every other kernel always).  The published columns are the same either way.
                bool short_fact = false;
                if constexpr (DO_PREDICT && PLAIN >= 1) {   // (streams-only and plain fused cycles; m = the filter's measurement dimension)
                    if (a.upd_short_ok != 0) {
                        const T q00 = ZQ[3], q10 = ZQ[6], q11 = ZQ[7], q20 = ZQ[9], q21 = ZQ[10], q22 = ZQ[11];
                        const T m2 = fma(q00, q11, -(q10 * q10));
                        const T det = fma(q22, m2, fma(q20, fma(q10, q21, -(q11 * q20)), -(q21 * fma(q00, q21, -(q10 * q20)))));
                        // leading minors of the m x m covariance the update uses (the rest of the 3 x 3 is padding)
                        const bool q_pd = (q00 > T(0)) && (m < 2 || m2 > T(0)) && (m < 3 || det > T(0));
                        short_fact = wave_all(!do_u || (p_commit && q_pd));
                    }
                }
                UKFB_PRIO(1);
                if (short_fact) rs2 = chol16<T, D, LS, RT + 3, RT + 3>(arow2, Lc, l, ok2);
                else rs2 = chol16<T, D, LS, D, RT + 3>(arow2, Lc, l, ok2);
                UKFB_PRIO(0);
                wsync();
            }
            sfence();
            UKFB_MARK("u_apply");
            // ---- applyDelta: mu' = mu [+] delta; only the SO(3) rows/columns are re-sampled.  Only the first
            // NC = RT + 3 columns of the (lower-triangular) factor have a rotation part, so there are 2 NC sigma
            // points to push through exp / log: lane j < NC takes +column j, lane NC + j takes -column j.
            constexpr int NC = RT + 3;
            static_assert(2 * NC <= 16, "signed rotation columns must fit one DPP row");
            T e0[4], rsg[3];
            {
                const bool plus = l < NC, minus = (l >= NC) && (l < 2 * NC);
                const int cidx = plus ? l : (minus ? (l - NC) : 0);
                const T rs_m = dpp_mov<0x110 + NC>(rs2);                 // row_shr: lane l reads lane l - NC
                const T wc = plus ? rs2 : (minus ? -rs_m : T(0));
                T c3[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) c3[k] = Lc[cidx * LS + RT + k] * wc;
                const T v0[3] = {drot[0], drot[1], drot[2]};
                const T vs[3] = {v0[0] + c3[0], v0[1] + c3[1], v0[2] + c3[2]};
                T es[4];
                so3_exp_fast(vs, T(1), es);
                // exp(delta_r) itself: lanes 2 NC .. 15 hold no signed column (wc = 0), their `es` IS exp(v0 + 0) -- one broadcast per
                // component instead of a second exponential on every lane
                static_assert(2 * NC < 16, "a lane without a signed column exists");
#pragma unroll
                for (int k = 0; k < 4; ++k) e0[k] = row_bcast<15>(es[k]);
                rot_minus_n(es, e0, T(1), rsg);   // log(conj(q e0) (q e+-)) = log(conj(e0) e+-): unit quaternions
            }
            sfence();
            UKFB_MARK("u_rr_cross");
            // rotation-rotation block: 0.5 sum over the 2 NC signed points of r r^T (lanes without a point hold
            // log(conj(e0) e0) = 0 up to rounding)
            T rr[6];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) rr[r * (r + 1) / 2 + c] = rsg[r] * rsg[c];
            row_allreduce_n<T, 6, (2 * NC <= 12) ? 3 : 4>(rr);   // lanes 12..15: r = log(conj(e0) e0), zero up to ~1e-17
#pragma unroll
            for (int k = 0; k < 6; ++k) rr[k] *= T(0.5);
            // cross terms of row l with the three rotation columns: sum_j L'[l][j] W_j, W_j = (r+_j - r-_j) / 2 on lane j
            T cr[3] = {T(0), T(0), T(0)};
            {
                T w[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T rmk = dpp_mov<0x100 + NC>(rsg[k]);           // row_shl: lane j reads lane j + NC
                    w[k] = T(0.5) * rs2 * (rsg[k] - rmk);
                    dpp_hazard_fence(w[k]);
                }
                static_for<0, NC>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const T v = Lc[j * LS + la];   // zero for j > la
                    fmac_bcast<j>(cr[0], w[0], v);
                    fmac_bcast<j>(cr[1], w[1], v);
                    fmac_bcast<j>(cr[2], w[2], v);
                });
            }
            st |= (do_u && (!ok1 || (accept && !ok2))) ? ST_ERR_CHOLESKY : 0u;
            st |= (do_u && ok1 && !accept) ? ST_REJECTED_GATE : 0u;
            st |= (do_u && ok1 && !zconv) ? ST_WARN_MEAN_NOCONV : 0u;
            u_commit = do_u && ok1 && ok2 && accept;

            sfence();
            UKFB_MARK("u_assemble");
            // ---- resampled covariance into the staging area, in three ordered passes of plain stores (one
            // wavefront: LDS stores retire in program order, later passes overwrite earlier ones):
            //   1. row l of the downdated Sigma' (the Euclidean block IS the resampled block, see the header note)
            //   2. the cross terms of row l with the three rotation columns, stored at the symmetric position
            //      (max, min) so no exchange between lanes is needed
            //   3. the rotation-rotation block, identical on every lane after the all-reduce (lane 0 stores it)
            {
                const bool wl = u_commit && has_pair;
                // Pass 1 stores the WHOLE row (b = D-1 .. 0) at rowbase + b with no per-entry predicate: the entries
                // beyond the diagonal spill into the slots of later rows, and every such slot (r, c) is rewritten
                // afterwards by its owner, because a spill from row l < r lands there at step b' = c + (rowbase(r) -
                // rowbase(l)) > c, i.e. earlier in this descending loop.  The last row has nothing beyond its diagonal.
                // (The stores of one lane never alias each other, so the compiler would be free to reorder or pair
                // them; the order that matters is between LANES, hence the compiler fence after every store.)
                unsigned char* const wbase = reinterpret_cast<unsigned char*>(base);
                constexpr uint32_t SINK = uint32_t(LY::DUM * sizeof(T));
                T* rowdst = reinterpret_cast<T*>(wbase + (wl ? asmo[0] : SINK));
#pragma unroll
                for (int b = D - 1; b >= 0; --b) {
                    rowdst[b] = srow2[b];
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) *reinterpret_cast<T*>(wbase + (wl ? asmo[1 + k] : SINK)) = cr[k];
                asm volatile("" ::: "memory");
                const bool w0 = u_commit && l == 0;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c <= r; ++c)
                        base[w0 ? LY::cv(RT + r, RT + c) : LY::DUM] = rr[r * (r + 1) / 2 + c];
            }
            sfence();
            UKFB_MARK("u_mean");
            // ---- new mean mu [+] delta: lane l < D adds ITS component of delta (it holds row l of the gain, so `del` is delta[l]) to the
            // stored component it belongs to; the three rotation lanes and lane 0 store the quaternion mu.q * exp(delta_r) instead
            {
                const T q[4] = {MUS[Q], MUS[Q + 1], MUS[Q + 2], MUS[Q + 3]};
                T nq[4];
                quat_mul(q, e0, nq);
                const bool rotl = (l >= RT) && (l < RT + 3);
                const int sl = (l < RT) ? l : (l + 1);                     // stored index of tangent component l outside the rotation
                const bool eu = u_commit && has_pair && !rotl;
                T old = MUS[has_pair ? sl : 0];
                keep(old);
                wsync();  // all lanes have read the old mean
                MUS[eu ? sl : (LY::DUM - LY::MUS)] = old + del;
                T* dst = (u_commit && l == 0) ? (MUS + Q) : DUMP;
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[k] = nq[k];
            }
            wsync();
        }
    }

    changed = changed || p_commit || u_commit;
    st_all |= st;
    if constexpr (PREFETCH) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pin.a[k] = nx_a[k];
            pin.w[k] = nx_w[k];
        }
        zq_l = nx_zq;
        mid_l = nx_mid;
    }
    } while (MULTI && ++cyc < ncyc);   // cycles

    UKFB_MARK("commit");
    // =========================================================================== commit
    const uint32_t st = st_all;
    const int64_t wgc = wg0_again();
    TS* const cov_c = a.cov + wgc * PK;
    TS* const mu_c = a.mu + wgc * S;
    uint32_t* const status_c = a.status + wgc;
    // (the store offsets are the prologue's load offsets; formed again from an opaque copy of the row's index, or the
    // compiler keeps all of them in registers across the whole kernel)
    IDX fcc = fc;
    if constexpr (!INDIRECT) asm volatile("" : "+v"(fcc));
    if (changed && fvalid) {
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const int e = l + G * t;
            if (e < PK) *at(cov_c, fcc * PK + IDX(e)) = TS(PKS[e]);
        }
        if (l < S) *at(mu_c, fcc * S + IDX(l)) = TS(MUS[l]);
    }
    if (fvalid && l == 0) *at(status_c, fcc) = status_accumulate_c ? (*at(status_c, fcc) | st) : st;
}

}  // namespace ukfb
