/* ukf_batch.h -- C-ABI of the MI355X-native batched UKF engine (libukf_batch.so).
 *
 * Drop-in boundary for the ONE hot path of rock-slam/slam-pose_estimation: everything the
 * reference reaches through its protected member
 *     boost::shared_ptr<ukfom::ukf<WState> > ukf;          (src/UnscentedKalmanFilter.hpp:150)
 * i.e. ukf->predict / ukf->update / ukf->mu() / ukf->sigma(), plus the per-filter plumbing around
 * it (time gate, latched inputs, process-noise shaping), for a BATCH of independent filters that
 * lives in the HBM of one MI355X.  Host C++ (include/pose_estimation/...hpp) and Python
 * (slam-pose_estimation_amd/engine.py, ctypes) sit on top of exactly these entry points.
 *
 * Conventions
 *  - plain C, opaque handle, int return codes (UKFB_OK == 0); no exception crosses the ABI.
 *  - the caller owns host buffers, the engine owns device buffers.
 *  - calls on one engine must come from one host thread at a time (the reference classes are
 *    boost::noncopyable and single-threaded, UnscentedKalmanFilter.hpp:16).
 *  - every call works on its engine's device and leaves the calling thread's current HIP device as it
 *    found it (one thread may drive engines on several devices, or share the thread with other HIP code).
 *  - work is enqueued on the engine's HIP stream; ukfb_sync() waits for it.  Functions that
 *    copy to host buffers synchronise themselves.
 *  - host-side numeric arrays are double and AoS ("host layout"):
 *      Pose   (UKFB_MODEL_POSE,   S=13, D=12): mu = position(3) orientation(x,y,z,w) velocity(3)
 *              angular_velocity(3)                              (PoseWithVelocity.hpp:18-23)
 *      Orient (UKFB_MODEL_ORIENT, S=14, D=13): mu = orientation(x,y,z,w) velocity(3) bias_gyro(3)
 *              bias_acc(3) gravity(1)                            (OrientationState.hpp:20-26)
 *      cov = D x D row-major (symmetric; the engine stores the lower triangle).
 *    Quaternion order (x,y,z,w) is Eigen's coefficient order.
 *  - "_dev" entry points take DEVICE pointers in the engine's compute precision (float or
 *    double) so that a caller whose inputs are already resident in HBM pays no PCIe copy.
 *    They are consumed on the engine's stream: buffers produced on another stream must be
 *    complete before the call (synchronise, or create the engine on the producer's stream with
 *    ukfb_create_on_stream), and must stay valid AND UNCHANGED until ukfb_sync() or a later
 *    synchronising call returns -- an engine that owns its stream may run a launch as two halves
 *    on two internal streams (ukfb_config.split_streams), so "the next call on the engine" is not
 *    a point after which an input buffer may be rewritten; ukfb_sync() is.
 *  - filters are independent (the reference's are separate objects): no filter's result depends on another filter's
 *    state or inputs.  Reproducibility: the same filter with the same inputs gives the same bits in any launch that
 *    places it in a wavefront whose other three filters run the same number of mean iterations (always the case for the
 *    same batch; also across batch sizes, shards and split launches in every workload of tests/ and bench.py);
 *    otherwise the results agree to the remainder of the re-based rotation deltas, below 4e-14 -- the iteration count
 *    is a wavefront's, a converged filter rides along unchanged.  Filters that commit nothing (uninitialised, gated
 *    out, not factorisable) never influence their wave-mates.
 *  - per-filter failures never abort a call: they are reported in the per-filter status word
 *    (UKFB_ST_*), and a failing filter keeps the state it had before the call (the reference
 *    throws before mutating: UnscentedKalmanFilter.hpp:110-124).
 */
#ifndef UKF_BATCH_H
#define UKF_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ukfb_engine ukfb_engine;

/* ---- return codes ---------------------------------------------------------------------- */
enum {
    UKFB_OK = 0,
    UKFB_ERR_INVALID_ARG = 1,
    UKFB_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime error at creation               */
    UKFB_ERR_HIP = 3,         /* HIP runtime error; text via ukfb_last_error()               */
    UKFB_ERR_OUT_OF_RANGE = 4,
    UKFB_ERR_WRONG_MODEL = 5  /* e.g. Orient-only call on a Pose engine                      */
};

/* ---- models / precision ----------------------------------------------------------------- */
enum { UKFB_MODEL_POSE = 0, UKFB_MODEL_ORIENT = 1 };
enum { UKFB_F64 = 0, UKFB_F32 = 1 };

/* measurement model ids = the integrateMeasurement overload they replace */
enum {
    UKFB_MEAS_NONE = -1,            /* filter takes no measurement in this call              */
    UKFB_MEAS_POS3 = 0,             /* PoseUKF.cpp:112-117  PositionMeasurement              */
    UKFB_MEAS_POS_XY = 1,           /* PoseUKF.cpp:119-124  XYMeasurement                    */
    UKFB_MEAS_POS_Z = 2,            /* PoseUKF.cpp:126-131  ZMeasurement                     */
    UKFB_MEAS_ORIENT_SO3 = 3,       /* PoseUKF.cpp:133-138  OrientationMeasurement (axis-angle in) */
    UKFB_MEAS_VEL3 = 4,             /* PoseUKF.cpp:140-145  VelocityMeasurement              */
    UKFB_MEAS_VEL_XY = 5,           /* PoseUKF.cpp:147-152  XYVelocityMeasurement            */
    UKFB_MEAS_VEL_Z = 6,            /* PoseUKF.cpp:154-159  ZVelocityMeasurement             */
    UKFB_MEAS_XVEL_YAWVEL = 7,      /* PoseUKF.cpp:161-166  XVelYawVelMeasurement            */
    UKFB_MEAS_ANGVEL3 = 8,          /* PoseUKF.cpp:168-173  AngularVelocityMeasurement       */
    UKFB_MEAS_ORIENT_BODYVEL3 = 9   /* OrientationUKF.cpp:65-72 VelocityMeasurement (Orient)  */
};

/* ---- per-filter status bits (uint32) ---------------------------------------------------- */
enum {
    UKFB_ST_OK = 0u,
    UKFB_ST_SKIPPED_FIRST_TS = 1u << 0,   /* UnscentedKalmanFilter.hpp:86-90                 */
    UKFB_ST_SKIPPED_SMALL_DT = 1u << 1,   /* UnscentedKalmanFilter.hpp:114-118               */
    UKFB_ST_ERR_NEG_DT = 1u << 2,         /* UnscentedKalmanFilter.hpp:110-113 (throws there) */
    UKFB_ST_ERR_DT_TOO_LARGE = 1u << 3,   /* UnscentedKalmanFilter.hpp:119-122 (throws there) */
    UKFB_ST_ERR_NONFINITE_MEAS = 1u << 4, /* UnscentedKalmanFilter.hpp:142-147 (throws there) */
    UKFB_ST_ERR_CHOLESKY = 1u << 5,       /* ukfom: Cholesky decomposition failed            */
    UKFB_ST_WARN_MEAN_NOCONV = 1u << 6,   /* ukfom: meanSigmaPoints() did not converge       */
    UKFB_ST_UNINITIALISED = 1u << 7,      /* UnscentedKalmanFilter.hpp:53,59                 */
    UKFB_ST_INACTIVE = 1u << 8,           /* filter masked out of this call                  */
    UKFB_ST_REJECTED_GATE = 1u << 9       /* mahalanobis gate rejected the update            */
};

/* ---- configuration ---------------------------------------------------------------------- */
typedef struct ukfb_config {
    double mean_tol;        /* ukfom meanSigmaPoints tolerance (1e-6)                        */
    int32_t mean_max_iter;  /* ukfom meanSigmaPoints cap (10000, as ukfom)                   */
    double gate_chi2;       /* < 0: accept_any_mahalanobis_distance (PoseUKF.cpp:116)        */
    double min_time_delta;  /* UnscentedKalmanFilter.hpp:31  (1e-9)                          */
    double max_time_delta;  /* UnscentedKalmanFilter.hpp:32  (DBL_MAX)                       */
    int32_t lanes_per_filter; /* 0 / 16 = the tuned layout (one DPP row per filter).  32 / 64 = one or two
                               * filters per wavefront, the brief's literal decomposition kept as an ablation:
                               * fp32 engines only in the shipped library (the fp64 instantiations need AGPRs and
                               * are a diagnostic build option, `make GENERIC_F64=1`); ukfb_layout_supported tells */
    int32_t bucket_models;  /* ukfb_cycle_dev with per-filter model ids (BASELINE config 5's mixed stream): 1 (default) = the
                             * filters are first grouped on the device by what their update costs -- no sample / a linear
                             * sub-state selection (PoseUKF.cpp:7-26,35-69) / the sigma-point path of
                             * OrientationMeasurement (PoseUKF.cpp:28-33,133-138) -- so that every wavefront runs ONE of
                             * the three paths; results are those of 0 (one launch in filter order) to rounding: a filter
                             * never reads another filter's data.  Batches below 16 384 filters are never bucketed.   */
    int32_t split_streams;  /* 1 (default): an engine that owns its stream (ukfb_create with stream = NULL) runs a launch over
                             * 16 384 ... 262 143 filters as two halves on two internal streams, so that the tail of one
                             * launch overlaps the head of the next (small launches lose 4-8 % to their partly empty last
                             * round of workgroups otherwise).  Bit-identical results; every other call joins the two streams
                             * first.  Engines on a caller's stream never split: plain stream order holds for them.        */
    int32_t wide_arithmetic; /* fp32 engines, tuned layout only; 0 (default) = fp32 arithmetic.  1 = the state, the inputs and the
                             * noise tables keep their fp32 format in HBM (same footprint, same C-ABI arrays), but every
                             * instruction of predict / update runs in fp64: values widen on load and narrow on commit.  The
                             * reference computes in fp64 throughout (Measurement.hpp:9-10); an fp32 evaluation of its recursion
                             * leaves the fp64 one by more than 1e-4 after 150 (OrientationState) / 500 (PoseWithVelocity)
                             * cycles of the bench workloads, and no cheaper mix of precisions holds it (DESIGN.md section 3,
                             * profiles/r04_f32_mixed_ab.txt).  This mode does, at the fp64 engine's rate.  Ignored by fp64
                             * engines; ukfb_set_config refuses it together with lanes_per_filter 32 / 64.                     */
    int32_t full_update_check; /* 0 (default): in a FUSED cycle without per-filter timestamps / time steps / activity masks / gate (the streams-only and plain kernels) the update
                             * factorises only the columns of the downdated covariance that applyDelta reads (RT + 3 of them) when
                             * positive definiteness of that covariance is already established: the filter's prediction was committed
                             * in the same launch (its input covariance factorised, the predicted one is a Gram matrix + noise), the
                             * batch-uniform process noise is positive semidefinite (checked on the host when it is set) and the
                             * sample's measurement covariance is positive definite (checked per filter in the kernel).  Results are
                             * bit-identical with the complete factorisation; what differs: a covariance that is indefinite through
                             * ROUNDING alone is reported by the next prediction's factorisation (UKFB_ST_ERR_CHOLESKY there), not by
                             * this update.  Any wavefront with a filter that does not meet the conditions, every update-only launch
                             * and every other kernel factorise completely, as ukfom's applyDelta does.  1: always complete.    */
} ukfb_config;

int ukfb_default_config(ukfb_config* cfg);
/* 1 if this build of the library can run `lanes_per_filter` lanes per filter at `precision`, else 0 */
int ukfb_layout_supported(int precision, int lanes_per_filter);

/* ---- lifetime ---------------------------------------------------------------------------- */
/* Replaces `new MTK_UKF(initial_state, state_cov)` per filter (UnscentedKalmanFilter.hpp:42)
 * by one engine holding `capacity` filters on HIP device `device`.  `stream` is a hipStream_t
 * (NULL: the engine creates its own non-blocking stream). */
int ukfb_create(ukfb_engine** out, int model, int precision, int64_t capacity, int device, void* stream);
/* The same, but `stream` is used exactly as given: NULL means the device's DEFAULT stream (hipStreamLegacy semantics),
 * not "create one".  For callers whose buffers are produced on that stream (e.g. a framework's current stream):
 * the "_dev" entry points then need no synchronisation between the producer and the engine. */
int ukfb_create_on_stream(ukfb_engine** out, int model, int precision, int64_t capacity, int device, void* stream);
/* Waits (bounded, see ukfb_sync) for the engine's stream, then frees everything.  If the stream never drains the
 * engine is abandoned without freeing device memory and UKFB_ERR_HIP is returned: exit the process. */
int ukfb_destroy(ukfb_engine* e);
const char* ukfb_last_error(void);
int ukfb_set_config(ukfb_engine* e, const ukfb_config* cfg);
int ukfb_get_config(const ukfb_engine* e, ukfb_config* cfg);
/* Waits for the engine's stream by polling, for at most UKFB_WAIT_TIMEOUT_S seconds (environment, default 120).  A
 * wait that gives up returns UKFB_ERR_HIP ("timed out ...") and POISONS the engine: work of unknown state is still
 * queued, every later call on it fails fast with UKFB_ERR_HIP, and the process is expected to exit non-zero. */
int ukfb_sync(ukfb_engine* e);

/* introspection: model, precision, capacity, S (stored scalars), D (DOF), packed cov length */
int ukfb_describe(const ukfb_engine* e, int* model, int* precision, int64_t* capacity, int* S, int* D, int* PK);

/* ---- state (UnscentedKalmanFilter.hpp:40-75) -------------------------------------------- */
/* initializeFilter(state, cov) for filters [first, first+count): sets mu, cov, marks the filter
 * initialised and zeroes its last_measurement_time (:40-44). */
int ukfb_initialize(ukfb_engine* e, int64_t first, int64_t count, const double* mu, const double* cov);
/* getCurrentState (:51-75): mu and/or cov may be NULL; initialised (uint8 per filter, may be
 * NULL) is the function's bool return. */
int ukfb_get_state(ukfb_engine* e, int64_t first, int64_t count, double* mu, double* cov, uint8_t* initialised);
/* per-filter status words of the most recent predict / update / cycle call */
int ukfb_get_status(ukfb_engine* e, int64_t first, int64_t count, uint32_t* status);
/* OR of all status words of the most recent call (cheap health check) */
int ukfb_get_status_summary(ukfb_engine* e, uint32_t* or_of_all);

/* get/setLastMeasurementTime (:131-133), int64 microseconds as base::Time */
int ukfb_set_last_measurement_time(ukfb_engine* e, int64_t first, int64_t count, const int64_t* t_us);
int ukfb_get_last_measurement_time(ukfb_engine* e, int64_t first, int64_t count, int64_t* t_us);

/* raw device views (engine precision): mu [capacity][S], packed lower-triangular cov
 * [capacity][PK] (row-major: index r(r+1)/2 + c, c <= r), status [capacity]. */
int ukfb_device_views(ukfb_engine* e, void** mu_dev, void** cov_packed_dev, uint32_t** status_dev);

/* ---- process noise / latched inputs ------------------------------------------------------ */
/* setProcessNoiseCovariance (:130): one D x D matrix for the whole batch ...
 * OrientationState engines: predictionStepImpl replaces the two leading 3x3 blocks N of the noise by R N R^T with the
 * rotation matrix R of the current orientation (OrientationUKF.cpp:81-86).  For a batch-uniform noise whose two blocks are
 * exact multiples of the identity, N = s I (the reference's zero default and every BASELINE configuration), the kernels
 * skip the rotation while every orientation quaternion of the wavefront has | |q|^2 - 1 | <= 1e-9 (fp64 engines) / 1e-4
 * (fp32 engines, where the stored norm drifts by ~1e-6 per hundred cycles): R s I R^T = s R R^T, and for Eigen's
 * un-normalised toRotationMatrix R R^T = I + O(|q|^2 - 1).  Deviation from the always-rotating reference: at most
 * 2 | |q|^2 - 1 | s dt^2 per entry and prediction -- 2e-4 relative to a noise entry that is itself ~1e-8 (config 4), i.e.
 * ~1e-12 absolute against covariance entries of 1e-4 ... 1e-2 whose fp32 rounding is 1e-11 ... 1e-9: below the
 * arithmetic's own rounding (profiles/r03_f32_drift_attribution.txt: the always-rotating build `d_iso` gives the same
 * distances to the oracle).  Anisotropic blocks, per-filter noise and non-unit quaternions take the rotated path. */
int ukfb_set_process_noise(ukfb_engine* e, const double* R);
/* ... or one per filter ([count][D][D]); the first per-filter call switches the engine to
 * per-filter storage (initialised from the batch-uniform matrix). */
int ukfb_set_process_noise_per_filter(ukfb_engine* e, int64_t first, int64_t count, const double* R);
int ukfb_get_process_noise(ukfb_engine* e, int64_t filter, double* R);

/* PoseUKF::integrateMeasurement(AccelerationMeasurement) (PoseUKF.cpp:175-178): latch acc.mu per
 * filter ([count][3]; a NaN row = "no acceleration", the ctor default PoseUKF.cpp:109) and the
 * batch-uniform acc.cov (3x3; NULL keeps the current one, default Identity, Measurement.hpp:12). */
int ukfb_pose_set_acceleration(ukfb_engine* e, int64_t first, int64_t count, const double* acc_mu,
                               const double* acc_cov);
/* device-resident variant: acc_mu_dev [capacity][3] in engine precision, used directly (no copy)
 * by later predict/cycle calls until replaced; NULL returns to the engine-owned buffer. */
int ukfb_pose_bind_acceleration_dev(ukfb_engine* e, const void* acc_mu_dev);

/* OrientationUKF ctor parameters (OrientationUKF.cpp:41-51): taus and earth rotation vector. */
int ukfb_orient_set_params(ukfb_engine* e, double gyro_bias_tau, double acc_bias_tau, const double earth_rotation[3]);
/* OrientationUKF::integrateMeasurement(RotationRate / Acceleration) (:53-63): latch per filter;
 * either pointer may be NULL to leave that input unchanged.  Non-finite rows are rejected
 * per filter (status ERR_NONFINITE_MEAS, previous value kept) as checkMeasurment does (:55,61). */
int ukfb_orient_set_inputs(ukfb_engine* e, int64_t first, int64_t count, const double* gyro, const double* acc);
int ukfb_orient_bind_inputs_dev(ukfb_engine* e, const void* gyro_dev, const void* acc_dev);
/* OrientationUKF::getRotationRate (:74-77) for filters [first, first+count): out [count][3] */
int ukfb_orient_get_rotation_rate(ukfb_engine* e, int64_t first, int64_t count, double* out);

/* ---- BodyStateMeasurement adapters (pose_with_velocity/BodyStateMeasurement.hpp:14-39) ---- */
/* One record per filter, 49 doubles: position(3) orientation(x,y,z,w) velocity(3) angular_velocity(3)
 * cov_position(9) cov_orientation(9) cov_velocity(9) cov_angular_velocity(9) (3x3 blocks, row-major) --
 * the fields of base::samples::RigidBodyState that the reference converts.
 * export = toRigidBodyState (:28-39): velocity is rotated into the navigation frame,
 *          velocity_out = orientation * velocity (:32); the blocks are the diagonal 3x3 blocks (:35-38).
 * import = fromRigidBodyState (:14-26) followed by initializeFilter: fields copied as they are (no inverse
 *          rotation, as in the reference), covariance = the four blocks at (0,0) (3,3) (6,6) (9,9), zero
 *          elsewhere (:21-25).  Pose engines only. */
#define UKFB_BODY_STATE_SCALARS 49
int ukfb_pose_export_body_states(ukfb_engine* e, int64_t first, int64_t count, double* out);
int ukfb_pose_import_body_states(ukfb_engine* e, int64_t first, int64_t count, const double* in);

/* ---- predict (UnscentedKalmanFilter.hpp:83-125 + predictionStepImpl) --------------------- */
/* predictionStep(delta_t) with one dt for every filter */
int ukfb_predict(ukfb_engine* e, double dt);
/* predictionStep(delta_t[i]) (host array [capacity]) */
int ukfb_predict_dt(ukfb_engine* e, const double* dt);
/* predictionStepFromSampleTime(ts[i]) (host array [capacity], int64 microseconds) */
int ukfb_predict_timestamps(ukfb_engine* e, const int64_t* ts_us);
/* device-resident variants (double / int64 device arrays [capacity]) */
int ukfb_predict_dt_dev(ukfb_engine* e, const double* dt_dev);
int ukfb_predict_timestamps_dev(ukfb_engine* e, const int64_t* ts_us_dev);

/* ---- update (integrateMeasurement -> ukf->update) ---------------------------------------- */
/* One model id for the batch.  z [capacity][3] (first m entries used; axis-angle for ORIENT_SO3),
 * Q [capacity][3][3] (leading m x m block used), active (uint8 [capacity], may be NULL = all). */
int ukfb_update(ukfb_engine* e, int meas_model, const double* z, const double* Q, const uint8_t* active);
/* per-filter model ids (int32 [capacity]; negative = no measurement for that filter) */
int ukfb_update_mixed(ukfb_engine* e, const int32_t* meas_model, const double* z, const double* Q);
/* device-resident variants: z_dev [capacity][3], Q_dev [capacity][9] in engine precision;
 * meas_model_dev int32 [capacity] or NULL (then meas_model_uniform applies to every filter). */
int ukfb_update_dev(ukfb_engine* e, int meas_model_uniform, const int32_t* meas_model_dev, const void* z_dev,
                    const void* Q_dev);

/* ---- fused cycle: predictionStep(dt) followed by integrateMeasurement, one launch -------- */
/* The host-array forms (ukfb_cycle, ukfb_cycle_uniform_q) upload their samples on a separate copy stream into one of two
 * staging sets, so that the upload of call k + 1 overlaps the kernel of call k; they return when the caller's arrays have been
 * consumed (the arrays may be rewritten at once), not when the kernel is done. */
int ukfb_cycle(ukfb_engine* e, double dt, int meas_model, const double* z, const double* Q);
int ukfb_cycle_dev(ukfb_engine* e, double dt, int meas_model_uniform, const int32_t* meas_model_dev, const void* z_dev,
                   const void* Q_dev);

/* One measurement covariance for the whole batch (the usual case: a sensor's covariance is a constant): Q9 is ONE 3x3
 * (9 doubles / 9 engine-precision scalars on the device; leading m x m block used).  Same results as the per-filter
 * forms with that matrix repeated; a quarter of the bytes across PCIe and no per-filter Q stream in the kernel. */
int ukfb_update_uniform_q(ukfb_engine* e, int meas_model, const double* z, const double* Q9, const uint8_t* active);
int ukfb_cycle_uniform_q(ukfb_engine* e, double dt, int meas_model, const double* z, const double* Q9);
int ukfb_cycle_uniform_q_dev(ukfb_engine* e, double dt, int meas_model, const void* z_dev, const void* Q9_dev);

/* `cycles` consecutive fused cycles in ONE launch (replay of buffered samples, catching up after a stall, fixed-rate
 * sensors whose samples are batched): predictionStep(dt) + integrateMeasurement(meas_model) `cycles` times for every
 * filter, exactly as `cycles` calls of ukfb_cycle_dev would -- same arithmetic, bit-identical state -- but the filter
 * stays in LDS between its cycles: the state crosses HBM once per launch instead of once per cycle.
 * The samples sit in rings of `slots` input sets on the device, engine precision: z_dev [slots][capacity][3],
 * Q_dev [slots][capacity][9]; cycle c (0-based) reads slot (first_slot + c) % slots.  in_a_dev / in_b_dev
 * ([slots][capacity][3], either may be NULL) replace the latched inputs per cycle in the same way -- Pose: in_a =
 * acceleration (acc.mu of integrateMeasurement(AccelerationMeasurement), PoseUKF.cpp:175-178; NaN = no acceleration,
 * constant-velocity branch), in_b unused; Orient: in_a = acceleration.mu, in_b = rotation_rate.mu
 * (OrientationUKF.cpp:53-63); NULL: the inputs latched in the engine serve every cycle.  The latches themselves are
 * not modified.  After the call the status word of a filter is the OR over its cycles; a cycle whose prediction is
 * gated or fails is skipped for that filter as in a single launch, the following cycles still run. */
int ukfb_cycle_multi_dev(ukfb_engine* e, int cycles, double dt, int meas_model, int slots, int first_slot,
                         const void* in_a_dev, const void* in_b_dev, const void* z_dev, const void* Q_dev);
/* the same with per-filter model ids per cycle (the asynchronous mixed stream of BASELINE config 5, buffered):
 * meas_model_dev is a ring int32 [slots][capacity] like z and Q, negative = no measurement for that filter in that cycle
 * (prediction only, status INACTIVE as in ukfb_cycle_dev) */
int ukfb_cycle_multi_mixed_dev(ukfb_engine* e, int cycles, double dt, int slots, int first_slot, const void* in_a_dev,
                               const void* in_b_dev, const int32_t* meas_model_dev, const void* z_dev, const void* Q_dev);
/* ukfb_cycle_multi_dev with a schedule: cycle c predicts by dt[c] and updates with model meas_model[c] (host arrays of `cycles`
 * entries; a negative model = prediction only in that cycle, i.e. a plain predictionStep).  One launch per 32 cycles.  This is an IMU-rate filter with slower aiding sensors replayed from a buffer: e.g. ten 100 Hz
 * predictions of which the last one carries a 10 Hz position fix (BASELINE config 2's workload) in one launch. */
int ukfb_cycle_schedule_dev(ukfb_engine* e, int cycles, const double* dt, const int32_t* meas_model, int slots, int first_slot,
                            const void* in_a_dev, const void* in_b_dev, const void* z_dev, const void* Q_dev);
/* ukfb_cycle_multi_dev from host arrays of doubles, one input set per cycle: z [cycles][capacity][3], Q [cycles][capacity][3][3],
 * in_a / in_b [cycles][capacity][3] or NULL (uploaded to an engine-owned ring, then one launch) */
int ukfb_cycle_multi(ukfb_engine* e, int cycles, double dt, int meas_model, const double* in_a, const double* in_b,
                     const double* z, const double* Q);

/* fused predictionStepFromSampleTime(ts[i]) + integrateMeasurement(model[i]) per filter, one launch.
 * ts_us[i] < 0: filter i has no sample in this call (untouched, status INACTIVE);
 * model[i] < 0: prediction only.  Host arrays [capacity] / device arrays in engine precision. */
int ukfb_cycle_timestamps(ukfb_engine* e, const int64_t* ts_us, const int32_t* meas_model, const double* z, const double* Q);
int ukfb_cycle_timestamps_dev(ukfb_engine* e, const int64_t* ts_us_dev, const int32_t* meas_model_dev, const void* z_dev,
                              const void* Q_dev);

/* ---- time-ordered asynchronous measurement stream ------------------------------------------ */
/* What Rock's stream aligner (drivers/aggregator, manifest.xml:14) does for one filter, for a batch:
 * n_events samples (filter index, timestamp, model id, z[3], Q[3][3]) arrive in ANY order.  Per filter
 * they are applied in timestamp order (stable for equal stamps), each as
 * predictionStepFromSampleTime(ts) followed by integrateMeasurement(model) (model < 0: prediction
 * only).  Filters are independent, so the r-th sample of every filter forms round r and each round
 * is one fused launch.  After the call ukfb_get_status returns, per filter, the OR of its status
 * words over all rounds; status_or / rounds (may be NULL) receive the batch-wide OR and the number
 * of launches. */
int ukfb_process_events(ukfb_engine* e, int64_t n_events, const int64_t* filter, const int64_t* ts_us,
                        const int32_t* meas_model, const double* z, const double* Q, uint32_t* status_or, int64_t* rounds);
/* Same stream already resident in HBM (z, Q in the engine's precision): ordering (two stable radix sorts:
 * timestamp, then filter index), ranking and the per-round scatter all run on the device; the host only
 * learns the number of rounds.  The host-pointer variant above uploads the five arrays and calls this. */
int ukfb_process_events_dev(ukfb_engine* e, int64_t n_events, const int64_t* filter_dev, const int64_t* ts_us_dev,
                            const int32_t* meas_model_dev, const void* z_dev, const void* Q_dev, uint32_t* status_or,
                            int64_t* rounds);


/* ---- device groups: one host process, several MI355X ------------------------------------------------------------------ */
/* north_star's multi-GPU shape for a C++ host.  The filters of a batch are independent -- every filter of the reference owns
 * its own `ukf` object (src/UnscentedKalmanFilter.hpp:150) -- so `total_filters` split into contiguous shards (the first
 * total % n shards own one filter more; ukfb_group_shard_range), one engine with its own stream per device, and NO
 * collective on the data path.  The calls below fan out to the shards from the calling thread: the hot-path calls only
 * enqueue (the devices then run concurrently), ukfb_group_sync waits for all of them.  The one exchange is the result
 * gather, an RCCL all-gather of the means over xGMI.  Errors: the usual codes, text in ukfb_last_error().
 * `devices` may name a device more than once (several shards on one GPU: everything but the gather works). */
typedef struct ukfb_group ukfb_group;
int ukfb_group_shard_range(int64_t total, int n_shards, int shard, int64_t* first, int64_t* count);
int ukfb_group_create(ukfb_group** out, int model, int precision, int64_t total_filters, const int* devices, int n_devices);
int ukfb_group_destroy(ukfb_group* g);
int ukfb_group_size(const ukfb_group* g);   /* shards, -1 for NULL */
/* shard `shard`: its engine (every per-engine call above applies to it), device, first filter and filter count */
int ukfb_group_shard(ukfb_group* g, int shard, ukfb_engine** engine, int* device, int64_t* first, int64_t* count);
int ukfb_group_set_config(ukfb_group* g, const ukfb_config* cfg);
/* whole-batch host arrays, [first, first + count) in BATCH numbering, routed to the shards that own the filters */
int ukfb_group_initialize(ukfb_group* g, int64_t first, int64_t count, const double* mu, const double* cov);
int ukfb_group_get_state(ukfb_group* g, int64_t first, int64_t count, double* mu, double* cov, uint8_t* initialised);
int ukfb_group_get_status(ukfb_group* g, int64_t first, int64_t count, uint32_t* status);
int ukfb_group_get_status_summary(ukfb_group* g, uint32_t* or_of_all);
int ukfb_group_set_process_noise(ukfb_group* g, const double* R);
int ukfb_group_pose_set_acceleration(ukfb_group* g, int64_t first, int64_t count, const double* acc_mu, const double* acc_cov);
int ukfb_group_orient_set_params(ukfb_group* g, double gyro_bias_tau, double acc_bias_tau, const double earth_rotation[3]);
int ukfb_group_orient_set_inputs(ukfb_group* g, int64_t first, int64_t count, const double* gyro, const double* acc);
/* hot path from host arrays over the whole batch (z [total][3], Q [total][3][3]).  From 32 768 filters on, the host-array calls
 * of a group (these, ukfb_group_initialize, ukfb_group_get_state, ukfb_group_cycle_timestamps) run one host thread per shard for
 * the duration of the call, so that the uploads of all devices proceed at once over their own PCIe links. */
int ukfb_group_predict(ukfb_group* g, double dt);
int ukfb_group_update(ukfb_group* g, int meas_model, const double* z, const double* Q);
int ukfb_group_cycle(ukfb_group* g, double dt, int meas_model, const double* z, const double* Q);
/* hot path from device-resident inputs: arrays of one device pointer PER SHARD (index = shard, memory on that shard's
 * device, engine precision, sized for the shard's filters) -- the shapes of ukfb_cycle_dev / ukfb_cycle_multi_dev /
 * ukfb_pose_bind_acceleration_dev / ukfb_orient_bind_inputs_dev */
int ukfb_group_pose_bind_acceleration_dev(ukfb_group* g, const void* const* acc_mu_dev);
int ukfb_group_orient_bind_inputs_dev(ukfb_group* g, const void* const* gyro_dev, const void* const* acc_dev);
int ukfb_group_cycle_dev(ukfb_group* g, double dt, int meas_model, const void* const* z_dev, const void* const* Q_dev);
int ukfb_group_cycle_multi_dev(ukfb_group* g, int cycles, double dt, int meas_model, int slots, int first_slot,
                               const void* const* in_a_dev, const void* const* in_b_dev, const void* const* z_dev,
                               const void* const* Q_dev);
/* per-filter model ids resident on the devices (ukfb_cycle_dev with meas_model_dev), one pointer per shard */
int ukfb_group_cycle_mixed_dev(ukfb_group* g, double dt, const int32_t* const* meas_model_dev, const void* const* z_dev,
                               const void* const* Q_dev);
/* ukfb_cycle_timestamps over the whole batch: host arrays [total] in batch numbering */
int ukfb_group_cycle_timestamps(ukfb_group* g, const int64_t* ts_us, const int32_t* meas_model, const double* z, const double* Q);
/* ukfb_process_events over the whole batch: `filter` in batch numbering, any arrival order.  Events are routed to the shard
 * that owns their filter (stable: per filter the arrival order survives), the shards order and apply their events
 * concurrently (one host thread per shard for the duration of the call).  rounds = the launches of the shard that needed most;
 * statuses as ukfb_process_events leaves them (a filter without samples: 0). */
int ukfb_group_process_events(ukfb_group* g, int64_t n_events, const int64_t* filter, const int64_t* ts_us,
                              const int32_t* meas_model, const double* z, const double* Q, uint32_t* status_or, int64_t* rounds);
int ukfb_group_sync(ukfb_group* g);
/* HIP-event timing on every shard's stream; elapsed_ms_max = the slowest shard, elapsed_ms_per_shard [shards] may be NULL */
int ukfb_group_timer_begin(ukfb_group* g);
int ukfb_group_timer_end(ukfb_group* g, float* elapsed_ms_max, float* elapsed_ms_per_shard);
/* Result gather: out_dev[shard] -- memory on that shard's device, engine precision, [total_filters][S] -- receives the mean
 * states of ALL filters in batch order on every device: ncclAllGather over one communicator per device (ncclCommInitAll at
 * the first call; RCCL is loaded at run time, librccl.so.1).  Stream-ordered after the launches enqueued so far; complete
 * after ukfb_group_sync.  Gather at the end of a run or every K cycles, not per cycle: at 131 072 Pose filters per device
 * the means are 6.8 MB (fp32) per shard, about the duration of one cycle over xGMI.  A group whose shards SHARE a device
 * (where RCCL has no rank to give them; e.g. a rehearsal of N shards on one GPU) exchanges by peer copies between the shards'
 * streams instead; staging and the ragged compaction are the same code either way.  ukfb_group_last_gather_exchange tells
 * which exchange the last gather used: 1 = RCCL all-gather, 2 = peer copies, 0 = no gather yet, -1 = NULL group. */
int ukfb_group_gather_means(ukfb_group* g, void* const* out_dev);
int ukfb_group_last_gather_exchange(const ukfb_group* g);

/* ---- measurement of the engine itself ---------------------------------------------------- */
/* name, dynamic LDS bytes per workgroup, filters per workgroup and grid size of the kernel the
 * most recent predict/update/cycle call launched (for profiles/ and bench.py) */
int ukfb_last_launch_info(const ukfb_engine* e, char* kernel_name, int name_capacity, int* lds_bytes,
                          int* filters_per_workgroup, int64_t* grid);
/* the filter list of the most recent launch that grouped its filters by update class (ukfb_cycle_dev with per-filter
 * model ids, ukfb_config.bucket_models): *items = entries the launch covered, the first min(capacity, *items) of them
 * copied to list -- filter indices, class by class (sigma-point updates first, then the linear selections, then the
 * filters without a sample), every class starting at a multiple of 4, -1 = padding.  UKFB_ERR_INVALID_ARG when no
 * launch of the engine has been grouped. */
int ukfb_last_model_groups(ukfb_engine* e, int32_t* list, int64_t capacity, int64_t* items);
/* HIP-event timing on the engine's stream: begin/end bracket a region, elapsed in ms */
int ukfb_timer_begin(ukfb_engine* e);
int ukfb_timer_end(ukfb_engine* e, float* elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* UKF_BATCH_H */
