// pose_estimation/Batch.hpp -- batched C++ siblings of PoseUKF / OrientationUKF: one object owns N
// filters resident on one MI355X.  Thin RAII over include/ukf_batch.h; arrays are AoS doubles in the
// layouts documented there.  This is the interface the engine is built for (a Rock component that
// tracks N hypotheses / particles / vehicles calls these instead of N scalar objects).
#ifndef _POSE_ESTIMATION_BATCH_HPP
#define _POSE_ESTIMATION_BATCH_HPP

#include <ukf_batch.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace pose_estimation
{

class BatchUKF
{
public:
    BatchUKF(int model, int precision, int64_t capacity, int device = 0, void* hip_stream = NULL) : engine(NULL)
    {
        if (ukfb_create(&engine, model, precision, capacity, device, hip_stream) != UKFB_OK)
            throw std::runtime_error(std::string("pose_estimation: MI355X engine unavailable: ") + ukfb_last_error());
        ukfb_describe(engine, NULL, NULL, &cap, &S, &D, &PK);
    }
    virtual ~BatchUKF() { ukfb_destroy(engine); }

    int64_t capacity() const { return cap; }
    int storedSize() const { return S; }
    int dof() const { return D; }
    ukfb_engine* handle() { return engine; }

    /** initializeFilter for filters [first, first + count) */
    virtual void initializeFilters(int64_t first, int64_t count, const double* mu, const double* cov) { check(ukfb_initialize(engine, first, count, mu, cov)); }
    /** getCurrentState for filters [first, first + count); cov may be NULL */
    void getCurrentStates(int64_t first, int64_t count, double* mu, double* cov, uint8_t* initialised = NULL) { check(ukfb_get_state(engine, first, count, mu, cov, initialised)); }
    void setProcessNoiseCovariance(const double* R) { check(ukfb_set_process_noise(engine, R)); }
    void setProcessNoiseCovariances(int64_t first, int64_t count, const double* R) { check(ukfb_set_process_noise_per_filter(engine, first, count, R)); }
    void setLastMeasurementTimes(int64_t first, int64_t count, const int64_t* t_us) { check(ukfb_set_last_measurement_time(engine, first, count, t_us)); }
    void setTimeDeltas(double min_dt, double max_dt)
    {
        ukfb_config c; ukfb_get_config(engine, &c); c.min_time_delta = min_dt; c.max_time_delta = max_dt; check(ukfb_set_config(engine, &c));
    }
    /** predictionStep(delta_t) for every filter; per-filter outcomes in status() */
    void predictionStep(double delta_t) { check(ukfb_predict(engine, delta_t)); }
    void predictionSteps(const double* delta_t) { check(ukfb_predict_dt(engine, delta_t)); }
    /** predictionStepFromSampleTime(ts[i]) per filter (int64 microseconds) */
    void predictionStepsFromSampleTimes(const int64_t* ts_us) { check(ukfb_predict_timestamps(engine, ts_us)); }
    /** integrateMeasurement with one model id (UKFB_MEAS_*) for the batch */
    void integrateMeasurements(int model, const double* z, const double* Q, const uint8_t* active = NULL) { check(ukfb_update(engine, model, z, Q, active)); }
    /** per-filter model ids (negative = none): the asynchronous mixed stream of BASELINE config 5 */
    void integrateMixedMeasurements(const int32_t* model, const double* z, const double* Q) { check(ukfb_update_mixed(engine, model, z, Q)); }
    /** fused predictionStep + integrateMeasurement in one launch */
    void cycle(double delta_t, int model, const double* z, const double* Q) { check(ukfb_cycle(engine, delta_t, model, z, Q)); }
    /** `cycles` fused cycles in one launch, the filters stay on chip in between: buffered fixed-rate samples, one input set
     *  per cycle (z [cycles][N][3], Q [cycles][N][9]; in_a / in_b [cycles][N][3] or NULL = the latched inputs).
     *  Pose: in_a = acceleration; Orient: in_a = acceleration, in_b = rotation rate */
    void cycles(int cycles, double delta_t, int model, const double* in_a, const double* in_b, const double* z, const double* Q)
    {
        check(ukfb_cycle_multi(engine, cycles, delta_t, model, in_a, in_b, z, Q));
    }
    /** fused predictionStepFromSampleTime(ts[i]) + integrateMeasurement(model[i]); ts < 0: no sample, model < 0: predict only */
    void cycleFromSampleTimes(const int64_t* ts_us, const int32_t* model, const double* z, const double* Q) { check(ukfb_cycle_timestamps(engine, ts_us, model, z, Q)); }
    /** time-ordered asynchronous stream of samples in any arrival order (the batched stream aligner); returns the number of launches */
    int64_t processEvents(int64_t n_events, const int64_t* filter, const int64_t* ts_us, const int32_t* model, const double* z, const double* Q)
    {
        int64_t rounds = 0;
        check(ukfb_process_events(engine, n_events, filter, ts_us, model, z, Q, NULL, &rounds));
        return rounds;
    }
    std::vector<uint32_t> status()
    {
        std::vector<uint32_t> st(static_cast<size_t>(cap), 0u);
        check(ukfb_get_status(engine, 0, cap, st.data()));
        return st;
    }
    void sync() { check(ukfb_sync(engine)); }

protected:
    void check(int rc) const { if (rc != UKFB_OK) throw std::runtime_error(std::string("pose_estimation engine: ") + ukfb_last_error()); }
    ukfb_engine* engine;
    int64_t cap;
    int S, D, PK;

private:
    BatchUKF(const BatchUKF&);
    BatchUKF& operator=(const BatchUKF&);
};

class BatchPoseUKF : public BatchUKF
{
public:
    BatchPoseUKF(int64_t capacity, int precision = UKFB_F64, int device = 0) : BatchUKF(UKFB_MODEL_POSE, precision, capacity, device)
    {
        double R[144] = {0};   // PoseUKF.cpp:103-107
        for (int k = 0; k < 3; ++k) { R[k * 13] = 0.01; R[(3 + k) * 13] = 0.001; R[(6 + k) * 13] = 0.00001; R[(9 + k) * 13] = 0.00001; }
        setProcessNoiseCovariance(R);
    }
    /** integrateMeasurement(AccelerationMeasurement) per filter; NaN row = none (PoseUKF.cpp:109,175-178) */
    void setAccelerations(int64_t first, int64_t count, const double* acc_mu, const double* acc_cov3x3) { check(ukfb_pose_set_acceleration(engine, first, count, acc_mu, acc_cov3x3)); }
};

class BatchOrientationUKF : public BatchUKF
{
public:
    BatchOrientationUKF(int64_t capacity, double gyro_bias_tau, double acc_bias_tau, const double earth_rotation[3], int precision = UKFB_F64, int device = 0)
        : BatchUKF(UKFB_MODEL_ORIENT, precision, capacity, device)
    {
        check(ukfb_orient_set_params(engine, gyro_bias_tau, acc_bias_tau, earth_rotation));
    }
    /** initializeFilter plus the reference constructor's input latches (OrientationUKF.cpp:49-50):
     *  rotation_rate.mu = 0 and acceleration.mu = (0, 0, initial gravity), so that a prediction before the first
     *  IMU sample holds the velocity steady exactly as the scalar class does. */
    virtual void initializeFilters(int64_t first, int64_t count, const double* mu, const double* cov)
    {
        BatchUKF::initializeFilters(first, count, mu, cov);
        std::vector<double> gyro(static_cast<size_t>(count) * 3, 0.0), acc(static_cast<size_t>(count) * 3, 0.0);
        for (int64_t i = 0; i < count; ++i) acc[static_cast<size_t>(i) * 3 + 2] = mu[static_cast<size_t>(i) * 14 + 13];
        setInputs(first, count, gyro.data(), acc.data());
    }
    void setInputs(int64_t first, int64_t count, const double* gyro, const double* acc) { check(ukfb_orient_set_inputs(engine, first, count, gyro, acc)); }
    void getRotationRates(int64_t first, int64_t count, double* out) { check(ukfb_orient_get_rotation_rate(engine, first, count, out)); }
};

/** One host process, several MI355X: `total` independent filters in contiguous shards, one engine per device
 *  (ukfb_group_* of ukf_batch.h).  Filters never read each other (UnscentedKalmanFilter.hpp:150), so the shards need no
 *  collective on the data path; gatherMeans is the one exchange (RCCL all-gather over xGMI).  Whole-batch host arrays are
 *  in batch numbering; device-pointer arguments are one pointer PER SHARD (memory on that shard's device). */
class ShardedBatchUKF
{
public:
    virtual ~ShardedBatchUKF() { ukfb_group_destroy(group); }

    int64_t capacity() const { return n; }
    int shards() const { return ukfb_group_size(group); }
    ukfb_group* handle() { return group; }
    /** engine of one shard (every single-engine call of ukf_batch.h applies), its device and filter range */
    ukfb_engine* shard(int r, int* device = NULL, int64_t* first = NULL, int64_t* count = NULL)
    {
        ukfb_engine* e = NULL;
        check(ukfb_group_shard(group, r, &e, device, first, count));
        return e;
    }
    virtual void initializeFilters(int64_t first, int64_t count, const double* mu, const double* cov) { check(ukfb_group_initialize(group, first, count, mu, cov)); }
    void getCurrentStates(int64_t first, int64_t count, double* mu, double* cov, uint8_t* initialised = NULL) { check(ukfb_group_get_state(group, first, count, mu, cov, initialised)); }
    void setProcessNoiseCovariance(const double* R) { check(ukfb_group_set_process_noise(group, R)); }
    void predictionStep(double delta_t) { check(ukfb_group_predict(group, delta_t)); }
    void integrateMeasurements(int model, const double* z, const double* Q) { check(ukfb_group_update(group, model, z, Q)); }
    void cycle(double delta_t, int model, const double* z, const double* Q) { check(ukfb_group_cycle(group, delta_t, model, z, Q)); }
    /** samples already resident on the devices: z_dev[r] / Q_dev[r] on shard r's device, engine precision */
    void cycleDev(double delta_t, int model, const void* const* z_dev, const void* const* Q_dev) { check(ukfb_group_cycle_dev(group, delta_t, model, z_dev, Q_dev)); }
    /** per-filter model ids resident on the devices: meas_model_dev[r] = int32 [filters of shard r] */
    void cycleMixedDev(double delta_t, const int32_t* const* meas_model_dev, const void* const* z_dev, const void* const* Q_dev)
    {
        check(ukfb_group_cycle_mixed_dev(group, delta_t, meas_model_dev, z_dev, Q_dev));
    }
    /** fused predictionStepFromSampleTime(ts[i]) + integrateMeasurement(model[i]) per filter, host arrays over the whole batch */
    void cycleFromSampleTimes(const int64_t* ts_us, const int32_t* model, const double* z, const double* Q) { check(ukfb_group_cycle_timestamps(group, ts_us, model, z, Q)); }
    /** time-ordered asynchronous stream over the sharded batch (filter indices in batch numbering); returns the launches of the shard that needed most */
    int64_t processEvents(int64_t n_events, const int64_t* filter, const int64_t* ts_us, const int32_t* model, const double* z, const double* Q)
    {
        int64_t rounds = 0;
        check(ukfb_group_process_events(group, n_events, filter, ts_us, model, z, Q, NULL, &rounds));
        return rounds;
    }
    /** RCCL all-gather: out_dev[r] ([total][S], engine precision, on shard r's device) receives every filter's mean */
    void gatherMeans(void* const* out_dev) { check(ukfb_group_gather_means(group, out_dev)); }
    uint32_t statusSummary() { uint32_t v = 0; check(ukfb_group_get_status_summary(group, &v)); return v; }
    void sync() { check(ukfb_group_sync(group)); }

protected:
    ShardedBatchUKF(int model, int64_t total, const std::vector<int>& devices, int precision) : group(NULL), n(total)
    {
        if (ukfb_group_create(&group, model, precision, total, devices.data(), static_cast<int>(devices.size())) != UKFB_OK)
            throw std::runtime_error(std::string("pose_estimation: MI355X engine group unavailable: ") + ukfb_last_error());
    }
    void check(int rc) const { if (rc != UKFB_OK) throw std::runtime_error(std::string("pose_estimation engine group: ") + ukfb_last_error()); }
    ukfb_group* group;
    int64_t n;

private:
    ShardedBatchUKF(const ShardedBatchUKF&);
    ShardedBatchUKF& operator=(const ShardedBatchUKF&);
};

class ShardedBatchPoseUKF : public ShardedBatchUKF
{
public:
    ShardedBatchPoseUKF(int64_t total, const std::vector<int>& devices, int precision = UKFB_F64) : ShardedBatchUKF(UKFB_MODEL_POSE, total, devices, precision)
    {
        double R[144] = {0};   // PoseUKF.cpp:103-107
        for (int k = 0; k < 3; ++k) { R[k * 13] = 0.01; R[(3 + k) * 13] = 0.001; R[(6 + k) * 13] = 0.00001; R[(9 + k) * 13] = 0.00001; }
        setProcessNoiseCovariance(R);
    }
    void setAccelerations(int64_t first, int64_t count, const double* acc_mu, const double* acc_cov3x3) { check(ukfb_group_pose_set_acceleration(group, first, count, acc_mu, acc_cov3x3)); }
    void bindAccelerationsDev(const void* const* acc_mu_dev) { check(ukfb_group_pose_bind_acceleration_dev(group, acc_mu_dev)); }
};

class ShardedBatchOrientationUKF : public ShardedBatchUKF
{
public:
    ShardedBatchOrientationUKF(int64_t total, const std::vector<int>& devices, double gyro_bias_tau, double acc_bias_tau, const double earth_rotation[3],
                               int precision = UKFB_F64)
        : ShardedBatchUKF(UKFB_MODEL_ORIENT, total, devices, precision)
    {
        check(ukfb_group_orient_set_params(group, gyro_bias_tau, acc_bias_tau, earth_rotation));
    }
    /** initializeFilter plus the reference constructor's input latches (OrientationUKF.cpp:49-50), as BatchOrientationUKF */
    virtual void initializeFilters(int64_t first, int64_t count, const double* mu, const double* cov)
    {
        ShardedBatchUKF::initializeFilters(first, count, mu, cov);
        std::vector<double> gyro(static_cast<size_t>(count) * 3, 0.0), acc(static_cast<size_t>(count) * 3, 0.0);
        for (int64_t i = 0; i < count; ++i) acc[static_cast<size_t>(i) * 3 + 2] = mu[static_cast<size_t>(i) * 14 + 13];
        setInputs(first, count, gyro.data(), acc.data());
    }
    void setInputs(int64_t first, int64_t count, const double* gyro, const double* acc) { check(ukfb_group_orient_set_inputs(group, first, count, gyro, acc)); }
    void bindInputsDev(const void* const* gyro_dev, const void* const* acc_dev) { check(ukfb_group_orient_bind_inputs_dev(group, gyro_dev, acc_dev)); }
};

}

#endif
