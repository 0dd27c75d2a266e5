// pose_estimation/UnscentedKalmanFilter.hpp -- host mirror of the reference's filter base template
// (src/UnscentedKalmanFilter.hpp:15-155): same public surface, same defaults, same exceptions.  The
// protected `boost::shared_ptr<MTK_UKF> ukf` member (:150) is replaced by a handle into the MI355X
// engine (include/ukf_batch.h); there is NO CPU arithmetic behind this class -- construction throws if
// no HIP device is present.  One object = a batch of one filter; the batched siblings
// (pose_estimation/Batch.hpp) are what the engine is built for.
#ifndef _POSE_ESTIMATION_UKF_HPP
#define _POSE_ESTIMATION_UKF_HPP

#include <base/Time.hpp>
#include <pose_estimation/Types.hpp>
#include <ukf_batch.h>

#include <limits>
#include <stdexcept>
#include <string>

namespace pose_estimation
{

// Manifold requirements: enum { DOF, STORED, ENGINE_MODEL }; void toArray(double*) const;
// void fromArray(const double*).
template<typename Manifold>
class UnscentedKalmanFilter
{
public:
    enum {
        DOF = Manifold::DOF
    };
    typedef Manifold State;
    typedef ukfom::mtkwrap<Manifold> WState;       // :23  (value type with operator+ / operator-, pose_estimation/Manifold.hpp)
    typedef ukfom::ukf<WState> MTK_UKF;            // :24  (the type names cov / scalar_type / state; the arithmetic is the engine's)
    typedef typename MTK_UKF::cov Covariance;      // :25

    UnscentedKalmanFilter() : engine(NULL), initialised(false)
    {
        process_noise_cov = Covariance::Zero();                    // :29
        last_measurement_time.microseconds = 0;                    // :30
        min_time_delta = 1.0e-9;                                   // :31
        max_time_delta = std::numeric_limits<double>::max();       // :32
        int rc = ukfb_create(&engine, Manifold::ENGINE_MODEL, UKFB_F64, 1, 0, NULL);
        if (rc != UKFB_OK)
            throw std::runtime_error(std::string("pose_estimation: MI355X engine unavailable: ") + ukfb_last_error());
        ukfb_get_config(engine, &engine_cfg);
        pushed_min_time_delta = engine_cfg.min_time_delta;
        pushed_max_time_delta = engine_cfg.max_time_delta;
    }

    virtual ~UnscentedKalmanFilter() { ukfb_destroy(engine); }

    /** (Re-)initializes the UKF filter from a given state (:40-44). */
    void initializeFilter(const State& initial_state, const Covariance& state_cov)
    {
        double mu[Manifold::STORED];
        initial_state.toArray(mu);
        double cov_rm[Manifold::DOF * Manifold::DOF];
        to_row_major(state_cov, cov_rm);
        check(ukfb_initialize(engine, 0, 1, mu, cov_rm), "initializeFilter");
        initialised = true;
        last_measurement_time.microseconds = 0;
    }

    /** Provides the current state and covariance (:51-60). @returns false if not initialized. */
    bool getCurrentState(State& state, Covariance& state_cov) const
    {
        if(!initialised)
            return false;
        double mu[Manifold::STORED];
        double cov_rm[Manifold::DOF * Manifold::DOF];
        check(ukfb_get_state(engine, 0, 1, mu, cov_rm, NULL), "getCurrentState");
        from_row_major(cov_rm, state_cov);
        state.fromArray(mu);
        return true;
    }

    /** Provides the current state (:67-75). @returns false if not initialized. */
    bool getCurrentState(State& state) const
    {
        if(!initialised)
            return false;
        double mu[Manifold::STORED];
        check(ukfb_get_state(engine, 0, 1, mu, NULL, NULL), "getCurrentState");
        state.fromArray(mu);
        return true;
    }

    /** Computes the time delta from a sample timestamp and calls predictionStep(delta_t) (:83-100). */
    void predictionStepFromSampleTime(const base::Time& sample_time)
    {
        if(last_measurement_time.isNull())
        {
            last_measurement_time = sample_time;
            return;
        }
        double delta_t = (sample_time - last_measurement_time).toSeconds();
        if(delta_t > min_time_delta)
            last_measurement_time = sample_time;
        predictionStep(delta_t);
    }

    /** Calls predictionStepImpl after checking delta_t (:107-125). */
    void predictionStep(double delta_t)
    {
        if(delta_t < 0.0)
        {
            throw std::runtime_error("Delta time is negative!");
        }
        else if(delta_t <= min_time_delta)
        {
            return;
        }
        else if(delta_t > max_time_delta)
        {
            throw std::runtime_error("Delta time is greater then the allowed maximum!");
        }
        predictionStepImpl(delta_t);
    }

    unsigned getStateSize() const {return unsigned(Manifold::DOF);}
    bool isInitialized() const {return initialised;}
    const Covariance& getProcessNoiseCovariance() const {return process_noise_cov;}
    void setProcessNoiseCovariance(const Covariance& noise_cov)
    {
        process_noise_cov = noise_cov;
        double noise_rm[Manifold::DOF * Manifold::DOF];
        to_row_major(process_noise_cov, noise_rm);
        check(ukfb_set_process_noise(engine, noise_rm), "setProcessNoiseCovariance");
    }
    const base::Time& getLastMeasurementTime() const {return last_measurement_time;}
    void setLastMeasurementTime(const base::Time& last_measurement_time)
                               {this->last_measurement_time = last_measurement_time;}
    double getMaxTimeDelta() const {return max_time_delta;}
    void setMaxTimeDelta(double max_time_delta) {this->max_time_delta = max_time_delta;}
    double getMinTimeDelta() const {return min_time_delta;}
    void setMinTimeDelta(double min_time_delta) {this->min_time_delta = min_time_delta;}

    /** Status word of the last engine call (UKFB_ST_*): the batched counterpart of ukfom's cerr/assert. */
    uint32_t lastEngineStatus() const
    {
        uint32_t st = 0;
        ukfb_get_status(engine, 0, 1, &st);
        return st;
    }

protected:
    virtual void predictionStepImpl(double delta_t) = 0;

    template<int DIM, typename scalar_type>
    void checkMeasurment(const Matrix<scalar_type, DIM, 1>& mu, const Matrix<scalar_type, DIM, DIM>& cov) const
    {
        if(!mu.allFinite() || !cov.allFinite())
            throw std::runtime_error("Measurement or covariance contains non-finite values!");
    }

    /** One engine predict with the host-side gate already passed (the engine re-checks with the same bounds). */
    void enginePredict(double delta_t)
    {
        // the engine's own copy of the two bounds follows the accessors; it is pushed only when one of them changed
        if (min_time_delta != pushed_min_time_delta || max_time_delta != pushed_max_time_delta) {
            engine_cfg.min_time_delta = min_time_delta;
            engine_cfg.max_time_delta = max_time_delta;
            check(ukfb_set_config(engine, &engine_cfg), "set_config");
            pushed_min_time_delta = min_time_delta;
            pushed_max_time_delta = max_time_delta;
        }
        check(ukfb_predict(engine, delta_t), "predict");
        check(ukfb_sync(engine), "sync");
    }

    /** ukf->update(z, h, Q) for measurement model `model` (UKFB_MEAS_*); z, Q padded to 3 / 3x3. */
    template<int DIM>
    void engineUpdate(int model, const Matrix<double, DIM, 1>& mu, const Matrix<double, DIM, DIM>& cov)
    {
        double z[3] = {0, 0, 0}, Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int r = 0; r < DIM; ++r) {
            z[r] = mu(r);
            for (int c = 0; c < DIM; ++c) Q[r * 3 + c] = cov(r, c);
        }
        check(ukfb_update(engine, model, z, Q, NULL), "update");
        check(ukfb_sync(engine), "sync");
    }

    void check(int rc, const char* what) const
    {
        if (rc != UKFB_OK)
            throw std::runtime_error(std::string("pose_estimation engine call failed (") + what + "): " + ukfb_last_error());
    }

private:
    UnscentedKalmanFilter(const UnscentedKalmanFilter&);             // boost::noncopyable (:16)
    UnscentedKalmanFilter& operator=(const UnscentedKalmanFilter&);

protected:
    ukfb_engine* engine;              // replaces boost::shared_ptr<MTK_UKF> ukf (:150)
    ukfb_config engine_cfg;           // the engine's configuration as created (+ the two time bounds last pushed)
    double pushed_min_time_delta, pushed_max_time_delta;
    bool initialised;
    Covariance process_noise_cov;
    base::Time last_measurement_time;
    double max_time_delta;
    double min_time_delta;
};

}

#endif
