// pose_estimation/orientation_estimator/OrientationUKF.hpp -- host mirror of
// pose_estimation::OrientationUKF (reference: src/orientation_estimator/OrientationUKF.hpp:20-62,
// OrientationUKF.cpp:41-89) over the MI355X engine.
#ifndef _POSE_ESTIMATION_ORIENTATION_UKF_HPP
#define _POSE_ESTIMATION_ORIENTATION_UKF_HPP

#include "OrientationState.hpp"
#include "OrientationUKFConfig.hpp"
#include <pose_estimation/GravitationalModel.hpp>
#include <pose_estimation/Measurement.hpp>
#include <pose_estimation/UnscentedKalmanFilter.hpp>

namespace pose_estimation
{

class OrientationUKF : public UnscentedKalmanFilter<OrientationState>
{
public:
    MEASUREMENT(RotationRate, 3)
    MEASUREMENT(Acceleration, 3)
    MEASUREMENT(VelocityMeasurement, 3)

public:
    /** OrientationUKF.cpp:41-51: taus, earth rotation from the latitude, input latches. */
    OrientationUKF(const State& initial_state, const Covariance& state_cov,
                   double gyro_bias_tau, double acc_bias_tau, const LocationConfiguration& location) :
                        gyro_bias_tau(gyro_bias_tau), acc_bias_tau(acc_bias_tau)
    {
        initializeFilter(initial_state, state_cov);
        earth_rotation[0] = EARTHW * cos(location.latitude);
        earth_rotation[1] = 0.;
        earth_rotation[2] = EARTHW * sin(location.latitude);
        check(ukfb_orient_set_params(engine, gyro_bias_tau, acc_bias_tau, earth_rotation.data()), "params");
        rotation_rate.mu = RotationRate::Mu::Zero();
        acceleration.mu[0] = 0.; acceleration.mu[1] = 0.; acceleration.mu[2] = initial_state.gravity(0);
        check(ukfb_orient_set_inputs(engine, 0, 1, rotation_rate.mu.data(), acceleration.mu.data()), "inputs");
    }
    virtual ~OrientationUKF() {}

    /** Sets the current rotation rate of the IMU in rad/s (OrientationUKF.cpp:53-57). */
    void integrateMeasurement(const RotationRate& measurement)
    {
        checkMeasurment(measurement.mu, measurement.cov);
        rotation_rate = measurement;
        check(ukfb_orient_set_inputs(engine, 0, 1, rotation_rate.mu.data(), NULL), "rotation rate");
    }

    /** Sets the current acceleration of the IMU in m/s^2 (OrientationUKF.cpp:59-63). */
    void integrateMeasurement(const Acceleration& measurement)
    {
        checkMeasurment(measurement.mu, measurement.cov);
        acceleration = measurement;
        check(ukfb_orient_set_inputs(engine, 0, 1, NULL, acceleration.mu.data()), "acceleration");
    }

    /** Integrates the linear velocity of the IMU in m/s (OrientationUKF.cpp:65-72). */
    void integrateMeasurement(const VelocityMeasurement& measurement)
    {
        checkMeasurment(measurement.mu, measurement.cov);
        engineUpdate<3>(UKFB_MEAS_ORIENT_BODYVEL3, measurement.mu, measurement.cov);
    }

    /** Unbiased rotation rate in the IMU frame (OrientationUKF.cpp:74-77). */
    RotationRate::Mu getRotationRate()
    {
        RotationRate::Mu out;
        check(ukfb_orient_get_rotation_rate(engine, 0, 1, out.data()), "rotation rate read-out");
        return out;
    }

protected:
    /** OrientationUKF.cpp:79-89: noise rotation, delta^2 scaling and ukf->predict run on the device. */
    void predictionStepImpl(double delta) { enginePredict(delta); }

protected:
    RotationRate rotation_rate;
    Acceleration acceleration;
    Vector3d earth_rotation;
    double gyro_bias_tau;
    double acc_bias_tau;
};

}

#endif
