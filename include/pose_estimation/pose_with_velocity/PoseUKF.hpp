// pose_estimation/pose_with_velocity/PoseUKF.hpp -- host mirror of pose_estimation::PoseUKF
// (reference: src/pose_with_velocity/PoseUKF.hpp:20-96, PoseUKF.cpp:99-196): same nested MEASUREMENT
// types, constructor, the ten integrateMeasurement overloads and predictionStepImpl, each forwarding
// to the MI355X engine call that replaces `ukf->update(...)` / `ukf->predict(...)`.
#ifndef _POSE_ESTIMATION_POSE_UKF_HPP
#define _POSE_ESTIMATION_POSE_UKF_HPP

#include "PoseWithVelocity.hpp"
#include <pose_estimation/Measurement.hpp>
#include <pose_estimation/UnscentedKalmanFilter.hpp>

#include <limits>

namespace pose_estimation
{

class PoseUKF : public UnscentedKalmanFilter<PoseWithVelocity>
{
public:
    MEASUREMENT(PositionMeasurement, 3)
    MEASUREMENT(XYMeasurement, 2)
    MEASUREMENT(ZMeasurement, 1)
    MEASUREMENT(OrientationMeasurement, 3)
    MEASUREMENT(VelocityMeasurement, 3)
    MEASUREMENT(XYVelocityMeasurement, 2)
    MEASUREMENT(ZVelocityMeasurement, 1)
    MEASUREMENT(XVelYawVelMeasurement, 2)
    MEASUREMENT(AngularVelocityMeasurement, 3)
    MEASUREMENT(AccelerationMeasurement, 3)

public:
    /** PoseUKF.cpp:99-110: default process noise diag(0.01, 0.001, 1e-5, 1e-5), acceleration.mu = NaN. */
    PoseUKF(const State& initial_state, const Covariance& state_cov) : UnscentedKalmanFilter<PoseWithVelocity>()
    {
        initializeFilter(initial_state, state_cov);
        Covariance noise = Covariance::Zero();
        for (int k = 0; k < 3; ++k) {
            noise(k, k) = 0.01;            // position
            noise(3 + k, 3 + k) = 0.001;   // orientation
            noise(6 + k, 6 + k) = 0.00001; // velocity
            noise(9 + k, 9 + k) = 0.00001; // angular_velocity
        }
        setProcessNoiseCovariance(noise);
        acceleration.mu = std::numeric_limits<double>::quiet_NaN() * AccelerationMeasurement::Mu::Ones();
        latchAcceleration();
    }
    virtual ~PoseUKF() {}

    /** 3D position, body in navigation frame in m (PoseUKF.cpp:112-117). */
    void integrateMeasurement(const PositionMeasurement& m) { engineUpdate<3>(UKFB_MEAS_POS3, m.mu, m.cov); }
    /** XY position (PoseUKF.cpp:119-124). */
    void integrateMeasurement(const XYMeasurement& m) { engineUpdate<2>(UKFB_MEAS_POS_XY, m.mu, m.cov); }
    /** Z position (PoseUKF.cpp:126-131). */
    void integrateMeasurement(const ZMeasurement& m) { engineUpdate<1>(UKFB_MEAS_POS_Z, m.mu, m.cov); }
    /** 3D orientation as axis-angle (PoseUKF.cpp:133-138); the SO3::exp conversion happens on the device. */
    void integrateMeasurement(const OrientationMeasurement& m) { engineUpdate<3>(UKFB_MEAS_ORIENT_SO3, m.mu, m.cov); }
    /** 3D linear velocity (PoseUKF.cpp:140-145). */
    void integrateMeasurement(const VelocityMeasurement& m) { engineUpdate<3>(UKFB_MEAS_VEL3, m.mu, m.cov); }
    /** XY velocity (PoseUKF.cpp:147-152). */
    void integrateMeasurement(const XYVelocityMeasurement& m) { engineUpdate<2>(UKFB_MEAS_VEL_XY, m.mu, m.cov); }
    /** Z velocity (PoseUKF.cpp:154-159). */
    void integrateMeasurement(const ZVelocityMeasurement& m) { engineUpdate<1>(UKFB_MEAS_VEL_Z, m.mu, m.cov); }
    /** X velocity and yaw rate (PoseUKF.cpp:161-166). */
    void integrateMeasurement(const XVelYawVelMeasurement& m) { engineUpdate<2>(UKFB_MEAS_XVEL_YAWVEL, m.mu, m.cov); }
    /** 3D rotation rates (PoseUKF.cpp:168-173). */
    void integrateMeasurement(const AngularVelocityMeasurement& m) { engineUpdate<3>(UKFB_MEAS_ANGVEL3, m.mu, m.cov); }
    /** Latches the current acceleration; propagated by the next predictions (PoseUKF.cpp:175-178). */
    void integrateMeasurement(const AccelerationMeasurement& m)
    {
        acceleration = m;
        latchAcceleration();
    }

protected:
    /** PoseUKF.cpp:180-196: noise shaping (both branches, quirk included) and ukf->predict run on the device. */
    virtual void predictionStepImpl(const double delta) { enginePredict(delta); }

    void latchAcceleration()
    {
        double mu3[3], cov_rm[9];
        to_row_major(acceleration.mu, mu3);
        to_row_major(acceleration.cov, cov_rm);
        check(ukfb_pose_set_acceleration(engine, 0, 1, mu3, cov_rm), "acceleration");
    }

protected:
    AccelerationMeasurement acceleration;
};

}

#endif
