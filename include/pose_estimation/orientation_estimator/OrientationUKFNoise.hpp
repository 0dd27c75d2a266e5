// pose_estimation/orientation_estimator/OrientationUKFNoise.hpp -- host utilities that turn an OrientationUKFConfig
// (reference PODs: src/orientation_estimator/OrientationUKFConfig.hpp:9-49) into what an OrientationUKF is constructed and
// configured with: initial state, initial covariance and the 13 x 13 process noise.
//
// PARITY UNPINNED, by necessity: the reference declares the configuration structs but holds NO code that consumes them (the
// Rock component that does lives outside the repository), so there is no formula to match.  The ones below are the textbook
// conversions, chosen to be consistent with how the reference USES the noise: predictionStepImpl multiplies the whole matrix
// by delta^2 (src/orientation_estimator/OrientationUKF.cpp:79-89), i.e. its entries are variances of RATES held over one step.
//
//   tangent layout (OrientationState.hpp:20-26): orientation 0-2, velocity 3-5, bias_gyro 6-8, bias_acc 9-11, gravity 12
//   T = the IMU sampling period the filter is stepped with (seconds)
//
//   orientation block   diag(rotation_rate.randomwalk_i^2 / T)      white rate noise of density rw [(rad/s)/sqrt(Hz)] sampled at 1/T:
//                                                                     variance rw^2 / T; times delta^2 = T^2 it is rw^2 T per step
//   velocity block      diag(acceleration.randomwalk_i^2 / T)       the same for the accelerometers [(m/s^2)/sqrt(Hz)]
//   bias blocks         diag(2 bias_instability_i^2 / (bias_tau T)) first-order Gauss-Markov bias, stationary sigma = bias_instability,
//                                                                     time constant bias_tau (the process model decays the bias by
//                                                                     T / tau per step, OrientationUKF.cpp:27-30): driving variance per
//                                                                     step sigma^2 (1 - exp(-2 T / tau)) ~ 2 sigma^2 T / tau, over T^2
//   gravity             0                                            a constant of the site
//
//   initial state       orientation = identity, velocity = 0, bias_gyro / bias_acc = the configured bias_offset,
//                       gravity = GravitationalModel::WGS_84(latitude, altitude)     (GravitationalModel.hpp:33-44)
//   initial covariance  orientation: orientation_sigma^2 (argument; the config has no such entry), velocity: (max_velocity_i / 3)^2
//                       (max_velocity read as a 3-sigma bound), biases: bias_instability_i^2, gravity: gravity_sigma^2 (argument)
#ifndef _POSE_ESTIMATION_ORIENTATION_UKF_NOISE_HPP
#define _POSE_ESTIMATION_ORIENTATION_UKF_NOISE_HPP

#include "OrientationState.hpp"
#include "OrientationUKFConfig.hpp"
#include <pose_estimation/GravitationalModel.hpp>

#include <stdexcept>

namespace pose_estimation
{

struct OrientationUKFNoise
{
    typedef Matrix<double, 13, 13> Covariance;

    /** 13 x 13 process noise for setProcessNoiseCovariance(); imu_period = seconds between prediction steps (> 0) */
    static Covariance processNoise(const OrientationUKFConfig& config, double imu_period)
    {
        if (!(imu_period > 0.0)) throw std::invalid_argument("OrientationUKFNoise::processNoise: imu_period must be positive");
        if (!(config.rotation_rate.bias_tau > 0.0) || !(config.acceleration.bias_tau > 0.0))
            throw std::invalid_argument("OrientationUKFNoise::processNoise: bias_tau must be positive");
        Covariance noise = Covariance::Zero();
        for (int k = 0; k < 3; ++k) {
            const double rw_g = config.rotation_rate.randomwalk[k], rw_a = config.acceleration.randomwalk[k];
            const double bi_g = config.rotation_rate.bias_instability[k], bi_a = config.acceleration.bias_instability[k];
            noise(k, k) = rw_g * rw_g / imu_period;
            noise(3 + k, 3 + k) = rw_a * rw_a / imu_period;
            noise(6 + k, 6 + k) = 2.0 * bi_g * bi_g / (config.rotation_rate.bias_tau * imu_period);
            noise(9 + k, 9 + k) = 2.0 * bi_a * bi_a / (config.acceleration.bias_tau * imu_period);
        }
        return noise;
    }

    /** state an OrientationUKF starts from at rest: level, biases at their configured offsets, model gravity of the site */
    static OrientationState initialState(const OrientationUKFConfig& config)
    {
        OrientationState x;
        x.orientation = RotationType(Quaterniond::Identity());
        x.velocity = VelocityType(Vector3d::Zero());
        x.bias_gyro = BiasType(config.rotation_rate.bias_offset);
        x.bias_acc = BiasType(config.acceleration.bias_offset);
        x.gravity(0) = GravitationalModel::WGS_84(config.location.latitude, config.location.altitude);
        return x;
    }

    /** diagonal initial covariance; orientation_sigma in rad, gravity_sigma in m/s^2 */
    static Covariance initialCovariance(const OrientationUKFConfig& config, double orientation_sigma, double gravity_sigma = 1.0e-2)
    {
        Covariance cov = Covariance::Zero();
        for (int k = 0; k < 3; ++k) {
            const double sv = config.max_velocity[k] / 3.0;
            cov(k, k) = orientation_sigma * orientation_sigma;
            cov(3 + k, 3 + k) = sv * sv;
            cov(6 + k, 6 + k) = config.rotation_rate.bias_instability[k] * config.rotation_rate.bias_instability[k];
            cov(9 + k, 9 + k) = config.acceleration.bias_instability[k] * config.acceleration.bias_instability[k];
        }
        cov(12, 12) = gravity_sigma * gravity_sigma;
        return cov;
    }
};

}

#endif
