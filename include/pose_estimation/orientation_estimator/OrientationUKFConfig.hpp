// pose_estimation/orientation_estimator/OrientationUKFConfig.hpp -- the configuration PODs of the
// reference (src/orientation_estimator/OrientationUKFConfig.hpp:9-49) on the dependency-free vector type.
#ifndef _POSE_ESTIMATION_ORIENTATION_UKF_CONFIG_HPP
#define _POSE_ESTIMATION_ORIENTATION_UKF_CONFIG_HPP

#include <pose_estimation/Types.hpp>

namespace pose_estimation
{

struct InertialNoiseParameters
{
    Vector3d randomwalk;        /* (m/s^2)/sqrt(Hz) or (rad/s)/sqrt(Hz) */
    Vector3d bias_offset;       /* initial bias value */
    Vector3d bias_instability;  /* m/s^2 or rad/s */
    double bias_tau;            /* seconds */
};

struct LocationConfiguration
{
    double latitude;   /* radians */
    double longitude;  /* radians */
    double altitude;   /* meters */
};

struct OrientationUKFConfig
{
    InertialNoiseParameters acceleration;
    InertialNoiseParameters rotation_rate;
    LocationConfiguration location;
    Vector3d max_velocity;
};

}

#endif
