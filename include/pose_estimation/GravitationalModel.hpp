// pose_estimation/GravitationalModel.hpp -- WGS-84 constants and the theoretical gravity helper of the
// reference (src/GravitationalModel.hpp:10-44).  Only EARTHW (:16) is on the filter path
// (OrientationUKF.cpp:47); WGS_84() is an init-time scalar helper kept for drop-in completeness.
#ifndef _POSE_ESTIMATION_GRAVITATIONAL_MODEL_HPP
#define _POSE_ESTIMATION_GRAVITATIONAL_MODEL_HPP

#include <math.h>

namespace pose_estimation
{

static const double EQUATORIAL_RADIUS = 6378137.0;
static const double ECC = 0.0818191908426;
static const double GRAVITY = 9.79766542;
static const double GRAVITY_SI = 9.80665;
static const double GWGS0 = 9.7803267714;
static const double GWGS1 = 0.00193185138639;
static const double EARTHW = ((2.0 * M_PI) / 86164.0);

class GravitationalModel
{
public:
    /** theoretical gravity on the WGS-84 ellipsoid at `latitude` (rad) and `altitude` (m) */
    static double WGS_84(double latitude, double altitude)
    {
        const double s2 = pow(sin(latitude), 2);
        double g = GWGS0 * ((1 + GWGS1 * s2) / sqrt(1 - pow(ECC, 2) * s2));
        g = g * pow(EQUATORIAL_RADIUS / (EQUATORIAL_RADIUS + altitude), 2);
        return g;
    }
};

}

#endif
