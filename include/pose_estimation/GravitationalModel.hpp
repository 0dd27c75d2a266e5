// pose_estimation/GravitationalModel.hpp -- WGS-84 constants and the normal-gravity helper, host mirror of the
// reference's header (src/GravitationalModel.hpp:10-44).  Of these only EARTHW (:16) is on the filter path
// (earth rotation vector of the orientation filter, OrientationUKF.cpp:47); the engine's own copy is
// ukfb_orient_set_params.  WGS_84() is an init-time scalar helper kept so that callers compile unchanged.
#ifndef _POSE_ESTIMATION_GRAVITATIONAL_MODEL_HPP
#define _POSE_ESTIMATION_GRAVITATIONAL_MODEL_HPP

#include <cmath>

namespace pose_estimation
{

// names and values as the callers of the reference expect them (SI units)
static const double EARTHW = 2.0 * 3.14159265358979323846 / 86164.0;   // sidereal rotation rate, rad/s
static const double GWGS0 = 9.7803267714, GWGS1 = 0.00193185138639;     // Somigliana: equatorial gravity, k
static const double ECC = 0.0818191908426;                              // first eccentricity of the ellipsoid
static const double EQUATORIAL_RADIUS = 6378137.0;                      // semi-major axis, m
static const double GRAVITY = 9.79766542, GRAVITY_SI = 9.80665;         // WGS-84 mean / standard gravity, m/s^2

class GravitationalModel
{
public:
    /** Somigliana normal gravity at geodetic `latitude` (rad), reduced to `altitude` (m) by the inverse-square
     *  law about the equatorial radius. */
    static double WGS_84(double latitude, double altitude)
    {
        const double sin2 = std::sin(latitude) * std::sin(latitude);
        const double surface = GWGS0 * (1.0 + GWGS1 * sin2) / std::sqrt(1.0 - ECC * ECC * sin2);
        const double shrink = EQUATORIAL_RADIUS / (EQUATORIAL_RADIUS + altitude);
        return surface * shrink * shrink;
    }
};

}

#endif
