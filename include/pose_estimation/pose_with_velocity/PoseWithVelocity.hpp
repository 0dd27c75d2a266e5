// pose_estimation/pose_with_velocity/PoseWithVelocity.hpp -- host mirror of the 12-DOF manifold
// (reference: src/pose_with_velocity/PoseWithVelocity.hpp:14-25): position (vect3), orientation (SO3),
// velocity (vect3), angular_velocity (vect3), in that order.  Values only; boxplus / boxminus live in
// the engine (slam-pose_estimation_amd/csrc/ukf_device.hpp).
#ifndef _POSE_WITH_VELOCITY_HPP_
#define _POSE_WITH_VELOCITY_HPP_

#include <pose_estimation/Types.hpp>
#include <ukf_batch.h>

namespace pose_estimation
{

typedef Quaterniond RotationType;
typedef Vector3d TranslationType;
typedef Vector3d VelocityType;

struct PoseWithVelocity
{
    enum { DOF = 12, STORED = 13, ENGINE_MODEL = UKFB_MODEL_POSE };
    typedef double scalar;

    TranslationType position;
    RotationType orientation;
    VelocityType velocity;
    VelocityType angular_velocity;

    // engine layout: p(3) q(x,y,z,w) v(3) w(3)   (include/ukf_batch.h)
    void toArray(double* a) const
    {
        for (int k = 0; k < 3; ++k) { a[k] = position[k]; a[7 + k] = velocity[k]; a[10 + k] = angular_velocity[k]; }
        for (int k = 0; k < 4; ++k) a[3 + k] = orientation.coeffs()[k];
    }
    void fromArray(const double* a)
    {
        for (int k = 0; k < 3; ++k) { position[k] = a[k]; velocity[k] = a[7 + k]; angular_velocity[k] = a[10 + k]; }
        for (int k = 0; k < 4; ++k) orientation.coeffs()[k] = a[3 + k];
    }
};

typedef Matrix<PoseWithVelocity::scalar, PoseWithVelocity::DOF, PoseWithVelocity::DOF> PoseWithVelocityCovariance;

}

#endif
