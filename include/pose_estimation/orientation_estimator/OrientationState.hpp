// pose_estimation/orientation_estimator/OrientationState.hpp -- host mirror of the 13-DOF manifold
// (reference: src/orientation_estimator/OrientationState.hpp:15-26): orientation (SO3), velocity,
// bias_gyro, bias_acc (vect3 each), gravity (vect1).
#ifndef _ORIENTATION_STATE_HPP_
#define _ORIENTATION_STATE_HPP_

#include <pose_estimation/Types.hpp>
#include <ukf_batch.h>

namespace pose_estimation
{

typedef Matrix<double, 1, 1> GravityType;
typedef Vector3d BiasType;

struct OrientationState
{
    enum { DOF = 13, STORED = 14, ENGINE_MODEL = UKFB_MODEL_ORIENT };
    typedef double scalar;

    Quaterniond orientation;   // orientation of IMU in navigation/target frame
    Vector3d velocity;         // velocity of IMU in navigation/target frame
    BiasType bias_gyro;
    BiasType bias_acc;
    GravityType gravity;

    // engine layout: q(x,y,z,w) v(3) bg(3) ba(3) g   (include/ukf_batch.h)
    void toArray(double* a) const
    {
        for (int k = 0; k < 4; ++k) a[k] = orientation.coeffs()[k];
        for (int k = 0; k < 3; ++k) { a[4 + k] = velocity[k]; a[7 + k] = bias_gyro[k]; a[10 + k] = bias_acc[k]; }
        a[13] = gravity(0);
    }
    void fromArray(const double* a)
    {
        for (int k = 0; k < 4; ++k) orientation.coeffs()[k] = a[k];
        for (int k = 0; k < 3; ++k) { velocity[k] = a[4 + k]; bias_gyro[k] = a[7 + k]; bias_acc[k] = a[10 + k]; }
        gravity(0) = a[13];
    }
};

}

#endif
