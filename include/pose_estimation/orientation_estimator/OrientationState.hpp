// pose_estimation/orientation_estimator/OrientationState.hpp -- host mirror of the 13-DOF manifold
// (reference: src/orientation_estimator/OrientationState.hpp:15-26): orientation (SO3), velocity,
// bias_gyro, bias_acc (vect3 each), gravity (vect1), with the reference's type names.  boxplus / boxminus
// here are single-value caller conveniences (pose_estimation/Manifold.hpp); the filter's run in the engine.
#ifndef _ORIENTATION_STATE_HPP_
#define _ORIENTATION_STATE_HPP_

#include <pose_estimation/Types.hpp>
#include <ukf_batch.h>

namespace pose_estimation
{

typedef ukfom::mtkwrap< MTK::SO3<double> > RotationType;
typedef ukfom::mtkwrap<RotationType::vect_type> VelocityType;
typedef ukfom::mtkwrap<RotationType::vect_type> BiasType;
typedef ukfom::mtkwrap< MTK::vect<1> > GravityType;

struct OrientationState
{
    enum { DOF = 13, STORED = 14, ENGINE_MODEL = UKFB_MODEL_ORIENT };
    typedef double scalar;
    typedef Matrix<double, 13, 1> vectorized_type;

    RotationType orientation;  // orientation of IMU in navigation/target frame
    VelocityType velocity;     // velocity of IMU in navigation/target frame
    BiasType bias_gyro;
    BiasType bias_acc;
    GravityType gravity;

    /** tangent offset of a field named by its member pointer (MTK::subblock / MTK::setDiagonal) */
    static int tangentIndex(RotationType OrientationState::*) { return 0; }
    static int tangentIndex(VelocityType OrientationState::*f)
    {
        return f == &OrientationState::velocity ? 3 : (f == &OrientationState::bias_gyro ? 6 : 9);
    }
    static int tangentIndex(GravityType OrientationState::*) { return 12; }

    void boxplus(const vectorized_type& d, scalar scale = 1.0)
    {
        orientation.boxplus(Vector3d(d[0], d[1], d[2]), scale);
        velocity.boxplus(Vector3d(d[3], d[4], d[5]), scale);
        bias_gyro.boxplus(Vector3d(d[6], d[7], d[8]), scale);
        bias_acc.boxplus(Vector3d(d[9], d[10], d[11]), scale);
        gravity(0) += scale * d[12];
    }
    void boxminus(vectorized_type& res, const OrientationState& other) const
    {
        Vector3d t;
        orientation.boxminus(t, other.orientation); res[0] = t[0]; res[1] = t[1]; res[2] = t[2];
        velocity.boxminus(t, other.velocity);       res[3] = t[0]; res[4] = t[1]; res[5] = t[2];
        bias_gyro.boxminus(t, other.bias_gyro);     res[6] = t[0]; res[7] = t[1]; res[8] = t[2];
        bias_acc.boxminus(t, other.bias_acc);       res[9] = t[0]; res[10] = t[1]; res[11] = t[2];
        res[12] = gravity(0) - other.gravity(0);
    }

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
