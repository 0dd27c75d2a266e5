// pose_estimation/pose_with_velocity/BodyStateMeasurement.hpp -- host mirror of the reference's
// RigidBodyState <-> (PoseWithVelocity, covariance) conversion (src/pose_with_velocity/
// BodyStateMeasurement.hpp:12-41), plus the batched record layout of the engine
// (ukfb_pose_export_body_states / ukfb_pose_import_body_states in include/ukf_batch.h).
#ifndef _POSE_ESTIMATION_BODY_STATE_MEASUREMENT_HPP
#define _POSE_ESTIMATION_BODY_STATE_MEASUREMENT_HPP

#include <base/Time.hpp>
#include <base/samples/RigidBodyState.hpp>
#include "PoseWithVelocity.hpp"

namespace pose_estimation
{

struct BodyStateMeasurement
{
    /** tangent offset of the four 3x3 covariance blocks: position, orientation, velocity, angular velocity */
    static int blockOffset(int field) { return 3 * field; }

    /** one 49-double record of the engine's batched adapters (ukf_batch.h): p(3) q(4, xyzw) v(3) w(3) and the
     *  four 3x3 blocks, row-major */
    static void toRecord(const base::samples::RigidBodyState &rbs, double* rec)
    {
        for (int k = 0; k < 3; ++k) { rec[k] = rbs.position[k]; rec[7 + k] = rbs.velocity[k]; rec[10 + k] = rbs.angular_velocity[k]; }
        for (int k = 0; k < 4; ++k) rec[3 + k] = rbs.orientation.coeffs()[k];
        for (int k = 0; k < 9; ++k) { rec[13 + k] = rbs.cov_position[k]; rec[22 + k] = rbs.cov_orientation[k]; rec[31 + k] = rbs.cov_velocity[k]; rec[40 + k] = rbs.cov_angular_velocity[k]; }
    }
    static void fromRecord(const double* rec, base::samples::RigidBodyState &rbs)
    {
        for (int k = 0; k < 3; ++k) { rbs.position[k] = rec[k]; rbs.velocity[k] = rec[7 + k]; rbs.angular_velocity[k] = rec[10 + k]; }
        for (int k = 0; k < 4; ++k) rbs.orientation.coeffs()[k] = rec[3 + k];
        for (int k = 0; k < 9; ++k) { rbs.cov_position[k] = rec[13 + k]; rbs.cov_orientation[k] = rec[22 + k]; rbs.cov_velocity[k] = rec[31 + k]; rbs.cov_angular_velocity[k] = rec[40 + k]; }
    }

    /** reference :14-26.  Same statements as the device kernel import_body_states_kernel (ukf_batch.hip): the
     *  sample's fields become the state, its four covariances the diagonal blocks, everything else zero. */
    static void fromRigidBodyState(const base::samples::RigidBodyState &sample, PoseWithVelocity &x, PoseWithVelocityCovariance &P)
    {
        double rec[49];
        toRecord(sample, rec);
        for (int k = 0; k < 3; ++k) { x.position[k] = rec[k]; x.velocity[k] = rec[7 + k]; x.angular_velocity[k] = rec[10 + k]; }
        for (int k = 0; k < 4; ++k) x.orientation.coeffs()[k] = rec[3 + k];
        P = PoseWithVelocityCovariance::Zero();
        for (int f = 0; f < 4; ++f)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) P(blockOffset(f) + i, blockOffset(f) + j) = rec[13 + 9 * f + 3 * i + j];
    }

    /** reference :28-39.  The velocity leaves in the navigation frame: rotated by the orientation (:32). */
    static void toRigidBodyState(const PoseWithVelocity &x, const PoseWithVelocityCovariance &P, base::samples::RigidBodyState &sample)
    {
        double rec[49];
        const Vector3d v_nav = x.orientation * x.velocity;
        for (int k = 0; k < 3; ++k) { rec[k] = x.position[k]; rec[7 + k] = v_nav[k]; rec[10 + k] = x.angular_velocity[k]; }
        for (int k = 0; k < 4; ++k) rec[3 + k] = x.orientation.coeffs()[k];
        for (int f = 0; f < 4; ++f)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) rec[13 + 9 * f + 3 * i + j] = P(blockOffset(f) + i, blockOffset(f) + j);
        fromRecord(rec, sample);
    }
};

}

#endif
