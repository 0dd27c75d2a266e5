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
    /** :14-26 -- fields copied as they are; block-diagonal covariance at (0,0) (3,3) (6,6) (9,9). */
    static void fromRigidBodyState(const base::samples::RigidBodyState &body_state, PoseWithVelocity &filter_state, PoseWithVelocityCovariance &filter_state_cov)
    {
        filter_state.position = body_state.position;
        filter_state.orientation = body_state.orientation;
        filter_state.velocity = body_state.velocity;
        filter_state.angular_velocity = body_state.angular_velocity;

        filter_state_cov = PoseWithVelocityCovariance::Zero();
        filter_state_cov.setBlock<3, 3>(0, 0, body_state.cov_position);
        filter_state_cov.setBlock<3, 3>(3, 3, body_state.cov_orientation);
        filter_state_cov.setBlock<3, 3>(6, 6, body_state.cov_velocity);
        filter_state_cov.setBlock<3, 3>(9, 9, body_state.cov_angular_velocity);
    }

    /** :28-39 -- velocity is rotated into the navigation frame (:32). */
    static void toRigidBodyState(const PoseWithVelocity &filter_state, const PoseWithVelocityCovariance &filter_state_cov, base::samples::RigidBodyState &body_state)
    {
        body_state.position = filter_state.position;
        body_state.orientation = filter_state.orientation;
        body_state.velocity = body_state.orientation * filter_state.velocity;
        body_state.angular_velocity = filter_state.angular_velocity;

        body_state.cov_position = filter_state_cov.block<3, 3>(0, 0);
        body_state.cov_orientation = filter_state_cov.block<3, 3>(3, 3);
        body_state.cov_velocity = filter_state_cov.block<3, 3>(6, 6);
        body_state.cov_angular_velocity = filter_state_cov.block<3, 3>(9, 9);
    }

    /** one 49-double record of the engine's batched adapters <-> RigidBodyState */
    static void toRecord(const base::samples::RigidBodyState &b, double* r)
    {
        for (int k = 0; k < 3; ++k) { r[k] = b.position[k]; r[7 + k] = b.velocity[k]; r[10 + k] = b.angular_velocity[k]; }
        for (int k = 0; k < 4; ++k) r[3 + k] = b.orientation.coeffs()[k];
        for (int k = 0; k < 9; ++k) { r[13 + k] = b.cov_position[k]; r[22 + k] = b.cov_orientation[k]; r[31 + k] = b.cov_velocity[k]; r[40 + k] = b.cov_angular_velocity[k]; }
    }
    static void fromRecord(const double* r, base::samples::RigidBodyState &b)
    {
        for (int k = 0; k < 3; ++k) { b.position[k] = r[k]; b.velocity[k] = r[7 + k]; b.angular_velocity[k] = r[10 + k]; }
        for (int k = 0; k < 4; ++k) b.orientation.coeffs()[k] = r[3 + k];
        for (int k = 0; k < 9; ++k) { b.cov_position[k] = r[13 + k]; b.cov_orientation[k] = r[22 + k]; b.cov_velocity[k] = r[31 + k]; b.cov_angular_velocity[k] = r[40 + k]; }
    }
};

}

#endif
