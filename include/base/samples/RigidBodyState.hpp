// base/samples/RigidBodyState.hpp -- minimal stand-in for Rock's base::samples::RigidBodyState
// (base-types is not available in this image): only the fields BodyStateMeasurement.hpp converts
// (reference: src/pose_with_velocity/BodyStateMeasurement.hpp:16-38).  Remove from -I when base-types is present.
#ifndef UKFB_BASE_SAMPLES_RIGIDBODYSTATE_SHIM_HPP
#define UKFB_BASE_SAMPLES_RIGIDBODYSTATE_SHIM_HPP

#include <base/Time.hpp>
#include <pose_estimation/Types.hpp>

namespace base { namespace samples {

struct RigidBodyState {
    base::Time time;
    pose_estimation::Vector3d position;
    pose_estimation::Quaterniond orientation;
    pose_estimation::Vector3d velocity;
    pose_estimation::Vector3d angular_velocity;
    pose_estimation::Matrix3d cov_position;
    pose_estimation::Matrix3d cov_orientation;
    pose_estimation::Matrix3d cov_velocity;
    pose_estimation::Matrix3d cov_angular_velocity;
};

} }  // namespace base::samples

#endif
