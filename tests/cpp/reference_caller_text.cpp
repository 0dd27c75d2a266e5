// Source compatibility of the HOST value types with code written against the reference API.
// The statements inside caller::fromRigidBodyState / caller::toRigidBodyState are what a Rock component (or the
// reference's own BodyStateMeasurement helper, src/pose_with_velocity/BodyStateMeasurement.hpp:14-39) writes against
// Eigen / MTK types: assignable blocks `cov.block(r, c, h, w) = m`, `setZero()`, `RotationType(MTK::SO3<double>(q))`,
// `q * v`.  They were typed here by hand as CALLER code; they must compile against include/ unchanged, whether
// pose_estimation::Matrix is the dependency-free stand-in (this image) or Eigen::Matrix (an image with Eigen).
// Also exercised: the Eigen spellings the reference's public headers use (Eigen::Matrix<double, DIM, 1>,
// Mu::Zero(), Cov::Identity(), allFinite()), transpose / products / quaternion algebra.
#include <pose_estimation/Measurement.hpp>
#include <pose_estimation/pose_with_velocity/BodyStateMeasurement.hpp>
#include <pose_estimation/pose_with_velocity/PoseWithVelocity.hpp>

#include <cmath>
#include <cstdio>

using namespace pose_estimation;

namespace caller
{
void fromRigidBodyState(const base::samples::RigidBodyState& body_state, PoseWithVelocity& filter_state, PoseWithVelocityCovariance& filter_state_cov)
{
    filter_state.position = TranslationType(body_state.position);
    filter_state.orientation = RotationType(MTK::SO3<double>(body_state.orientation));
    filter_state.velocity = VelocityType(body_state.velocity);
    filter_state.angular_velocity = VelocityType(body_state.angular_velocity);

    filter_state_cov.setZero();
    filter_state_cov.block(0, 0, 3, 3) = body_state.cov_position;
    filter_state_cov.block(3, 3, 3, 3) = body_state.cov_orientation;
    filter_state_cov.block(6, 6, 3, 3) = body_state.cov_velocity;
    filter_state_cov.block(9, 9, 3, 3) = body_state.cov_angular_velocity;
}

void toRigidBodyState(const PoseWithVelocity& filter_state, const PoseWithVelocityCovariance& filter_state_cov, base::samples::RigidBodyState& body_state)
{
    body_state.position = filter_state.position;
    body_state.orientation = filter_state.orientation;
    body_state.velocity = body_state.orientation * filter_state.velocity;
    body_state.angular_velocity = filter_state.angular_velocity;

    body_state.cov_position = filter_state_cov.block(0, 0, 3, 3);
    body_state.cov_orientation = filter_state_cov.block(3, 3, 3, 3);
    body_state.cov_velocity = filter_state_cov.block(6, 6, 3, 3);
    body_state.cov_angular_velocity = filter_state_cov.block(9, 9, 3, 3);
}

// UnscentedKalmanFilter.hpp:142-147, as a free function
template <int DIM> bool finite(const Eigen::Matrix<double, DIM, 1>& mu, const Eigen::Matrix<double, DIM, DIM>& cov)
{
    return mu.allFinite() && cov.allFinite();
}
}  // namespace caller

MEASUREMENT(TestMeasurement, 2)

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

int main()
{
    base::samples::RigidBodyState rbs;
    rbs.position = Eigen::Vector3d(1.0, 2.0, 3.0);
    rbs.orientation = Eigen::Quaterniond(std::cos(0.25), 0.0, 0.0, std::sin(0.25));   // 0.5 rad about z
    rbs.velocity = Eigen::Vector3d(1.0, 0.0, 0.0);
    rbs.angular_velocity = Eigen::Vector3d(0.0, 0.0, 0.1);
    rbs.cov_position = 0.04 * Eigen::Matrix3d::Identity();
    rbs.cov_orientation = 0.01 * Eigen::Matrix3d::Identity();
    rbs.cov_velocity = 0.09 * Eigen::Matrix3d::Identity();
    rbs.cov_angular_velocity = 0.0025 * Eigen::Matrix3d::Identity();
    rbs.cov_position(0, 1) = rbs.cov_position(1, 0) = 0.003;

    PoseWithVelocity x;
    PoseWithVelocityCovariance P = PoseWithVelocityCovariance::Ones();
    caller::fromRigidBodyState(rbs, x, P);
    CHECK(P(0, 0) == 0.04 && P(0, 1) == 0.003 && P(4, 4) == 0.01 && P(7, 7) == 0.09 && P(11, 11) == 0.0025);
    CHECK(P(0, 3) == 0.0 && P(5, 6) == 0.0 && P(11, 0) == 0.0);
    CHECK(x.position[1] == 2.0 && x.orientation.w() == std::cos(0.25) && x.angular_velocity[2] == 0.1);

    // the in-tree helper (record based, the batched engine's layout) agrees with the caller-style statements
    PoseWithVelocity x2;
    PoseWithVelocityCovariance P2;
    BodyStateMeasurement::fromRigidBodyState(rbs, x2, P2);
    bool same = true;
    for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) same = same && P(i, j) == P2(i, j);
    CHECK(same && x2.velocity[0] == x.velocity[0]);

    base::samples::RigidBodyState out, out2;
    caller::toRigidBodyState(x, P, out);
    BodyStateMeasurement::toRigidBodyState(x, P, out2);
    CHECK(std::fabs(out.velocity[0] - std::cos(0.5)) < 1e-15 && std::fabs(out.velocity[1] - std::sin(0.5)) < 1e-15);   // :32
    CHECK(out.cov_velocity(2, 2) == 0.09 && out.cov_position(1, 0) == 0.003 && out.cov_angular_velocity(0, 0) == 0.0025);
    CHECK(out2.velocity[1] == out.velocity[1] && out2.cov_orientation(1, 1) == out.cov_orientation(1, 1));

    // measurement structs and the Eigen spellings of the reference's headers
    TestMeasurement m;
    CHECK(m.mu[0] == 0.0 && m.cov(0, 0) == 1.0 && m.cov(0, 1) == 0.0);
    CHECK(caller::finite<2>(m.mu, m.cov));
    m.mu[1] = std::nan("");
    CHECK(!caller::finite<2>(m.mu, m.cov));

    // rotation * block * rotation^T, the noise shaping of PoseUKF.cpp:184-185, written as a caller would
    Eigen::Matrix3d rot = x.orientation.toRotationMatrix();
    Eigen::Matrix3d blk = P.block(0, 0, 3, 3);
    Eigen::Matrix3d shaped = rot * blk * rot.transpose();
    CHECK(std::fabs(shaped.trace() - blk.trace()) < 1e-15);
    P.block(0, 0, 3, 3) = shaped;
    CHECK(std::fabs(P(0, 0) - shaped(0, 0)) == 0.0);
    Eigen::Quaterniond qi = x.orientation.inverse() * x.orientation;
    CHECK(std::fabs(qi.w() - 1.0) < 1e-15 && std::fabs(qi.z()) < 1e-15);
    Eigen::Vector3d back = x.orientation.inverse() * (x.orientation * Eigen::Vector3d(0.3, -0.2, 0.1));
    CHECK(std::fabs(back[0] - 0.3) < 1e-15 && std::fabs(back[2] - 0.1) < 1e-15);

    std::printf(fails ? "reference_caller_text: %d check(s) failed\n" : "reference_caller_text: ok\n", fails);
    return fails ? 1 : 0;
}
