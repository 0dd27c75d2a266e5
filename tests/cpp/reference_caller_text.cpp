// Source compatibility of the HOST value types with code written against the reference's API.
// Everything in namespace component is CALLER code of the kind a Rock task writes around the filter classes: a
// navigation component that seeds a PoseUKF from a GNSS / compass fix, publishes the estimate, and compares two states.
// It is written for this test (its functions, variable names and structure are its own; it is NOT the body of any
// reference function) and exercises the members such code relies on, with the reference's type names:
//   assignable blocks `P.block(r, c, h, w) = m`, `setZero()`, `RotationType(MTK::SO3<double>(q))`, `q * v`,
//   `MTK::SO3<double>::exp(v)` (src/pose_with_velocity/PoseUKF.cpp:135), `MTK::setDiagonal` / `MTK::subblock`
//   (:104-107, :184-185), `PoseUKF::MTK_UKF::cov`, `WState` with operator+ / operator- and boxplus / boxminus
//   (src/UnscentedKalmanFilter.hpp:19-25), `Eigen::Matrix<double, DIM, 1>`, `Mu::Zero()`, `Cov::Identity()`, `allFinite()`.
// It must compile -Werror against include/ unchanged, whether pose_estimation::Matrix is the dependency-free stand-in (this
// image) or Eigen::Matrix.  `dump` as first argument prints SO(3) exp / log / boxplus / boxminus values for
// tests/test_abi_and_host.py to compare with scipy's Rotation.
#include <pose_estimation/Measurement.hpp>
#include <pose_estimation/orientation_estimator/OrientationState.hpp>
#include <pose_estimation/orientation_estimator/OrientationUKF.hpp>
#include <pose_estimation/orientation_estimator/OrientationUKFNoise.hpp>
#include <pose_estimation/pose_with_velocity/BodyStateMeasurement.hpp>
#include <pose_estimation/pose_with_velocity/PoseUKF.hpp>
#include <pose_estimation/pose_with_velocity/PoseWithVelocity.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>

using namespace pose_estimation;

namespace component
{
struct NavFix {
    Eigen::Vector3d enu;            // m, local tangent plane
    Eigen::Vector3d compass_rpy;    // axis-angle of the compass solution
    Eigen::Matrix3d enu_cov;
    double heading_sigma;
};

typedef PoseUKF::MTK_UKF::cov FilterCov;            // UnscentedKalmanFilter.hpp:24-25
typedef PoseUKF::WState WrappedState;                // :23

// A stationary start: pose from the fix, rates zero, velocity uncertainty isotropic.
void seedFromFix(const NavFix& fix, PoseUKF::State& x0, FilterCov& P0)
{
    x0.orientation = RotationType(MTK::SO3<double>::exp(fix.compass_rpy));
    x0.position = TranslationType(fix.enu);
    x0.velocity = VelocityType(Eigen::Vector3d::Zero());
    x0.angular_velocity = VelocityType(Eigen::Vector3d::Zero());

    P0.setZero();
    P0.block(0, 0, 3, 3) = fix.enu_cov;
    MTK::setDiagonal(P0, &PoseUKF::State::orientation, fix.heading_sigma * fix.heading_sigma);
    MTK::setDiagonal(P0, &PoseUKF::State::velocity, 0.25);
    MTK::setDiagonal(P0, &PoseUKF::State::angular_velocity, 1e-4);
}

// What the component writes to its output port: velocity expressed in the world frame, position block of the covariance.
void publish(const PoseUKF::State& est, const FilterCov& P, base::samples::RigidBodyState& port)
{
    port.orientation = est.orientation;
    port.position = est.position;
    const Eigen::Vector3d v_world = est.orientation * est.velocity;
    port.velocity = v_world;
    port.angular_velocity = est.angular_velocity;
    port.cov_position = MTK::subblock(P, &PoseUKF::State::position);
    port.cov_velocity = P.block(6, 6, 3, 3);
}

// The position noise of a component that models it in the body frame, turned into the navigation frame.
Eigen::Matrix3d bodyNoiseInNavFrame(const PoseUKF::State& est, const FilterCov& body_noise)
{
    const Eigen::Matrix3d rot = est.orientation.matrix();
    FilterCov shaped = body_noise;
    MTK::subblock(shaped, &PoseUKF::State::position) = rot * MTK::subblock(body_noise, &PoseUKF::State::position) * rot.transpose();
    return Eigen::Matrix3d(MTK::subblock(shaped, &PoseUKF::State::position));
}

// A consistency monitor: tangent-space distance between two estimates.
double tangentDistance(const WrappedState& a, const WrappedState& b)
{
    const WrappedState::vectorized_type d = a - b;   // boxminus
    return d.norm();
}

template <int DIM> bool usable(const Eigen::Matrix<double, DIM, 1>& mu, const Eigen::Matrix<double, DIM, DIM>& cov)
{
    return mu.allFinite() && cov.allFinite();
}
}  // namespace component

MEASUREMENT(SonarRange, 2)

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

static int dump()
{
    // deterministic rotation vectors incl. tiny, beyond-pi/2 and near-pi angles
    const double vs[][3] = {{0.3, -0.2, 0.1}, {1e-9, 2e-9, -1e-9}, {0.0, 0.0, 0.0}, {1.2, 0.9, -1.4}, {-2.0, 1.5, 1.0},
                            {0.01, 0.02, 0.03}, {3.0, 0.2, -0.3}, {1e-3, -1e-3, 5e-4}};
    const int n = int(sizeof(vs) / sizeof(vs[0]));
    for (int i = 0; i < n; ++i) {
        const Eigen::Vector3d v(vs[i][0], vs[i][1], vs[i][2]);
        const MTK::SO3<double> q = MTK::SO3<double>::exp(v);
        const Eigen::Vector3d back = MTK::SO3<double>::log(q);
        std::printf("exp %.17g %.17g %.17g -> %.17g %.17g %.17g %.17g log %.17g %.17g %.17g\n", v[0], v[1], v[2], q.x(), q.y(), q.z(), q.w(),
                    back[0], back[1], back[2]);
        const Eigen::Vector3d w(vs[(i + 3) % n][0] * 0.5, vs[(i + 3) % n][1] * 0.5, vs[(i + 3) % n][2] * 0.5);
        MTK::SO3<double> p = q;
        p.boxplus(w, 0.7);
        Eigen::Vector3d diff;
        p.boxminus(diff, q);
        std::printf("boxplus %.17g %.17g %.17g scale 0.7 -> %.17g %.17g %.17g %.17g boxminus %.17g %.17g %.17g\n", w[0], w[1], w[2], p.x(), p.y(),
                    p.z(), p.w(), diff[0], diff[1], diff[2]);
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::strcmp(argv[1], "dump") == 0) return dump();

    component::NavFix fix;
    fix.enu = Eigen::Vector3d(12.0, -7.5, 1.25);
    fix.compass_rpy = Eigen::Vector3d(0.0, 0.0, 0.5);   // 0.5 rad about z
    fix.enu_cov = 0.04 * Eigen::Matrix3d::Identity();
    fix.enu_cov(0, 1) = fix.enu_cov(1, 0) = 0.003;
    fix.heading_sigma = 0.1;

    PoseUKF::State x;
    component::FilterCov P = component::FilterCov::Ones();
    component::seedFromFix(fix, x, P);
    CHECK(P(0, 0) == 0.04 && P(0, 1) == 0.003 && P(1, 0) == 0.003 && std::fabs(P(4, 4) - 0.01) < 1e-16 && P(7, 7) == 0.25 && P(11, 11) == 1e-4);
    CHECK(P(0, 3) == 0.0 && P(5, 6) == 0.0 && P(11, 0) == 0.0 && P(3, 4) == 0.0);
    CHECK(x.position[1] == -7.5 && std::fabs(x.orientation.w() - std::cos(0.25)) < 1e-16 && std::fabs(x.orientation.z() - std::sin(0.25)) < 1e-16);
    CHECK(x.orientation.x() == 0.0 && x.velocity[2] == 0.0);

    // the in-tree adapter (record based: the batched engine's layout) agrees with what the component publishes
    x.velocity = VelocityType(Eigen::Vector3d(1.0, 0.0, 0.0));
    x.angular_velocity = VelocityType(Eigen::Vector3d(0.0, 0.0, 0.1));
    base::samples::RigidBodyState port, adapter;
    component::publish(x, P, port);
    BodyStateMeasurement::toRigidBodyState(x, P, adapter);
    CHECK(std::fabs(port.velocity[0] - std::cos(0.5)) < 1e-15 && std::fabs(port.velocity[1] - std::sin(0.5)) < 1e-15);
    CHECK(adapter.velocity[0] == port.velocity[0] && adapter.velocity[1] == port.velocity[1] && adapter.position[2] == port.position[2]);
    CHECK(port.cov_position(1, 0) == 0.003 && adapter.cov_position(1, 0) == 0.003 && port.cov_velocity(2, 2) == 0.25 && adapter.cov_velocity(2, 2) == 0.25);
    PoseUKF::State x_back;
    component::FilterCov P_back;
    adapter.velocity = x.velocity;   // (the adapter's import takes the sample's velocity as it is)
    BodyStateMeasurement::fromRigidBodyState(adapter, x_back, P_back);
    CHECK(x_back.position[0] == 12.0 && x_back.orientation.z() == x.orientation.z() && P_back(0, 1) == 0.003 && P_back(3, 9) == 0.0);

    // rotation * block * rotation^T through MTK::subblock
    component::FilterCov noise = component::FilterCov::Zero();
    noise.block(0, 0, 3, 3) = fix.enu_cov;
    const Eigen::Matrix3d shaped = component::bodyNoiseInNavFrame(x, noise);
    CHECK(std::fabs(shaped.trace() - fix.enu_cov.trace()) < 1e-15 && std::fabs(shaped(2, 2) - 0.04) < 1e-16);
    const double c = std::cos(0.5), s = std::sin(0.5);
    CHECK(std::fabs(shaped(0, 0) - (0.04 * c * c - 2 * 0.003 * c * s + 0.04 * s * s)) < 1e-15);

    // WState: operator+ is a copy moved along the tangent, operator- the tangent vector between two states
    component::WrappedState a(x);
    component::WrappedState::vectorized_type d = component::WrappedState::vectorized_type::Zero();
    d[0] = 0.5; d[5] = 0.2; d[7] = -0.3; d[11] = 0.01;
    const component::WrappedState b = a + d;
    CHECK(b.position[0] == 12.5 && b.velocity[1] == -0.3 && std::fabs(b.angular_velocity[2] - 0.11) < 1e-16 && a.position[0] == 12.0);
    const component::WrappedState::vectorized_type back = b - a;
    bool round_trip = true;
    for (int k = 0; k < 12; ++k) round_trip = round_trip && std::fabs(back[k] - d[k]) < 1e-15;
    CHECK(round_trip && std::fabs(component::tangentDistance(b, a) - d.norm()) < 1e-15);
    PoseUKF::State moved = x;
    moved.boxplus(d, 2.0);                                   // scale argument as in processModel's boxplus(v, dt)
    CHECK(moved.position[0] == 13.0 && std::fabs(moved.velocity[1] + 0.6) < 1e-16);
    component::WrappedState::vectorized_type twice;
    moved.boxminus(twice, x);
    CHECK(std::fabs(twice[5] - 0.4) < 1e-15 && std::fabs(twice[0] - 1.0) < 1e-15);

    // SO(3): exp / log, the [+] of the process models, the measurement conversion of PoseUKF.cpp:135
    const Eigen::Vector3d rv(0.3, -0.2, 0.1);
    const RotationType r1(MTK::SO3<double>::exp(rv));
    const Eigen::Vector3d rv_back = MTK::SO3<double>::log(r1);
    CHECK(std::fabs(rv_back[0] - 0.3) < 1e-15 && std::fabs(rv_back[1] + 0.2) < 1e-15 && std::fabs(rv_back[2] - 0.1) < 1e-15);
    RotationType r2 = r1;
    r2.boxplus(Eigen::Vector3d(0.0, 0.0, 2.0), 0.05);       // yaw rate 2 rad/s for 50 ms
    const Eigen::Vector3d step = r2 - r1;
    CHECK(std::fabs(step[2] - 0.1) < 1e-15 && std::fabs(step[0]) < 1e-15);
    const Eigen::Quaterniond qi = x.orientation.inverse() * x.orientation;
    CHECK(std::fabs(qi.w() - 1.0) < 1e-15 && std::fabs(qi.z()) < 1e-15);
    const Eigen::Vector3d there_and_back = x.orientation.inverse() * (x.orientation * Eigen::Vector3d(0.3, -0.2, 0.1));
    CHECK(std::fabs(there_and_back[0] - 0.3) < 1e-15 && std::fabs(there_and_back[2] - 0.1) < 1e-15);

    // OrientationState: the same surface, 13 DOF with a one-dimensional gravity field
    OrientationUKF::WState o1, o2;
    o1.gravity(0) = 9.81;
    OrientationUKF::WState::vectorized_type od = OrientationUKF::WState::vectorized_type::Zero();
    od[2] = 0.25; od[4] = 1.5; od[12] = -0.01;
    o2 = o1 + od;
    const OrientationUKF::WState::vectorized_type od_back = o2 - o1;
    CHECK(std::fabs(od_back[2] - 0.25) < 1e-15 && od_back[4] == 1.5 && std::fabs(od_back[12] + 0.01) < 1e-15 && int(OrientationUKF::MTK_UKF::n) == 13);
    OrientationUKF::MTK_UKF::cov oc = OrientationUKF::MTK_UKF::cov::Zero();
    MTK::setDiagonal(oc, &OrientationState::bias_gyro, 1e-6);
    MTK::setDiagonal(oc, &OrientationState::gravity, 1e-8);
    CHECK(oc(6, 6) == 1e-6 && oc(8, 8) == 1e-6 && oc(9, 9) == 0.0 && oc(12, 12) == 1e-8 && oc(5, 5) == 0.0);

    // OrientationUKFConfig -> initial state, initial covariance, process noise (OrientationUKFNoise.hpp; the formulas are this
    // package's, the reference holds none: checked here against their closed forms)
    OrientationUKFConfig cfg;
    cfg.rotation_rate.randomwalk = Eigen::Vector3d(1e-3, 2e-3, 3e-3);
    cfg.rotation_rate.bias_offset = Eigen::Vector3d(1e-4, -2e-4, 0.0);
    cfg.rotation_rate.bias_instability = Eigen::Vector3d(1e-5, 1e-5, 2e-5);
    cfg.rotation_rate.bias_tau = 3600.0;
    cfg.acceleration.randomwalk = Eigen::Vector3d(1e-2, 1e-2, 2e-2);
    cfg.acceleration.bias_offset = Eigen::Vector3d(0.0, 0.01, -0.02);
    cfg.acceleration.bias_instability = Eigen::Vector3d(1e-3, 1e-3, 1e-3);
    cfg.acceleration.bias_tau = 1800.0;
    cfg.location.latitude = 0.92698121;     // the latitude of the reference's own test (test/test_coordinate_projection.cpp:11)
    cfg.location.longitude = 0.15;
    cfg.location.altitude = 12.0;
    cfg.max_velocity = Eigen::Vector3d(3.0, 3.0, 1.5);
    const OrientationUKFNoise::Covariance pn = OrientationUKFNoise::processNoise(cfg, 0.01);
    CHECK(std::fabs(pn(1, 1) - 4e-6 / 0.01) < 1e-18 && std::fabs(pn(5, 5) - 4e-4 / 0.01) < 1e-16 && pn(0, 1) == 0.0 && pn(12, 12) == 0.0);
    CHECK(std::fabs(pn(8, 8) - 2.0 * 4e-10 / (3600.0 * 0.01)) < 1e-24 && std::fabs(pn(9, 9) - 2.0 * 1e-6 / (1800.0 * 0.01)) < 1e-20);
    // one step of dt = T: the reference scales by delta^2 -> orientation variance rw^2 T, bias variance 2 sigma^2 T / tau
    CHECK(std::fabs(0.01 * 0.01 * pn(0, 0) - 1e-6 * 0.01) < 1e-20);
    const OrientationState x_init = OrientationUKFNoise::initialState(cfg);
    CHECK(x_init.orientation.w() == 1.0 && x_init.velocity[0] == 0.0 && x_init.bias_gyro[1] == -2e-4 && x_init.bias_acc[2] == -0.02);
    CHECK(std::fabs(x_init.gravity(0) - GravitationalModel::WGS_84(0.92698121, 12.0)) == 0.0 && x_init.gravity(0) > 9.78 && x_init.gravity(0) < 9.84);
    const OrientationUKFNoise::Covariance p_init = OrientationUKFNoise::initialCovariance(cfg, 0.05);
    CHECK(std::fabs(p_init(2, 2) - 0.0025) < 1e-18 && p_init(3, 3) == 1.0 && p_init(5, 5) == 0.25 && std::fabs(p_init(8, 8) - 4e-10) < 1e-24 && std::fabs(p_init(12, 12) - 1e-4) < 1e-18);
    bool threw = false;
    try { OrientationUKFNoise::processNoise(cfg, 0.0); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);

    // measurement structs and the Eigen spellings of the reference's headers
    SonarRange m;
    CHECK(m.mu[0] == 0.0 && m.cov(0, 0) == 1.0 && m.cov(0, 1) == 0.0);
    CHECK(component::usable<2>(m.mu, m.cov));
    m.cov(1, 0) = std::nan("");
    CHECK(!component::usable<2>(m.mu, m.cov));
    CHECK(ukfom::accept_any_mahalanobis_distance<PoseUKF::WState::scalar>(1e30));

    std::printf(fails ? "reference_caller_text: %d check(s) failed\n" : "reference_caller_text: ok\n", fails);
    return fails ? 1 : 0;
}
