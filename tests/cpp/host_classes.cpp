// Drives the host C++ mirror (include/pose_estimation/...) the way a Rock component drives the reference
// classes and prints the resulting state as one JSON object; tests/test_gpu_host_cpp.py compares it
// with the CPU oracle.  Inputs are fixed constants so both sides can restate them.
#include <pose_estimation/orientation_estimator/OrientationUKF.hpp>
#include <pose_estimation/pose_with_velocity/PoseUKF.hpp>
#include <pose_estimation/Batch.hpp>
#include <pose_estimation/GravitationalModel.hpp>

#include <cmath>
#include <limits>

#include <cstdio>

using namespace pose_estimation;

static void print_state(const char* name, const double* mu, int S, const double* cov, int D, bool last)
{
    std::printf("\"%s\": {\"mu\": [", name);
    for (int i = 0; i < S; ++i) std::printf("%s%.17g", i ? ", " : "", mu[i]);
    std::printf("], \"cov\": [");
    for (int i = 0; i < D * D; ++i) std::printf("%s%.17g", i ? ", " : "", cov[i]);
    std::printf("]}%s\n", last ? "" : ",");
}

int main()
{
    // ---------------- PoseUKF
    PoseWithVelocity x0;
    x0.position[0] = 1.0; x0.position[1] = -2.0; x0.position[2] = 0.5;
    x0.orientation = Quaterniond(0.9238795325112867, 0.0, 0.3826834323650898, 0.0);   // 45 deg about y
    x0.velocity[0] = 0.3; x0.velocity[1] = 0.1; x0.velocity[2] = -0.2;
    x0.angular_velocity[0] = 0.05; x0.angular_velocity[1] = -0.02; x0.angular_velocity[2] = 0.1;
    PoseUKF::Covariance P0 = PoseUKF::Covariance::Zero();
    for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) P0(i, j) = (i == j ? 0.04 : 0.0) + 0.001 / (1.0 + i + j);
    PoseUKF f(x0, P0);
    bool threw_negative = false;
    f.predictionStepFromSampleTime(base::Time::fromMicroseconds(1000000));   // first call: latch only
    f.predictionStepFromSampleTime(base::Time::fromMicroseconds(1020000));   // dt = 0.02, constant velocity
    PoseUKF::AccelerationMeasurement acc;
    acc.mu[0] = 0.2; acc.mu[1] = -0.1; acc.mu[2] = 0.05;
    acc.cov = 0.01 * PoseUKF::AccelerationMeasurement::Cov::Identity();
    f.integrateMeasurement(acc);
    f.predictionStepFromSampleTime(base::Time::fromMicroseconds(1030000));   // dt = 0.01, acceleration branch
    try { f.predictionStepFromSampleTime(base::Time::fromMicroseconds(1000000)); } catch (const std::runtime_error&) { threw_negative = true; }
    PoseUKF::PositionMeasurement zp;
    zp.mu[0] = 1.02; zp.mu[1] = -1.97; zp.mu[2] = 0.49;
    zp.cov = 0.0025 * PoseUKF::PositionMeasurement::Cov::Identity();
    f.integrateMeasurement(zp);
    PoseUKF::XVelYawVelMeasurement zv;
    zv.mu[0] = 0.31; zv.mu[1] = 0.09;
    zv.cov = 0.01 * PoseUKF::XVelYawVelMeasurement::Cov::Identity();
    f.integrateMeasurement(zv);
    PoseUKF::OrientationMeasurement zo;
    zo.mu[0] = 0.01; zo.mu[1] = 0.79; zo.mu[2] = -0.02;
    zo.cov = 0.001 * PoseUKF::OrientationMeasurement::Cov::Identity();
    f.integrateMeasurement(zo);
    PoseWithVelocity x; PoseUKF::Covariance P;
    bool ok = f.getCurrentState(x, P);
    double mu[13]; x.toArray(mu);
    std::printf("{\"pose_ok\": %s, \"threw_negative\": %s, \"state_size\": %u,\n", ok ? "true" : "false", threw_negative ? "true" : "false", f.getStateSize());
    print_state("pose", mu, 13, P.data(), 12, false);

    // ---------------- OrientationUKF
    OrientationState o0;
    o0.orientation = Quaterniond(0.9914448613738104, 0.0, 0.0, 0.13052619222005157);    // 15 deg yaw
    o0.velocity[0] = 0.1; o0.velocity[1] = 0.0; o0.velocity[2] = -0.05;
    o0.bias_gyro[0] = 1e-4; o0.bias_gyro[1] = -2e-4; o0.bias_gyro[2] = 5e-5;
    o0.bias_acc[0] = 1e-3; o0.bias_acc[1] = 2e-3; o0.bias_acc[2] = -1e-3;
    o0.gravity(0) = 9.81;
    OrientationUKF::Covariance Q0 = OrientationUKF::Covariance::Zero();
    const double sd[13] = {0.05, 0.05, 0.05, 0.1, 0.1, 0.1, 1e-3, 1e-3, 1e-3, 1e-2, 1e-2, 1e-2, 1e-2};
    for (int i = 0; i < 13; ++i) Q0(i, i) = sd[i] * sd[i];
    LocationConfiguration loc; loc.latitude = 0.92698121; loc.longitude = 0.15; loc.altitude = 10.0;
    OrientationUKF g(o0, Q0, 3600.0, 1800.0, loc);
    OrientationUKF::Covariance Rn = OrientationUKF::Covariance::Zero();
    const double rn[13] = {1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4, 1e-10, 1e-10, 1e-10, 1e-8, 1e-8, 1e-8, 1e-12};
    for (int i = 0; i < 13; ++i) Rn(i, i) = rn[i];
    g.setProcessNoiseCovariance(Rn);
    OrientationUKF::RotationRate w; w.mu[0] = 0.01; w.mu[1] = -0.02; w.mu[2] = 0.15;
    OrientationUKF::Acceleration a; a.mu[0] = 0.1; a.mu[1] = -0.05; a.mu[2] = 9.79;
    g.integrateMeasurement(w);
    g.integrateMeasurement(a);
    g.predictionStep(0.01);
    OrientationUKF::VelocityMeasurement vz; vz.mu[0] = 0.09; vz.mu[1] = 0.03; vz.mu[2] = -0.04;
    vz.cov = 0.0025 * OrientationUKF::VelocityMeasurement::Cov::Identity();
    g.integrateMeasurement(vz);
    bool threw_nonfinite = false;
    OrientationUKF::VelocityMeasurement bad = vz; bad.mu[1] = std::numeric_limits<double>::infinity();
    try { g.integrateMeasurement(bad); } catch (const std::runtime_error&) { threw_nonfinite = true; }
    OrientationState o; OrientationUKF::Covariance Po;
    g.getCurrentState(o, Po);
    double mo[14]; o.toArray(mo);
    OrientationUKF::RotationRate::Mu rr = g.getRotationRate();
    std::printf("\"threw_nonfinite\": %s, \"rotation_rate\": [%.17g, %.17g, %.17g], \"wgs84\": %.17g,\n", threw_nonfinite ? "true" : "false", rr[0], rr[1], rr[2],
                GravitationalModel::WGS_84(0.92698121, 10.0));
    print_state("orient", mo, 14, Po.data(), 13, false);

    // ---------------- BatchOrientationUKF: a prediction BEFORE any IMU sample must use the constructor's latches
    // (rotation_rate = 0, acceleration = (0, 0, gravity), OrientationUKF.cpp:49-50) exactly as the scalar class does
    double mu0[14]; o0.toArray(mu0);
    OrientationUKF scalar(o0, Q0, 3600.0, 1800.0, loc);
    scalar.setProcessNoiseCovariance(Rn);
    scalar.predictionStep(0.02);
    OrientationState os; OrientationUKF::Covariance Ps; scalar.getCurrentState(os, Ps);
    double ms[14]; os.toArray(ms);
    const double earth[3] = {EARTHW * std::cos(loc.latitude), 0.0, EARTHW * std::sin(loc.latitude)};
    BatchOrientationUKF batch(3, 3600.0, 1800.0, earth);
    batch.setProcessNoiseCovariance(Rn.data());
    double mu3[3 * 14], cov3[3 * 169];
    for (int i = 0; i < 3; ++i) { for (int k = 0; k < 14; ++k) mu3[i * 14 + k] = mu0[k]; for (int k = 0; k < 169; ++k) cov3[i * 169 + k] = Q0.data()[k]; }
    batch.initializeFilters(0, 3, mu3, cov3);
    batch.predictionStep(0.02);
    batch.getCurrentStates(0, 3, mu3, cov3);
    double dmax = 0.0;
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 14; ++k) dmax = std::fmax(dmax, std::fabs(mu3[i * 14 + k] - ms[k]));
        for (int k = 0; k < 169; ++k) dmax = std::fmax(dmax, std::fabs(cov3[i * 169 + k] - Ps.data()[k]));
    }
    std::printf("\"batch_orient_vs_scalar\": %.3g, \"batch_orient_dvz\": %.3g,\n", dmax, mu3[6] - mu0[6]);

    // ---------------- BatchOrientationUKF::cycles (several buffered IMU + velocity samples in ONE launch) against the same
    // samples applied one cycle() at a time: the state must agree bit for bit
    {
        const int C = 3, NB = 3;
        double gyro[C * NB * 3], accs[C * NB * 3], zz[C * NB * 3], QQ[C * NB * 9];
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < NB; ++i) {
                const int o = c * NB + i;
                gyro[o * 3 + 0] = 0.01 * (c + 1); gyro[o * 3 + 1] = -0.02; gyro[o * 3 + 2] = 0.05 * (i + 1);
                accs[o * 3 + 0] = 0.1 * i; accs[o * 3 + 1] = -0.05 * c; accs[o * 3 + 2] = 9.79;
                zz[o * 3 + 0] = 0.09; zz[o * 3 + 1] = 0.03 * c; zz[o * 3 + 2] = -0.04 * i;
                for (int k = 0; k < 9; ++k) QQ[o * 9 + k] = (k % 4 == 0) ? 0.0025 : 0.0;
            }
        BatchOrientationUKF one(NB, 3600.0, 1800.0, earth), many(NB, 3600.0, 1800.0, earth);
        double mi[NB * 14], ci[NB * 169];
        for (int i = 0; i < NB; ++i) { for (int k = 0; k < 14; ++k) mi[i * 14 + k] = mu0[k]; for (int k = 0; k < 169; ++k) ci[i * 169 + k] = Q0.data()[k]; }
        one.setProcessNoiseCovariance(Rn.data()); many.setProcessNoiseCovariance(Rn.data());
        one.initializeFilters(0, NB, mi, ci); many.initializeFilters(0, NB, mi, ci);
        for (int c = 0; c < C; ++c) {
            one.setInputs(0, NB, gyro + c * NB * 3, accs + c * NB * 3);
            one.cycle(0.01, UKFB_MEAS_ORIENT_BODYVEL3, zz + c * NB * 3, QQ + c * NB * 9);
        }
        many.cycles(C, 0.01, UKFB_MEAS_ORIENT_BODYVEL3, accs, gyro, zz, QQ);
        double ma[NB * 14], ca[NB * 169], mb[NB * 14], cb[NB * 169];
        one.getCurrentStates(0, NB, ma, ca); many.getCurrentStates(0, NB, mb, cb);
        bool same = true;
        for (int k = 0; k < NB * 14; ++k) same = same && ma[k] == mb[k];
        for (int k = 0; k < NB * 169; ++k) same = same && ca[k] == cb[k];
        std::printf("\"batch_cycles_bit_equal\": %s,\n", same ? "true" : "false");
        // ShardedBatchOrientationUKF: the same three cycles on two shards (2 + 1 filters) of device 0, latches included
        std::vector<int> devs(2, 0);
        ShardedBatchOrientationUKF sharded(NB, devs, 3600.0, 1800.0, earth);
        sharded.setProcessNoiseCovariance(Rn.data());
        sharded.initializeFilters(0, NB, mi, ci);
        sharded.predictionStep(0.02);                 // (before any IMU sample: the constructor's latches)
        BatchOrientationUKF ref(NB, 3600.0, 1800.0, earth);
        ref.setProcessNoiseCovariance(Rn.data());
        ref.initializeFilters(0, NB, mi, ci);
        ref.predictionStep(0.02);
        for (int c = 0; c < C; ++c) {
            sharded.setInputs(0, NB, gyro + c * NB * 3, accs + c * NB * 3);
            sharded.cycle(0.01, UKFB_MEAS_ORIENT_BODYVEL3, zz + c * NB * 3, QQ + c * NB * 9);
            ref.setInputs(0, NB, gyro + c * NB * 3, accs + c * NB * 3);
            ref.cycle(0.01, UKFB_MEAS_ORIENT_BODYVEL3, zz + c * NB * 3, QQ + c * NB * 9);
        }
        sharded.sync();
        ref.getCurrentStates(0, NB, ma, ca); sharded.getCurrentStates(0, NB, mb, cb);
        same = sharded.shards() == 2;
        for (int k = 0; k < NB * 14; ++k) same = same && ma[k] == mb[k] && std::isfinite(mb[k]);
        for (int k = 0; k < NB * 169; ++k) same = same && ca[k] == cb[k];
        std::printf("\"sharded_orient_bit_equal\": %s,\n", same ? "true" : "false");
    }

    // ---------------- ShardedBatchPoseUKF (ukfb_group_*): two shards -- here both on device 0, on a node one per GPU --
    // against ONE BatchPoseUKF over the same filters: the state must agree bit for bit (filters are independent)
    {
        const int NS = 7;   // shards of 4 and 3 filters
        double ms[NS * 13], cs[NS * 144], am[NS * 3], zz[NS * 3], QQ[NS * 9];
        x0.toArray(mu);
        for (int i = 0; i < NS; ++i) {
            for (int k = 0; k < 13; ++k) ms[i * 13 + k] = mu[k] + ((k < 3 || k > 6) ? 0.01 * i * (k + 1) : 0.0);
            for (int k = 0; k < 144; ++k) cs[i * 144 + k] = P0.data()[k] * (1.0 + 0.1 * i);
            am[i * 3 + 0] = 0.2; am[i * 3 + 1] = -0.1 * i; am[i * 3 + 2] = 0.05;
            zz[i * 3 + 0] = ms[i * 13 + 0] + 0.02; zz[i * 3 + 1] = ms[i * 13 + 1] - 0.01; zz[i * 3 + 2] = ms[i * 13 + 2];
            for (int k = 0; k < 9; ++k) QQ[i * 9 + k] = (k % 4 == 0) ? 0.0025 : 0.0;
        }
        const double acov[9] = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01};
        BatchPoseUKF single(NS);
        std::vector<int> devs(2, 0);
        ShardedBatchPoseUKF sharded(NS, devs);
        single.initializeFilters(0, NS, ms, cs); sharded.initializeFilters(0, NS, ms, cs);
        single.setAccelerations(0, NS, am, acov); sharded.setAccelerations(0, NS, am, acov);
        single.cycle(0.01, UKFB_MEAS_POS3, zz, QQ); sharded.cycle(0.01, UKFB_MEAS_POS3, zz, QQ);
        single.predictionStep(0.02); sharded.predictionStep(0.02);
        // the asynchronous stream over both shards: filter 5 has two samples that arrive out of order, filters 1 and 6 one each
        {
            const int64_t ef[4] = {5, 1, 6, 5}, et[4] = {2030000, 2010000, 2010000, 2010000};
            const int32_t em[4] = {UKFB_MEAS_POS3, UKFB_MEAS_POS3, -1, UKFB_MEAS_POS3};
            double ez[12], eq[36];
            for (int j = 0; j < 4; ++j) {
                for (int k = 0; k < 3; ++k) ez[j * 3 + k] = zz[ef[j] * 3 + k] + 0.001 * (j + 1);
                for (int k = 0; k < 9; ++k) eq[j * 9 + k] = QQ[k];
            }
            const int64_t r1 = single.processEvents(4, ef, et, em, ez, eq), r2 = sharded.processEvents(4, ef, et, em, ez, eq);
            if (r1 != 2 || r2 != 2) std::printf("\"sharded_event_rounds\": [%lld, %lld],\n", (long long)r1, (long long)r2);
        }
        sharded.sync();
        double ma[NS * 13], ca[NS * 144], mb[NS * 13], cb[NS * 144];
        single.getCurrentStates(0, NS, ma, ca); sharded.getCurrentStates(0, NS, mb, cb);
        uint32_t st_single = 0u;
        {
            const std::vector<uint32_t> st = single.status();
            for (size_t k = 0; k < st.size(); ++k) st_single |= st[k];
        }
        bool same = sharded.shards() == 2 && sharded.statusSummary() == st_single;
        double moved = 0.0;
        for (int k = 0; k < NS * 13; ++k) { same = same && ma[k] == mb[k]; moved = std::fmax(moved, std::fabs(ma[k] - ms[k])); }
        for (int k = 0; k < NS * 144; ++k) same = same && ca[k] == cb[k];
        int64_t f1 = -1, c1 = -1; int d1 = -1;
        sharded.shard(1, &d1, &f1, &c1);
        std::printf("\"sharded_bit_equal\": %s, \"sharded_moved\": %.3g, \"sharded_shard1\": [%d, %lld, %lld]\n", same ? "true" : "false", moved, d1,
                    (long long)f1, (long long)c1);
    }
    std::printf("}\n");
    return 0;
}
