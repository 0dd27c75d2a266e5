// Instantiates the ukf_kernel variants for (double, PoseM) -- one translation unit per pair so the
// four compile in parallel (see Makefile).
#include "ukf_launch.inc.hpp"

namespace ukfb {
int launch_pose_f64(ukfb_engine* e, const LaunchReq& r) { return launch_typed<double, PoseM<double>>(e, r); }
}  // namespace ukfb
