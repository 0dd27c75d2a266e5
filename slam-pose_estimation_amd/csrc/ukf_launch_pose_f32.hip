// Instantiates the ukf_kernel variants for (float, PoseM) -- one translation unit per pair so the
// four compile in parallel (see Makefile).
#include "ukf_launch.inc.hpp"

namespace ukfb {
int launch_pose_f32(ukfb_engine* e, const LaunchReq& r) { return launch_typed<float, PoseM<float>>(e, r); }
}  // namespace ukfb
