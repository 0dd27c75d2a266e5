// Instantiates the ukf_kernel variants for (double, OrientM) -- one translation unit per pair so the
// four compile in parallel (see Makefile).
#include "ukf_launch.inc.hpp"

namespace ukfb {
int launch_orient_f64(ukfb_engine* e, const LaunchReq& r) { return launch_typed<double, OrientM<double>>(e, r); }
}  // namespace ukfb
