// Instantiates the WIDE-ARITHMETIC kernels of the fp32 engines for OrientM: fp32 arrays in HBM (KArgs<float>), the tuned
// kernel computing in double (ukf_kernel16<double, OrientM<double>, ..., float>).  ukfb_config::wide_arithmetic selects them.
#define UKFB_LAUNCH_WIDE 1
#include "ukf_launch.inc.hpp"

namespace ukfb {
int launch_orient_f32w(ukfb_engine* e, const LaunchReq& r) { return launch_typed<float, OrientM<float>>(e, r); }
}  // namespace ukfb
