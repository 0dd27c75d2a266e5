// prints the compile-time lane tables of the covariance phase (ukf_kernel16.hpp: CovTab) as JSON; host code only
#include <cstdio>
#include "ukf_kernel16.hpp"
template <class T, class M> static void dump(const char* name, bool last) {
    using CT = ukfb::CovTab<T, M>;
    using LY = ukfb::Layout16<T, M>;
    constexpr auto t = CT::make();
    std::printf("\"%s\": {\"SZ\": %d, \"D\": %d, \"NL\": %d, \"TR\": %d, \"TC\": %d, \"AEL\": %d, \"PKS\": %d, \"DUM\": %d, \"TNL\": %d, \"LAF\": %d, \"ST\": %d, \"TRIP\": %d, \"NRD\": %d, \"NWR\": %d,\n \"rd\": [",
                name, CT::SZ, CT::D, CT::NL, CT::TR, CT::TC, CT::AEL, LY::PKS, LY::DUM, LY::TNL, LY::LAF, LY::ST, LY::TRIP, CT::NRD, CT::NWR);
    for (int l = 0; l < 16; ++l) {
        std::printf("%s[", l ? ", " : "");
        for (int k = 0; k < CT::NRD; ++k) std::printf("%s%u", k ? ", " : "", t.rd[l][k]);
        std::printf("]");
    }
    std::printf("],\n \"wr\": [");
    for (int l = 0; l < 17; ++l) {
        std::printf("%s[", l ? ", " : "");
        for (int k = 0; k < CT::NWR; ++k) std::printf("%s%u", k ? ", " : "", t.wr[l][k]);
        std::printf("]");
    }
    std::printf("]}%s\n", last ? "" : ",");
}
// the OrientationState kernels' 16-bit tables (OCovTab)
template <class T, class M> static void dump_o(const char* name) {
    using OT = ukfb::OCovTab<T, M>;
    using LY = ukfb::Layout16<T, M>;
    constexpr auto t = OT::make();
    std::printf("\"%s\": {\"SZ\": %d, \"D\": %d, \"NL\": %d, \"TR\": %d, \"TC\": %d, \"AEL\": %d, \"PKS\": %d, \"DUM\": %d, \"NSH_SINK\": %d, \"TNL\": %d, \"LAF\": %d, \"ST\": %d, \"TRIP\": %d, \"NRD\": %d, \"NWR\": %d,\n \"rd\": [",
                name, OT::SZ, OT::D, OT::NL, OT::TR, OT::TC, OT::AEL, LY::PKS, LY::DUM, LY::NSH_SINK, LY::TNL, LY::LAF, LY::ST, LY::TRIP, OT::NRD, OT::NWR);
    for (int l = 0; l < 16; ++l) {
        std::printf("%s[", l ? ", " : "");
        for (int k = 0; k < OT::NRD; ++k) std::printf("%s%u", k ? ", " : "", unsigned(t.rd[l][k]));
        std::printf("]");
    }
    std::printf("],\n \"wr\": [");
    for (int l = 0; l < 17; ++l) {
        std::printf("%s[", l ? ", " : "");
        for (int k = 0; k < OT::NWR; ++k) std::printf("%s%u", k ? ", " : "", unsigned(t.wr[l][k]));
        std::printf("]");
    }
    std::printf("]},\n");
}
int main() {
    std::printf("{\n");
    dump_o<double, ukfb::OrientM<double>>("orient_f64_o");
    dump_o<float, ukfb::OrientM<float>>("orient_f32_o");
    dump<double, ukfb::PoseM<double>>("pose_f64", false);
    dump<float, ukfb::PoseM<float>>("pose_f32", false);
    dump<float, ukfb::OrientM<float>>("orient_f32", false);
    // XCD-aware numbering of the workgroups (group_of_block) for a few grid sizes
    std::printf("\"group_of_block\": {");
    const unsigned sizes[] = {1, 5, 8, 9, 13, 64, 1000, 16384};
    for (unsigned k = 0; k < sizeof(sizes) / sizeof(sizes[0]); ++k) {
        std::printf("%s\"%u\": [", k ? ", " : "", sizes[k]);
        for (unsigned b = 0; b < sizes[k]; ++b) std::printf("%s%u", b ? ", " : "", ukfb::group_of_block(b, sizes[k]));
        std::printf("]");
    }
    std::printf("}\n}\n");
    return 0;
}
