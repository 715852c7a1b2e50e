// spec_k_f32.hip -- fp32 instantiations of the generic kernels (gfx950).
#include "spec_kernels.h"

namespace specgpu {

int plan_lpw(int log2n) {
    switch (log2n) {
    case 1: return Plan<1>::LPW;   case 2: return Plan<2>::LPW;   case 3: return Plan<3>::LPW;
    case 4: return Plan<4>::LPW;   case 5: return Plan<5>::LPW;
    case 6: return Plan<6>::LPW;   case 7: return Plan<7>::LPW;   case 8: return Plan<8>::LPW;
    case 9: return Plan<9>::LPW;   case 10: return Plan<10>::LPW; case 11: return Plan<11>::LPW;
    case 12: return Plan<12>::LPW; case 13: return Plan<13>::LPW; case 14: return Plan<14>::LPW;
    default: return 0;
    }
}

bool plan_supported(int log2n, bool f64) { return log2n >= 1 && log2n <= (f64 ? 13 : 14); }

hipError_t launch_spectro_f32(const WfArgs &a, int log2n, hipStream_t s) {
    switch (log2n) {
    case 1: return launch_spectro_one<float, 1>(a, s);
    case 2: return launch_spectro_one<float, 2>(a, s);
    case 3: return launch_spectro_one<float, 3>(a, s);
    case 4: return launch_spectro_one<float, 4>(a, s);
    case 5: return launch_spectro_one<float, 5>(a, s);
    case 6: return launch_spectro_one<float, 6>(a, s);
    case 7: return launch_spectro_one<float, 7>(a, s);
    case 8: return launch_spectro_one<float, 8>(a, s);
    case 9: return launch_spectro_one<float, 9>(a, s);
    case 10: return launch_spectro_one<float, 10>(a, s);
    case 11: return launch_spectro_one<float, 11>(a, s);
    case 12: return launch_spectro_one<float, 12>(a, s);
    case 13: return launch_spectro_one<float, 13>(a, s);
    case 14: return launch_spectro_one<float, 14>(a, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace specgpu
