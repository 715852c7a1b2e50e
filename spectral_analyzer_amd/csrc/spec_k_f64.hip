// spec_k_f64.hip -- fp64 instantiations (cf64 input and the DB20_F64 / POW_F64
// strict-parity outputs: the whole pipeline in double, as the reference).
#include "spec_kernels.h"

namespace specgpu {

hipError_t launch_spectro_f64(const WfArgs &a, int log2n, hipStream_t s) {
    switch (log2n) {
    case 1: return launch_spectro_one<double, 1>(a, s);
    case 2: return launch_spectro_one<double, 2>(a, s);
    case 3: return launch_spectro_one<double, 3>(a, s);
    case 4: return launch_spectro_one<double, 4>(a, s);
    case 5: return launch_spectro_one<double, 5>(a, s);
    case 6: return launch_spectro_one<double, 6>(a, s);
    case 7: return launch_spectro_one<double, 7>(a, s);
    case 8: return launch_spectro_one<double, 8>(a, s);
    case 9: return launch_spectro_one<double, 9>(a, s);
    case 10: return launch_spectro_one<double, 10>(a, s);
    case 11: return launch_spectro_one<double, 11>(a, s);
    case 12: return launch_spectro_one<double, 12>(a, s);
    case 13: return launch_spectro_one<double, 13>(a, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace specgpu
