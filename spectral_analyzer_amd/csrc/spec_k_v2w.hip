// spec_k_v2w.hip -- Welch instantiations of the packed-fp32 kernel family (spec_v2.h)
#include "spec_v2.h"

namespace specgpu {

#ifdef SPEC_V2_STAMPS
// development build only (build.py --variant v2stamp, tools/v2_timeline.py): where the kernels leave their phase stamps
static void *g_stamp_buf = nullptr;
static uint32_t g_stamp_first = 0;
extern "C" __attribute__((visibility("default"))) void spec_debug_v2_stamps(void *device_buf, uint32_t first_segment) {
    g_stamp_buf = device_buf;
    g_stamp_first = first_segment;
}
#endif


// Welch partial sums: slabs [n_psd][wgs_per_unit * LPW][N]
hipError_t launch_v2_welch(const WelchArgs &w, int log2n, uint32_t run, uint32_t wgs_per_unit, hipStream_t s) {
    V2Args a{};
    a.iq = w.iq; a.unit_stride = w.psd_stride_bytes; a.n_units = w.n_psd; a.n_lines = w.n_seg; a.hop = w.hop;
    a.run = run; a.wgs_per_unit = wgs_per_unit; a.tw = w.tw; a.win = w.win; a.out = w.partial; a.out_fmt = 0; a.be = w.be; a.win_hann = w.win_hann;
    a.final_out = wgs_per_unit == 1 && v2_lpw(log2n) == 1 ? w.final_out : nullptr; a.norm = w.norm; a.db = w.db;
    a.rows = log2n == 14 ? w.rows : 0;
#ifdef SPEC_V2_STAMPS
    a.sel = static_cast<const int32_t *>(g_stamp_buf);
    a.out_stride = g_stamp_first;
#endif
    return v2_launch_n<1>(a, log2n, w.kind, s);
}

}  // namespace specgpu
