// spec_k_v2r.hip -- "selected bins" instantiations of the packed-fp32 kernel family (spec_v2.h, MODE 2):
// spectrogram lines of which only the bins the renderer samples are stored (one per pixel row, MC:1280)
#include "spec_v2.h"

namespace specgpu {

bool v2_sel_applicable(int log2n, int kind, int be, uint64_t n_lines, uint32_t hop) {
    // 16-point threads only: a 32-point thread has no registers left for its 32 column offsets
    return log2n <= 12 && v2_applicable(log2n, kind, be, OUT_DB20_F32, n_lines, hop);
}

// compact dB lines [n_lines][out_stride]; sel[k] = column of unshifted bin k, or -1
hipError_t launch_v2_spectro_sel(const WfArgs &w, int log2n, uint32_t run, const int32_t *sel, uint32_t out_stride,
                                 hipStream_t s) {
    V2Args a{};
    a.iq = w.iq; a.unit_stride = 0; a.n_units = 1; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.run = run;
    const uint32_t per_wg = (uint32_t)v2_lpw(log2n) * run;
    a.wgs_per_unit = (a.n_lines + per_wg - 1) / per_wg;
    a.tw = w.tw; a.win = w.win; a.out = w.out; a.out_fmt = OUT_DB20_F32; a.be = w.be;
    a.sel = sel; a.out_stride = out_stride;
    return v2_launch_n<2>(a, log2n, w.kind, s);
}

}  // namespace specgpu
