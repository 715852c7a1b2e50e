// spec_k_v2s.hip -- spectrogram instantiations of the packed-fp32 kernel family (spec_v2.h)
#include "spec_v2.h"

namespace specgpu {

bool v2_applicable(int log2n, int kind, int be, int out_fmt, uint64_t n_lines, uint32_t hop) {
    // 64 and 128 points stay on the generic kernel: with 16 points per thread a line would be
    // 4 or 8 lanes wide (32 / 64-byte global segments) and measured slower
    if (log2n < 8 || log2n > 14) return false;
    (void)be;  // either byte order
    if (kind != K_CF32 && kind != K_CI16 && kind != K_CU8 && kind != K_CI8) return false;
    if (out_fmt != OUT_DB20_F32 && out_fmt != OUT_POW_F32) return false;
    // 32-bit offsets inside a workgroup's span
    return n_lines < (1ull << 31) && hop <= (8u << log2n);  // span of a workgroup stays far below 4 GiB
}

int v2_lpw(int log2n) {
    switch (log2n) {
    case 8: return Plan2<8>::LPW;
    case 9: return Plan2<9>::LPW;   case 10: return Plan2<10>::LPW; case 11: return Plan2<11>::LPW;
    case 12: return Plan2<12>::LPW; case 13: return Plan2<13>::LPW; case 14: return Plan2<14>::LPW;
    default: return 1;
    }
}

// spectrogram over one unit; `run` lines per sub-line
hipError_t launch_v2_spectro(const WfArgs &w, int log2n, uint32_t run, hipStream_t s) {
    V2Args a{};
    a.iq = w.iq; a.unit_stride = 0; a.n_units = 1; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.run = run;
    const uint32_t per_wg = (uint32_t)v2_lpw(log2n) * run;
    a.wgs_per_unit = (a.n_lines + per_wg - 1) / per_wg;
    a.tw = w.tw; a.win = w.win; a.out = w.out; a.out_fmt = w.out_fmt; a.be = w.be; a.win_hann = w.win_hann;
    return v2_launch_n<0>(a, log2n, w.kind, s);
}

}  // namespace specgpu
