// spec_k_v3d.hip -- instantiations of the fp64 member of the packed kernel family (spec_v3d.h)
#include "spec_v3d.h"

namespace specgpu {

bool v3d_applicable(int log2n, int kind, uint64_t n_lines, uint32_t hop) {
    if (log2n < 8 || log2n > 13) return false;
    if (kind != K_CF64 && kind != K_CF32 && kind != K_CI16 && kind != K_CU8 && kind != K_CI8) return false;
    return n_lines < (1ull << 31) && hop <= (8u << log2n);
}

hipError_t launch_v3d_spectro(const WfArgs &w, int log2n, uint32_t run, hipStream_t s) {
    V2Args a{};
    a.iq = w.iq; a.unit_stride = 0; a.n_units = 1; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.run = run;
    const uint32_t per_wg = (log2n == 13 ? 1u : (uint32_t)v2_lpw(log2n)) * run;  // plan 113: one line per workgroup
    a.wgs_per_unit = (a.n_lines + per_wg - 1) / per_wg;
    a.tw = w.tw; a.win = w.win; a.out = w.out; a.out_fmt = w.out_fmt; a.be = w.be; a.win_hann = w.win_hann;
    switch (log2n) {
    case 8: return v3d_launch_kind<8>(a, w.kind, s);
    case 9: return v3d_launch_kind<9>(a, w.kind, s);
    case 10: return v3d_launch_kind<10>(a, w.kind, s);
    case 11: return v3d_launch_kind<11>(a, w.kind, s);
    case 12: return v3d_launch_kind<12>(a, w.kind, s);
    case 13: return v3d_launch_kind<113>(a, w.kind, s);
    default: return hipErrorInvalidValue;
    }
}

// fp64 Welch partial sums (spec_v3d.h MODE 1): w.partial receives double slabs [n_psd][wgs * LPW][N]
int v3d_lpw(int log2n) { return log2n == 13 ? 1 : v2_lpw(log2n); }
hipError_t launch_v3d_welch(const WelchArgs &w, int log2n, uint32_t run, uint32_t wgs_per_unit, hipStream_t s) {
    V2Args a{};
    a.iq = w.iq; a.unit_stride = w.psd_stride_bytes; a.n_units = w.n_psd; a.n_lines = w.n_seg; a.hop = w.hop;
    a.run = run; a.wgs_per_unit = wgs_per_unit; a.tw = w.tw; a.win = w.win; a.out = w.partial; a.out_fmt = 0; a.be = w.be; a.win_hann = w.win_hann;
    switch (log2n) {
    case 8: return v3d_launch_welch_kind<8>(a, w.kind, s);
    case 9: return v3d_launch_welch_kind<9>(a, w.kind, s);
    case 10: return v3d_launch_welch_kind<10>(a, w.kind, s);
    case 11: return v3d_launch_welch_kind<11>(a, w.kind, s);
    case 12: return v3d_launch_welch_kind<12>(a, w.kind, s);
    case 13: return v3d_launch_welch_kind<113>(a, w.kind, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace specgpu
