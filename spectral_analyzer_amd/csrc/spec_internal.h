// spec_internal.h -- host-side interface between the C ABI (spec_capi.hip) and
// the kernel translation units.  Not installed; include/specgpu.h is the public
// header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace specgpu {

// out_fmt values mirror spec_out_fmt
enum : int { OUT_DB20_F32 = 0, OUT_POW_F32 = 1, OUT_DB20_F64 = 2, OUT_POW_F64 = 3 };

// Arguments of one spectrogram launch.  `iq` already points at the first byte
// of line 0; every line is known to be inside the buffer (the host clips).
struct WfArgs {
    const uint8_t *iq;
    uint64_t n_lines;
    uint32_t hop;        // samples between line starts
    uint32_t bps;        // bytes per IQ pair
    int kind;            // K_* decode kind (spec_fft.h)
    int be;              // big-endian components
    const void *tw;      // cx<R>[N] twiddle table W_N^m
    const void *win;     // R[N] window or nullptr (rectangular)
    int win_hann = 0;    // 1: `win` is the periodic Hann table (w[N/2 - n] = w[n + N/2] = 1 - w[n]: spec_v2.h keeps a quarter of it in LDS)
    void *out;           // n_lines x N, row-major
    int out_fmt;
    uint32_t lines_per_wg;  // contiguous lines handled by one workgroup (multiple of LPW)
};

// Arguments of one Welch partial-sum launch (spec_v2.h, MODE 1): every sub-line sums |X|^2
// over its run of segments into one slab of partial[psd][slab][N].
struct WelchArgs {
    const uint8_t *iq;          // first byte of segment 0 of PSD 0
    uint64_t psd_stride_bytes;
    uint32_t n_psd, n_seg;
    uint32_t hop, bps;
    int kind, be;
    const void *tw, *win;
    int win_hann = 0;           // as WfArgs; 2: `win` is all ones (rectangular Welch)
    void *partial;              // [n_psd][slabs][N] unshifted power sums (fp32; fp64 for launch_v3d_welch)
    // launch_v2_welch with ONE workgroup and ONE sub-line per PSD (wgs_per_unit == 1, whole-workgroup lines): the
    // kernel finishes the PSD itself -- sum * norm, fftshift, optional 10 log10 -- into final_out (float[n_psd][N])
    // and no slab is written; nullptr: slabs + launch_welch_finalize
    void *final_out = nullptr;
    double norm = 0.0;
    int db = 0;
    int rows = 0;               // experiment library only ("welch_rows", -DSPEC_V2_ROWS): 16384 points through the plan 16 x (32 x 32)
};

int plan_lpw(int log2n);  // lines a workgroup transforms concurrently
bool plan_supported(int log2n, bool f64);

hipError_t launch_spectro_f32(const WfArgs &a, int log2n, hipStream_t s);
hipError_t launch_spectro_f64(const WfArgs &a, int log2n, hipStream_t s);

// large-N four-step path (spec_k_large.hip): w.tw is the W_N table, tw1/tw2 the
// W_N1 / W_N2 tables of the split, scratch holds n_lines * N complex values
bool large_split(int log2n, bool f64, int *l1, int *l2);
size_t large_scratch_bytes_per_line(int log2n, bool f64);
// run_if != nullptr: the kernels start only when *run_if != 0 (the fall-back behind the team kernel)
hipError_t launch_spectro_large(const WfArgs &w, int log2n, bool f64, const void *tw1, const void *tw2,
                                void *scratch, hipStream_t s, const uint32_t *run_if = nullptr);
// the fall-back behind the team kernel as ONE guarded launch over all lines: every workgroup owns whole lines and a
// line-sized intermediate of its own (`scratch` = grid * N complex values); nothing in it waits for another workgroup
hipError_t launch_spectro_large_solo(const WfArgs &w, int log2n, bool f64, const void *tw1, const void *tw2, void *scratch,
                                     uint32_t grid, hipStream_t s, const uint32_t *run_if);
// the same decomposition as ONE persistent launch with the intermediate kept in each XCD's L2 (spec_k_team.hip):
// `sync` (large_team_sync_bytes(), zeroed on the stream before the call) carries the tickets and ring counters;
// word large_team_abort_word() is non-zero afterwards when a bounded wait timed out (output incomplete).
// query_only: report in *teams_max how many teams the launch can form at most (scratch = teams_max * ring * nfft
// complex values) without launching.
size_t large_team_sync_bytes();
uint32_t large_team_abort_word();
uint32_t large_team_prof_offset_bytes();  // development builds (-DSPEC_TEAM_PROF): per-workgroup wait cycles behind the block
hipError_t launch_spectro_team(const WfArgs &w, int log2n, bool f64, const void *tw1, const void *tw2, void *scratch,
                               uint32_t ring, uint32_t *sync, int n_cu, uint32_t *teams_max, bool query_only,
                               hipStream_t s, int wg = 512, uint32_t block = 0);

// packed-fp32 family (spec_v2.h): every LDS-resident size, cf32/ci16/cu8/ci8 little endian
bool v2_applicable(int log2n, int kind, int be, int out_fmt, uint64_t n_lines, uint32_t hop);
int v2_lpw(int log2n);  // sub-lines per workgroup
hipError_t launch_v2_spectro(const WfArgs &w, int log2n, uint32_t run, hipStream_t s);
// fp64 member of the family (spec_v3d.h): 256 ... 4096 points, any sample format, fp64 arithmetic
bool v3d_applicable(int log2n, int kind, uint64_t n_lines, uint32_t hop);
hipError_t launch_v3d_spectro(const WfArgs &w, int log2n, uint32_t run, hipStream_t s);
// 64- and 128-point lines (spec_k_v2n.hip): a wave works on 16 / 8 consecutive lines at a time, all global traffic in
// 16-byte-per-lane pieces through its own LDS region; `first` = address of the first line's first byte (alignment)
bool v2n_applicable(int log2n, int kind, int out_fmt, uint64_t n_lines, uint32_t hop, const void *first, int max_log2n);
hipError_t launch_v2n_spectro(const WfArgs &w, int log2n, int n_cu, hipStream_t s);
// 32768- (and, as an option, 16384-) point fp32 lines in one workgroup (spec_k_v2h.hip): w.tw = v2f W_N table, tw_half = v2f
// W_(N/2) table, w.win = non-null for the Hann window; `run` consecutive lines per workgroup
bool v2h_applicable(int log2n, int kind, int out_fmt, uint64_t n_lines, uint32_t hop);
hipError_t launch_v2h_spectro(const WfArgs &w, int log2n, const void *tw_half, const void *tw_full64, uint32_t run, hipStream_t s);  // tw_full64: v2d W_N table, needed with a window
// 65536-point fp32 lines by PAIRS of workgroups (spec_k_v2q.hip): w.tw = v2f W_65536 table, tw_q = v2f W_16384 table,
// tw_full64 = v2d W_65536 table (needed with a window); `run` consecutive lines per pair
bool v2q_applicable(int log2n, int kind, int out_fmt, uint64_t n_lines, uint32_t hop);
// interleave: the sixteen pairs of an XCD take every sixteenth line of a shared block of 16 * run lines (L2 then serves the overlap)
hipError_t launch_v2q_spectro(const WfArgs &w, const void *tw_q, const void *tw_full64, uint32_t run, int interleave, hipStream_t s);
// 16384-point fp64 lines in one workgroup (spec_k_v3h.hip): w.tw = v2d W_16384 table, tw_half = v2d W_8192 table
bool v3h_applicable(int log2n, int kind, uint64_t n_lines, uint32_t hop);
// 32768-point fp64 lines by PAIRS of workgroups (spec_k_v3h.hip, v3q_kernel): w.tw = v2d W_32768 table, tw_q = v2d W_8192 table
bool v3q_applicable(int log2n, int kind, uint64_t n_lines, uint32_t hop);
hipError_t launch_v3q_spectro(const WfArgs &w, const void *tw_q, uint32_t run, int interleave, hipStream_t s);
hipError_t launch_v3h_spectro(const WfArgs &w, const void *tw_half, uint32_t run, hipStream_t s);
bool v2_sel_applicable(int log2n, int kind, int be, uint64_t n_lines, uint32_t hop);
hipError_t launch_v2_spectro_sel(const WfArgs &w, int log2n, uint32_t run, const int32_t *sel, uint32_t out_stride,
                                 hipStream_t s);
hipError_t launch_v2_welch(const WelchArgs &w, int log2n, uint32_t run, uint32_t wgs_per_unit, hipStream_t s);

hipError_t launch_fill(void *out, uint64_t n_elems, double value, int is_f64, hipStream_t s);
// *out += sum_i word_i * (2 i + 1) mod 2^64 over the n_bytes / 4 words at p ("multi_verify")
hipError_t launch_checksum(const void *p, uint64_t n_bytes, unsigned long long *out, hipStream_t s);
// slabs_f64 / out_f64: element type of the slabs / of psd_out
hipError_t launch_welch_finalize(const void *partial, int slabs_f64, uint32_t n_psd, uint32_t n_slabs,
                                 uint32_t nfft, double norm, int db, void *psd_out, int out_f64, hipStream_t s);
// fp64 member: double slabs (spec_v3d.h MODE 1); v3d_lpw = sub-lines per workgroup of that family
int v3d_lpw(int log2n);
hipError_t launch_v3d_welch(const WelchArgs &w, int log2n, uint32_t run, uint32_t wgs_per_unit, hipStream_t s);
// fallback Welch: acc[k] += sum over n lines of fftshifted power lines (float or double);
// then scale / dB into psd_out
hipError_t launch_welch_accum(const void *lines, int lines_f64, uint64_t n, uint32_t nfft, double *acc, hipStream_t s);
hipError_t launch_welch_scale(const double *acc, uint32_t nfft, double norm, int db, void *psd_out, int out_f64,
                              hipStream_t s);
// Welch PSD of any length by a plain fp64 DFT (non power-of-two nfft, ADC:303-307): tw = cx<double>[nfft] table
// W_N^m, win = double[nfft] or nullptr; writes the scaled, fftshifted PSDs (float or double)
hipError_t launch_welch_dft(const uint8_t *iq, uint64_t psd_stride_bytes, uint32_t n_psd, uint32_t n_seg, uint32_t hop,
                            uint32_t bps, int kind, int be, uint32_t nfft, const void *tw, const void *win, double norm,
                            int db, void *out, int out_f64, hipStream_t s);
// compact != 0: tile is [width][height] with column f = the bin pixel row f samples (launch_v2_spectro_sel)
hipError_t launch_render(const float *tile, uint32_t width, uint32_t nfft, uint32_t height, double conversion,
                         double min_db, double max_db, int colormap, int compact, void *bgra, hipStream_t s);
hipError_t launch_interleave(const double *re, const double *im, void *out, uint64_t n, hipStream_t s);
hipError_t launch_synth(void *out, int kind, int be, uint64_t seed, uint64_t first_sample,
                        uint64_t n_samples, hipStream_t s);

// burst chain (spec_burst.hip): reader + mixer, FIR + decimate, EMA traces (fp64)
hipError_t launch_extract_mix(const uint8_t *raw, int kind, int be, uint32_t stride, uint64_t count, double freq_off,
                              double *re, double *im, hipStream_t s);
hipError_t launch_boxcar_decim(const uint8_t *raw, int kind, int be, uint32_t stride, double freq_off, uint32_t down,
                               double *ore, double *oim, uint64_t n_out, hipStream_t s);
hipError_t launch_fir_decim(const double *mr, const double *mi, uint64_t n, const double *h, uint32_t K, uint32_t c,
                            uint32_t down, double *ore, double *oim, uint64_t n_out, hipStream_t s);
bool mix_fir_applicable(uint32_t K, uint32_t down);
hipError_t launch_mix_fir(const uint8_t *raw, int kind, int be, uint32_t stride, uint64_t n, double freq_off,
                          const double *h, uint32_t K, uint32_t c, uint32_t down, double *ore, double *oim, uint64_t n_out,
                          hipStream_t s);
size_t trace_scratch_bytes(uint64_t n_out);
// kind_trace 0: magnitude (n_out = n), 1: instantaneous frequency (n_out = n - 1)
hipError_t launch_trace(int kind_trace, const double *re, const double *im, uint64_t n_out, double alpha, double fs,
                        double add, void *scratch, double *out, hipStream_t s);

}  // namespace specgpu
