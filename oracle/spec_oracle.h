/*
 * spec_oracle.h -- CPU restatement of the reference spectrogram / PSD hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call it, and there only as the checker / the reported CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" against an execution of the reference -- the
 * reference is Java (no JDK / jars in the build container, no reference tests
 * with numeric vectors).  The restatement is pinned instead against
 *   (1) analytic known-answer tests derived from the formulas at
 *       SpectralService.java:40-82 (tests/test_oracle.py, K1..K9), and
 *   (2) two independent implementations of the DFT / Welch definitions
 *       (numpy.fft.fft, scipy.signal.welch), see oracle/spec_oracle.py, and
 *   (3) the literals of commons-math3's root-of-unity tables known from its published source
 *       (tools/gen_cm3_roots.py).
 *
 * Reference files restated (paths relative to the reference repo root,
 * src/main/java/net/kcundercover/spectral_analyzer/...):
 *   services/SpectralService.java:33-85         computeMagnitudes
 *   services/ExtractDownConvertService.java:60-97   cf64 decode, bytes per IQ
 *   sigmf/Global.java:67-79                     getBytesPerSample
 *   sigmf/SigMfHelper.java:87-91                byte order rule
 *   controllers/MainController.java:980-999     line loop, range test, -150 fill
 *   controllers/MainController.java:1270-1283   dB/Hz display normalisation
 *   controllers/AnalysisDialogController.java:303-313   PSD call shape
 *   controllers/AnalysisDialogController.java:219-284   magnitude / instantaneous-frequency traces
 *   services/ExtractDownConvertService.java:54-117      burst reader + down-converter call
 * Third-party arithmetic (not in the reference tree, restated from the
 * published definitions):
 *   org.apache.commons:commons-math3:3.6.1  FastFourierTransformer(STANDARD).
 *       transform(x, FORWARD): X[k] = sum_n x[n] exp(-2 pi i k n / N), unscaled,
 *       power-of-two N.  Restated as PUBLISHED (so_fft_forward_cm3): bitReversalShuffle2, a
 *       4-term first stage, then per stage a multiplicative twiddle recurrence seeded from the
 *       library's 63-entry root tables; Complex.abs() in its scaled sqrt(1 + q^2) form.  The
 *       recurrence costs accuracy (twiddle r of a stage is off by ~r eps): the reference's lines
 *       carry that error and so does this oracle; so_fft_forward_exact (twiddles evaluated in long
 *       double, rounded once) is kept beside it as the accuracy yardstick.
 *   com.github.GassiusODude:jdsp:v1.3.1  PowerSpectralDensity.calculatePsdWelch:
 *       source absent -> this file defines the build's PSD (Welch, see
 *       so_welch_psd) and is checked against scipy.signal.welch only.
 */
#ifndef SPEC_ORACLE_H
#define SPEC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* window ids / psd scaling ids (same numeric values as include/specgpu.h) */
enum { SO_WIN_RECT = 0, SO_WIN_HANN = 1 };
enum { SO_PSD_DENSITY = 0, SO_PSD_SPECTRUM = 1 };

/* Global.java:67-79 -- bytes per interleaved IQ pair for a SigMF datatype
 * string (startsWith matching; unknown -> 8, the reference's fallback). */
int so_bytes_per_sample(const char *datatype);

/* SigMfHelper.java:87-91 -- 1 when the buffer would be BIG endian, i.e. the
 * datatype does not end in "_le". */
int so_is_big_endian(const char *datatype);

/* SpectralService.java:40-65 (+ cf64 per ExtractDownConvertService.java:79-81
 * when cf64_decode != 0; cf64_decode == 0 reproduces the reference defect:
 * computeMagnitudes has no cf64 branch, so the sample decodes to 0+0i).
 * Decodes sample i of the line starting at byte start_byte. */
void so_decode_sample(const uint8_t *buf, uint64_t start_byte, uint64_t i,
                      const char *datatype, int cf64_decode,
                      double *re, double *im);

/* which transform a waterfall is computed with */
enum { SO_FFT_CM3 = 0,   /* the reference's: commons-math3 3.6.1 as published (twiddle recurrence) */
       SO_FFT_EXACT = 1  /* yardstick: same radix-2 structure, exact twiddles, hypot() */ };

/* commons-math3 3.6.1 FastFourierTransformer(STANDARD), FORWARD, restated as published: in-place,
 * unscaled, n must be a power of two (returns -1 otherwise, where the library throws).
 * so_fft_forward is the same function (the transform the reference calls, SS:68). */
int so_fft_forward_cm3(double *re, double *im, uint32_t n);
int so_fft_forward(double *re, double *im, uint32_t n);
/* accuracy yardstick, NOT the reference's arithmetic: every twiddle evaluated in long double and
 * rounded to double once */
int so_fft_forward_exact(double *re, double *im, uint32_t n);
/* org.apache.commons.math3.complex.Complex.abs() (SS:80), restated */
double so_complex_abs(double re, double im);

/* SpectralService.java:33-85 -- one spectrogram line, out[nfft]:
 * out[(i + n/2) % n] = 20 log10(|X_i| + 1e-10). Returns 0, or -1 on bad nfft. */
int so_compute_magnitudes(const uint8_t *buf, uint64_t start_byte, uint32_t nfft,
                          const char *datatype, int cf64_decode, double *out);

/* MainController.java:980-999 with a hop parameter (reference: hop == nfft,
 * window == RECT).  Line t starts at byte start_byte + t*hop*bps; a line whose
 * last byte would exceed `capacity` is filled with eof_fill (-150.0 in the
 * reference).  out is [n_lines][nfft] row-major.  window HANN is the periodic
 * Hann w[n] = 0.5 - 0.5 cos(2 pi n / N) applied to the decoded samples.
 * power_out != 0 stores |X|^2 (no log, no epsilon) instead of dB. */
int so_waterfall(const uint8_t *buf, uint64_t capacity, uint64_t start_byte,
                 const char *datatype, int cf64_decode, uint32_t nfft,
                 uint32_t hop, uint64_t n_lines, int window, double eof_fill,
                 int power_out, double *out);

/* so_waterfall with the transform chosen: SO_FFT_CM3 (= so_waterfall) or SO_FFT_EXACT */
int so_waterfall_fft(const uint8_t *buf, uint64_t capacity, uint64_t start_byte,
                     const char *datatype, int cf64_decode, uint32_t nfft,
                     uint32_t hop, uint64_t n_lines, int window, double eof_fill,
                     int power_out, int fft, double *out);

/* Number of whole lines available: floor((S - nfft)/hop) + 1 for S samples
 * from start_byte to capacity (0 when S < nfft). */
uint64_t so_count_lines(uint64_t capacity, uint64_t start_byte,
                        const char *datatype, uint32_t nfft, uint32_t hop);

/* Build-defined Welch PSD (JDSP source absent; see header comment):
 * segments s = 0..n_seg-1 start at sample s*hop, each windowed (no detrend),
 * P[k] = mean_s |FFT(w x_s)[k]|^2 / (fs * sum w^2)   (DENSITY)
 *      = mean_s |FFT(w x_s)[k]|^2 / (sum w)^2        (SPECTRUM)
 * two-sided, fftshifted so index 0 is -fs/2; freq[k] = (k - N/2) fs / N
 * (AnalysisDialogController.java:324-328 adds centerFreq to row 0).
 * psd_db != 0 returns 10 log10(P + 1e-20).  JDSP's own transform is unknown, so the estimate uses the
 * exact-twiddle transform (so_fft_forward_exact; a plain DFT for lengths that are not powers of
 * two).  Returns 0 / -1. */
int so_welch_psd(const uint8_t *buf, uint64_t capacity, uint64_t start_byte,
                 const char *datatype, int cf64_decode, uint32_t nfft,
                 uint32_t hop, uint32_t n_seg, int window, int scaling,
                 double fs, int psd_db, double *freq_out, double *psd_out);

/* MainController.java:1273-1274 -- the constant subtracted before colour
 * mapping: 10 log10(fs/N) + 20 log10(N). */
double so_display_conversion(double fs, uint32_t nfft);

/* MainController.java:1261-1291 renderSpectrogram + MainController.java:926-957
 * getColorForMagnitude, restated: waterfall is [width][nfft] dB lines; pixel (t, height-1-f)
 * takes bin (int)((double)f / height * nfft), minus so_display_conversion, normalised to
 * [min_db, max_db], through colour map 0 = "Grayscale" / 1 = "Heatmap" (javafx Color.interpolate
 * in float, 8-bit channel = round(c * 255) as PixelWriter.setColor does -- JavaFX is not in the
 * reference tree: that rounding is the published behaviour, unpinned).  out: height x width BGRA. */
void so_render_spectrogram(const double *waterfall, uint32_t width, uint32_t nfft, uint32_t height,
                           double fs, double min_db, double max_db, int colormap, uint8_t *bgra_out);

/* SURVEY 8(d) synthetic IQ (counter based, any shard can generate its own
 * span): writes n_samples IQ pairs starting at absolute sample first_sample in
 * the byte layout of `datatype` (honours _be). */
int so_synth_iq(uint8_t *out, const char *datatype, uint64_t seed,
                uint64_t first_sample, uint64_t n_samples);

/* ---- the Analysis dialog's burst chain ------------------------
 * services/ExtractDownConvertService.java:54-117, controllers/AnalysisDialogController.java:219-284 */

/* EDC:60-97 reader: planar doubles; ref_cf64_stride8 reproduces the reference's 8-byte cf64
 * stride (EDC:60-67), 0 = 16 bytes.  Unknown datatypes read as cf32 (EDC:94-96). -1: out of range. */
int so_extract_iq(const uint8_t *buf, uint64_t capacity, uint64_t start_sample, uint64_t count,
                  const char *datatype, int ref_cf64_stride8, double *re, double *im);

/* Build-defined down-converter (JDSP Resampler absent: parity unpinned; see spec_oracle.c).
 * mode 0 = boxcar ("fast", EDC:104-107), 1 = windowed-sinc low-pass (EDC:108-113).
 * so_down_convert_taps returns the tap count (h may be NULL) and the alignment index. */
uint32_t so_down_convert_taps(uint32_t down, int mode, double *h, uint32_t *centre);
uint64_t so_down_convert_len(uint64_t n, uint32_t down);
int so_down_convert(const double *re, const double *im, uint64_t n, double freq_off, uint32_t down,
                    int mode, double *re_out, double *im_out);

/* ADC:219-246: db[i] = 20 log10(EMA_alpha(hypot(re, im)))[i], n values */
void so_magnitude_trace(const double *re, const double *im, uint64_t n, double alpha, double *db);
/* ADC:256-284: EMA_alpha(wrapped phase step / 2 pi * fs) + center_freq, n - 1 values */
void so_inst_freq_trace(const double *re, const double *im, uint64_t n, double alpha, double fs,
                        double center_freq, double *out);

/* Timed driver for bench.py's cpu_baseline leg: runs so_waterfall over the
 * buffer with `threads` pthreads splitting the lines; returns seconds. */
double so_time_waterfall(const uint8_t *buf, uint64_t capacity,
                         const char *datatype, uint32_t nfft, uint32_t hop,
                         uint64_t n_lines, int window, int threads,
                         double *out_checksum);

#ifdef __cplusplus
}
#endif
#endif
