/*
 * specgpu.h -- C ABI of the MI355X-native spectrogram / PSD engine.
 *
 * This is the drop-in boundary for ONE path of GassiusODude/spectral_analyzer:
 * sample reader -> (window) -> FFT -> |X| -> 20 log10 / Welch PSD, i.e. the code
 * behind the Spectrogram and PSD views.  Every entry point names the reference
 * interface it replaces (paths relative to the reference repo,
 * src/main/java/net/kcundercover/spectral_analyzer/...).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types, no exceptions.
 *   - every call returns a spec_status; spec_last_error() gives the text.
 *   - the caller owns every buffer it passes; the library owns only what lives
 *     inside a spec_ctx (twiddle / window tables, staging and scratch memory).
 *   - a spec_ctx is bound to one GPU and one HIP stream.  It may be shared by host
 *     threads (the reference runs ExtractDownConvertService on a thread pool,
 *     AsyncExtractDownConvertService.java:27-35): every call holds the context's lock,
 *     so calls on one context are serialised; distinct contexts are independent and run
 *     concurrently.  A call leaves the calling thread's current HIP device as it found it.
 *   - there is NO CPU backend: without a usable gfx950 device spec_create fails
 *     with SPEC_EDEVICE.
 *   - device-pointer calls are asynchronous on the context's stream; host-pointer
 *     calls return after the results are in the caller's memory.
 *
 * Numerical contract (FROZEN in round 5; asserted by tests/test_gpu_parity.py and, for the drop-in, by
 * integration/java-test/SpectralServiceParityTest.java -- these are the figures a host author may rely on).
 * With M = max_k |X[k]| of a line and the reference = SpectralService.computeMagnitudes (fp64, commons-math3 3.6.1):
 *   fp32 pipeline (SPEC_OUT_*_F32 from cu8 / ci8 / ci16 / cf32 input):
 *     every bin:               | |X| - |X|_ref |  <=  4e-6 M log2(nfft)
 *     bins with |X| >= 1e-4 M: | dB - dB_ref |    <=  8.686 * 1.2e-7 * M / |X| + 2e-5      (a noise floor of at most 1.2e-7 M:
 *                                                     1.1e-3 dB on a bin of 1e-3 M, 1.0e-2 dB on a bin of 1e-4 M)
 *   fp64 pipeline (SPEC_OUT_*_F64, cf64 input, spec_compute_magnitudes -- the drop-in's double[]):
 *     bins with |X| >= 1e-9 M: | |X| - |X|_ref |  <=  max(8e-15 log2(nfft) + 5e-14, 4e-17 nfft) M
 *     bins with |X| >= 1e-5 M: | dB - dB_ref |    <=  max(1e-9, 3e-12 nfft) dB
 *     (the nfft-proportional terms are the REFERENCE transform's own rounding error -- its twiddles are running products;
 *     against the exact DFT the fp64 pipeline is within 1e-15 M and 1e-10 dB at every size, which the suite also asserts)
 *   Welch PSD: 5e-6 of the PSD's peak (fp32 pipeline), 1e-9 (fp64 entry) against the build's stated estimator (JDSP's is unknown).
 * The suite additionally holds the kernels to REGRESSION bounds at about three times what they measure today
 * (linear 5e-7 M log2 nfft; 2e-3 dB down to 1e-3 M, 4e-3 dB down to 1e-4 M on its fixed inputs): tighter than the contract, not part of it.
 */
#ifndef SPECGPU_H
#define SPECGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPECGPU_VERSION_MAJOR 0
#define SPECGPU_VERSION_MINOR 2

typedef struct spec_ctx spec_ctx;

typedef enum {
    SPEC_OK = 0,
    SPEC_EINVAL = 1,       /* bad argument (non power-of-two nfft, hop == 0, null pointer ...);
                              Java: IllegalArgumentException (commons-math3 throws
                              MathIllegalArgumentException for a bad length, SS:29) */
    SPEC_ERANGE = 2,       /* byte range outside the buffer; Java: IndexOutOfBoundsException
                              (the MappedByteBuffer getters at SS:44-57) */
    SPEC_EDEVICE = 3,      /* HIP error / no gfx950 device */
    SPEC_ENOMEM = 4,       /* host or device allocation failed */
    SPEC_EUNSUPPORTED = 5  /* valid request the build does not implement yet */
} spec_status;

/* Sample formats.  SigMF strings are matched with startsWith exactly as
 * SpectralService.java:35-38 / Global.java:67-79 do; the byte order follows
 * SigMfHelper.java:87-91 (suffix "_le" -> little endian, anything else big). */
typedef enum {
    SPEC_DT_UNKNOWN = 0,   /* decodes to 0+0i -> flat -200 dB (SS:60-63) */
    SPEC_DT_CU8 = 1,       /* (b & 0xFF - 127.5) / 128          SS:50-54 */
    SPEC_DT_CI8 = 2,       /* b / 128                           SS:55-59 */
    SPEC_DT_CI16_LE = 3,   /* getShort / 32768.0                SS:42-45 */
    SPEC_DT_CI16_BE = 4,
    SPEC_DT_CF32_LE = 5,   /* getFloat                          SS:46-49 */
    SPEC_DT_CF32_BE = 6,
    SPEC_DT_CF64_LE = 7,   /* getDouble at +0 / +8   ExtractDownConvertService.java:79-81 */
    SPEC_DT_CF64_BE = 8
} spec_dtype;

typedef enum {
    SPEC_WIN_RECT = 0,     /* the reference applies no window (SS:40-68) */
    SPEC_WIN_HANN = 1      /* periodic Hann, w[n] = 0.5 - 0.5 cos(2 pi n / N) */
} spec_window;

typedef enum {
    SPEC_OUT_DB20_F32 = 0, /* float  20 log10(|X| + 1e-10), fftshifted   (SS:76-82 in fp32) */
    SPEC_OUT_POW_F32 = 1,  /* float  |X|^2, fftshifted */
    SPEC_OUT_DB20_F64 = 2, /* double 20 log10(|X| + 1e-10): whole pipeline in fp64, as the reference */
    SPEC_OUT_POW_F64 = 3   /* double |X|^2, fp64 pipeline */
} spec_out_fmt;

typedef enum {
    SPEC_PSD_DENSITY = 0,  /* P / (fs * sum w^2)   "Power/Hz" (AnalysisDialogController.java:330) */
    SPEC_PSD_SPECTRUM = 1  /* P / (sum w)^2 */
} spec_psd_scaling;

/* spec_create flags */
#define SPEC_FLAG_REF_CF64_ZERO 0x1u /* reproduce the reference defect: computeMagnitudes has
                                        no cf64 branch, so cf64 input yields -200 dB (SS:35-63) */

#define SPEC_FLAG_REF_EDC_CF64_STRIDE8 0x4u /* reproduce the reference defect in the burst reader:
                                        extractAndDownConvert strides cf64 by 8 bytes (EDC:60-67) */

#define SPEC_FLAG_NULL_STREAM 0x2u   /* with hip_stream == NULL: launch on the device's default (null)
                                        stream instead of creating a private one */

/* ---- context ------------------------------------------------------------ */

/* Bind a context to HIP device `device` (>= 0).  `hip_stream` is a hipStream_t
 * the caller already owns (e.g. PyTorch's current stream) or NULL to let the
 * context create its own.  Replaces: the @Service singleton construction of
 * SpectralService (SS:15-23, one FastFourierTransformer per service). */
spec_status spec_create(int device, void *hip_stream, uint32_t flags, spec_ctx **out);
void spec_destroy(spec_ctx *ctx);

/* Text of the last failure on `ctx` (or of the last failed spec_create when ctx
 * is NULL).  Never NULL.  On a context shared by threads a thread sees its OWN last failure on that context if it
 * has one (whatever other threads did since), otherwise a copy of the context's last failure.  The pointer belongs to
 * the calling thread and stays valid until that thread's next call that fails on this context (own failure) or its
 * next spec_last_error() about a context it has no failure of its own on (copy); asking about one context never
 * changes what was returned for another. */
const char *spec_last_error(const spec_ctx *ctx);
const char *spec_status_string(spec_status st);

/* Block until everything queued on the context's stream has finished. */
spec_status spec_sync(spec_ctx *ctx);

/* The hipStream_t the context launches on (for callers that time with HIP events). */
void *spec_stream(const spec_ctx *ctx);

/* Tuning / testing knobs (not needed for normal use):
 *   "force_generic" = 1  route every request through the generic (scalar-math) kernels
 *   "lines_per_wg"  = n  consecutive lines (Welch: segments) walked by one sub-line / workgroup (0 = automatic)
 *   "large_chunk_mb" = m scratch size of the two-launch four-step path ("large_team" = 0 and calls of < 64 lines;
 *                     default 1024 MiB)
 *   "welch_two_pass" = 1 always sum Welch partial slabs in a second launch (default 0: batches of >= two PSDs per CU
 *                     with nfft >= 2048 are finished by the workgroup that walked the PSD's segments)
 *   "rec_pread" = 0 | 1   recordings opened by path: 0 (default) stage from the library's own mapping of the file,
 *                     1 = pread into a pinned two-slot ring (one more host copy; for files that cannot be mapped)
 *   "large_team" = 0 | 1 | 2 | 3   lines longer than the LDS holds (fp32 nfft >= 32768, fp64 nfft >= 16384): 1 (default)
 *                     = one persistent launch that keeps the intermediate in each XCD's L2, for calls of >= 64 lines,
 *                     with ONE guarded launch of a self-contained fall-back behind it (it runs only if a bounded wait of
 *                     the persistent launch timed out: a shared GPU); 0 = two-launch path only; 2 = the persistent
 *                     launch for any number of lines and no fall-back (the call then waits for the kernel and returns
 *                     SPEC_EDEVICE if one of its bounded waits timed out); 3 (tests) = the fall-back alone, as if the
 *                     persistent launch had timed out
 *   "large_team_fake_abort" = 1 (tests)  the next default-mode large-N call behaves as if its persistent launch had timed out
 *                     (a context that sees an abort takes the two-launch path from then on; setting "large_team" re-arms it)
 *   "large_ring" = 0..4  line-sized slots of intermediate per team of the persistent launch.  0 (default) = automatic:
 *                     2 for fp64 lines (the slots share the XCD's 4 MiB L2 with the input and output streams; 3 measured
 *                     5 % slower, 4 20 %), 3 for fp32 lines (slots of half the size: 3-4 % faster than 2)
 *   "large_wg" = 512 | 256 | 1024  geometry of the persistent launch.  512 (default): one 512-thread workgroup per CU,
 *                     16-bin = 128-byte output runs -- the only one the product library carries.  256 (two workgroups per
 *                     CU, one of either role on every CU) and 1024 (two 512-thread workgroups per CU at 128 registers) were
 *                     measured slower and live in the variant library lib/libspecgpu_teamvar.so
 *                     (python -m spectral_analyzer_amd.build --variant teamvar); the product returns SPEC_EUNSUPPORTED
 *   "large_single" = 1 | 0   32768-point fp32 and 16384-point fp64 lines (256 KiB; the default dispatch, "large_team" = 1):
 *                     1 (default) = ONE workgroup per line -- a radix-2 step in registers, then two half-size transforms
 *                     through the same LDS buffer, nothing handed over between workgroups (fp32: cf32 0.45, ci16 0.43 of
 *                     8 TB/s against 0.26 / 0.10 for the four-step team kernel; fp64: cf64 0.37 against 0.32, cf32 -> f64
 *                     0.30 against 0.14); 0 = the four-step paths of "large_team"
 *   "mid_single" = 2 | 1 | 0   16384-point fp32 lines through the same half-line kernel (256-thread workgroups, two per CU):
 *                     2 (default) = in the cells of the GENERATED table csrc/spec_dispatch_table.h (size x format x byte order x
 *                     hop class x window, measured cell by cell by tools/tune_dispatch.py and re-generated whenever a kernel
 *                     changes; at the end of round 5 eleven cells, most of them hops other than N, N/2, N/4 -- where the family's
 *                     kernel has no register-reuse variant -- and hop N without a window), 1 = always, 0 = never
 *   "small_single" = 2 | 1 | 0   8192-point fp32 lines through it (16 points per thread and half, three workgroups per CU):
 *                     2 (default) = by the same table (at the end of round 5 six cells: cf32 at hops other than N, N/2, N/4,
 *                     either byte order, and two ci8 cells), 1 = always, 0 = never
 *   "coop_256" = 2 | 1 | 0   256-point fp32 lines through the wave-cooperative kernel of the 64- / 128-point lines (a wave reads
 *                     the span of four consecutive lines with 16 bytes per lane into LDS and stores them the same way):
 *                     2 (default) = where it was measured faster than the family's kernel (cu8 / ci8: 1.17x ... 1.46x; cf32 / ci16
 *                     at hops other than N, N/2, N/4: 1.07x ... 1.17x), 1 = always, 0 = never
 *   "large_pair" = 1 | 0 | 2   65536-point fp32 lines (the default dispatch, "large_team" = 1): 1 (default) = a PAIR of workgroups
 *                     per line, each a single-workgroup kernel on two of the four outputs of a radix-4 step taken in registers
 *                     (two 16384-point transforms each; nothing is handed over, nothing waits): 1.04x ... 5.3x the four-step
 *                     team kernel in all 30 measured format / hop / window cases (cf32 0.25 -> 0.31 of 8 TB/s, ci16 0.09 -> 0.34);
 *                     0 = the four-step paths of "large_team".  32768-point fp64 lines have the fp64 twin of that kernel: 1 = where
 *                     it was measured faster than the team kernel (every format but little-endian cf64: 1.2x ... 1.8x), 2 = always
 *   "pair_interleave" = 1 | 0   line order of that kernel: 1 (default) = the sixteen pairs of an XCD walk one block of lines
 *                     together (pair s takes lines s, s + 16, ...: what a line shares with its neighbour is in that XCD's L2),
 *                     0 = consecutive lines per pair
 *   "debug_twiddle_bits" = k (tests)  twiddle tables built from now on lose their k low mantissa bits -- a deliberately
 *                     degraded transform for the suite's mutation test; accepted only before a context's first transform
 *   "welch_rows" = 1 (experiment library lib/libspecgpu_v2rows.so only; the product returns SPEC_EUNSUPPORTED)
 *   "multi_verify" = 0 | 1   spec_waterfall_multi / spec_welch_psd_multi with a device-resident result: 1 = every piece a
 *                     peer context sends is checksummed on its own device before it leaves and again where it landed on
 *                     the consumer's device; a difference is SPEC_EDEVICE and the message names piece, devices and the path
 *                     the copy took.  Set on ctx[0] (all peers) or on one peer.  Default 0.
 *   "stage_chunk_mb" = m chunk of the host-buffer pipeline of spec_waterfall (default 64 MiB)
 *   "render_fused" = 0 | 1   spec_waterfall_render stores only the bins the image samples (default 1; 0 = full
 *                     dB tile, then the colour kernel: the two forms give identical pixels)
 *   "readahead_lines" = n slices spec_compute_magnitudes computes per launch once its calls walk a buffer
 *                     slice by slice (default 256; 0 or 1 = every call is its own launch) */
spec_status spec_set_option(spec_ctx *ctx, const char *key, int64_t value);
/* Current value of a knob of spec_set_option, or of the read-only state "large_team_disabled" (1 once a default-mode
 * call of this context has seen its persistent large-N launch give up; "large_team" re-arms it), "multi_peer_access" (the
 * path this context's last peer copy took as a PEER of a multi call: -1 none yet, 0 staged by the runtime -- no peer
 * access to the consumer's device --, 1 direct -- hipDeviceEnablePeerAccess succeeded --, 2 consumer on the same device)
 * and "multi_verified" (pieces of this context whose checksums agreed in its last "multi_verify" call). */
spec_status spec_get_option(spec_ctx *ctx, const char *key, int64_t *value);

/* ---- datatype table ------------------------------------------------------ */

/* SigMF datatype string -> spec_dtype with the reference's startsWith rules
 * (SS:35-38, ExtractDownConvertService.java:79) and byte-order rule
 * (SigMfHelper.java:87-91). Unknown strings give SPEC_DT_UNKNOWN. */
spec_dtype spec_dtype_from_sigmf(const char *datatype);

/* Bytes per interleaved IQ pair: Global.getBytesPerSample() (Global.java:67-79);
 * SPEC_DT_UNKNOWN -> 8, the reference's fallback. */
uint32_t spec_bytes_per_sample(spec_dtype dt);

/* Whole lines available in [start_byte, n_bytes): floor((S - nfft)/hop) + 1.
 * Mirrors the range test of MainController.java:987. */
uint64_t spec_count_lines(uint64_t n_bytes, uint64_t start_byte, spec_dtype dt,
                          uint32_t nfft, uint32_t hop);

/* ---- spectrogram --------------------------------------------------------- */

/* Exact replacement for
 *   double[] SpectralService.computeMagnitudes(MappedByteBuffer buffer,
 *                                              int startByte, int nfft, String datatype)
 * (SS:33-85).  `buffer` / `capacity` are the mapped bytes (host memory; the
 * direct-buffer address on the Java side), `big_endian` is the buffer's
 * ByteOrder (SigMfHelper.java:87-91).  Writes nfft doubles to `out`
 * (index 0 = -fs/2).  The whole pipeline runs in fp64 on the GPU.
 * Errors: non power-of-two nfft -> SPEC_EINVAL; start_byte + nfft*bps >
 * capacity -> SPEC_ERANGE (the reference's getters would throw).
 * The reference calls this once per slice, slice after slice (MC:982-993).  When a call continues
 * the previous one (start_byte advanced by exactly one slice of the same buffer), the library
 * transforms the next "readahead_lines" slices in the same launch and serves the following calls
 * from that batch -- only while the caller's input bytes still compare equal to the bytes the
 * batch was computed from, so the returned line is always the transform of the current bytes. */
spec_status spec_compute_magnitudes(spec_ctx *ctx, const void *buffer, uint64_t capacity,
                                    int64_t start_byte, uint32_t nfft, const char *datatype,
                                    int big_endian, double *out);

/* Batched replacement for the slice loop of MainController.updateDisplay()
 * (MainController.java:980-999): line t covers samples [t*hop, t*hop + nfft)
 * counted from start_byte; the reference uses hop == nfft and no window.  A
 * line whose last byte would pass n_bytes is filled with eof_fill (-150.0 in
 * the reference, MC:994-998).  `iq` points at byte 0 of the buffer (device
 * memory when iq_on_device != 0, else host memory that the library stages);
 * `out` receives n_lines x nfft values of out_fmt, row-major, index 0 = -fs/2.
 * Device input must be aligned to the component size of dt at iq + start_byte. */
spec_status spec_waterfall(spec_ctx *ctx, const void *iq, int iq_on_device, uint64_t n_bytes,
                           uint64_t start_byte, spec_dtype dt, uint32_t nfft, uint32_t hop,
                           uint64_t n_lines, spec_window window, spec_out_fmt out_fmt,
                           double eof_fill, void *out, int out_on_device);

/* ---- one waterfall across several devices ------------------------ */

/* Lines are independent (MainController.java:982-993 computes every waterfall[t] from its own sample span), so
 * a long recording shards by time slice: shard r of n takes the contiguous lines [r L / n, (r + 1) L / n) and
 * reads its own sample span plus the nfft - hop halo it shares with its neighbour -- no input exchange.  The
 * reference host is ONE process (JavaFxApplication.java:31-54), so the sharding is offered here at the C ABI,
 * one context (= one device) per shard, not only to multi-process hosts (spectral_analyzer_amd/dist.py).
 * spec_shard_lines / spec_shard_span are the partition spec_waterfall_multi uses (identical to dist.shard_lines /
 * dist.shard_span), for callers that place each shard's bytes on its device themselves. */
void spec_shard_lines(uint64_t n_lines, uint32_t n_shards, uint32_t shard, uint64_t *first_line, uint64_t *end_line);
/* bytes of the recording, counted from its start_byte, that lines [first_line, end_line) read */
void spec_shard_span(uint64_t first_line, uint64_t end_line, spec_dtype dt, uint32_t nfft, uint32_t hop,
                     uint64_t *first_byte, uint64_t *n_bytes);

/* spec_waterfall with the lines sharded over n_ctx contexts, one host thread per context, all devices at work at
 * once.  L = min(n_lines, spec_count_lines(n_bytes, start_byte, ...)) lines are computed, shard r by ctx[r]; lines
 * past the end of the recording are eof_fill (MC:994-998), as in spec_waterfall.
 *   iq_on_device == 0: iq[0] is the whole recording in HOST memory (n_bytes long; what a JVM holds: the mapped
 *       file); every context stages the span of its own shard through its own pipeline.  iq[1 ...] are not read.
 *   iq_on_device != 0: iq[r] is DEVICE memory of ctx[r]'s device holding exactly shard r's span: byte 0 of iq[r]
 *       is byte start_byte + first_byte_r of the recording (spec_shard_span of spec_shard_lines(L, n_ctx, r)).
 *   out_on_device == 0: `out` is the HOST tile n_lines x nfft; every context copies its own rows there.
 *   out_on_device != 0: `out` is device memory of ctx[0]'s device (the consumer).  ctx[0] computes its rows in
 *       place; every other context computes its range in n_chunks pieces (0 = 8) and sends each finished piece
 *       straight into the consumer's rows with hipMemcpyPeerAsync on a second stream behind an event recorded
 *       after that piece's kernels -- the transfer of piece j overlaps the kernels of piece j + 1 (over xGMI every
 *       peer has its own link to the consumer) -- through a two-slot buffer, so no context holds a second tile.
 *       `out` must be idle when the call is made: the peers write it from streams of their own, ordered against
 *       nothing the caller queued earlier (on ctx[0]'s stream or elsewhere).
 * The call returns when the whole tile is in `out`.  Contexts may share a device (then they share its kernels'
 * time -- and the persistent large-N kernel of "large_team", which wants a whole device to itself, falls back to its
 * two-launch form after a bounded wait: give every context its own device); the same context may not appear twice.  On failure the first failing shard's status is returned and its
 * text is spec_last_error(ctx[0]). */
spec_status spec_waterfall_multi(spec_ctx *const *ctx, uint32_t n_ctx, const void *const *iq, int iq_on_device,
                                 uint64_t n_bytes, uint64_t start_byte, spec_dtype dt, uint32_t nfft, uint32_t hop,
                                 uint64_t n_lines, spec_window window, spec_out_fmt out_fmt, double eof_fill,
                                 void *out, int out_on_device, uint32_t n_chunks);

/* ---- recordings on disk (SURVEY 8f "next" #3) ------------------------------ */

/* Replaces the mapping step of SigMfHelper.load (sigmf/SigMfHelper.java:69-94): the reference maps at
 * most Integer.MAX_VALUE bytes of the data file (SMH:78-84) and addresses them with int offsets
 * (MainController.java:985, SS:33), so only the first 2 GiB of a recording can be shown.  A
 * spec_recording is the data file itself -- `data_path` as SigMfHelper resolves it (core:dataset or
 * the .sigmf-data sibling, SMH:49-57), `header_bytes` = captures[0] core:header_bytes (SMH:60-67) --
 * with 64-bit offsets and no size limit.  The library maps the whole file itself (64-bit length) and
 * feeds the same two-deep device pipeline as spec_waterfall from the mapping; where the mapping is
 * refused, or with spec_set_option("rec_pread", 1), it reads the slices it needs with pread into a
 * pinned two-slot ring instead.  A file that shrinks after it was opened is noticed (fstat before every call) and read
 * with pread for that call -- bytes past the new end read as zero -- instead of touching the mapping past its end. */
typedef struct spec_recording spec_recording;
spec_status spec_open_recording(spec_ctx *ctx, const char *data_path, uint64_t header_bytes,
                                spec_recording **out);
/* payload bytes after the header: max(0, file size - header_bytes) (SMH:74-76), the `capacity` of
 * the buffer the reference would have mapped, without the cap */
uint64_t spec_recording_bytes(const spec_recording *rec);
void spec_close_recording(spec_recording *rec);

/* spec_waterfall over a recording on disk; start_byte counts from the end of the header. */
spec_status spec_waterfall_recording(spec_ctx *ctx, const spec_recording *rec, uint64_t start_byte,
                                     spec_dtype dt, uint32_t nfft, uint32_t hop, uint64_t n_lines,
                                     spec_window window, spec_out_fmt out_fmt, double eof_fill, void *out,
                                     int out_on_device);

/* spec_compute_magnitudes (SS:33-85) with a 64-bit slice offset into a recording on disk. */
spec_status spec_compute_magnitudes_recording(spec_ctx *ctx, const spec_recording *rec, uint64_t start_byte,
                                              uint32_t nfft, const char *datatype, int big_endian,
                                              double *out);

/* ---- rendering (SURVEY 8f "next" #1) ------------------------------------- */

typedef enum {
    SPEC_CMAP_GRAYSCALE = 0,  /* "Grayscale"  MainController.java:939-941 */
    SPEC_CMAP_HEATMAP = 1     /* "Heatmap"    MainController.java:943-953 */
} spec_colormap;

/* Replacement for MainController.renderSpectrogram (MainController.java:1261-1291) with
 * getColorForMagnitude (MainController.java:926-957): `tile` is width x nfft fp32 dB lines
 * (SPEC_OUT_DB20_F32, one line per canvas column); pixel (x, height-1-f) shows bin
 * (int)((double)f / height * nfft) minus 10 log10(fs/nfft) + 20 log10(nfft) (MC:1273-1274),
 * mapped through [min_db, max_db] and the colour map.  Output: height x width pixels, 4 bytes
 * B,G,R,A each (= JavaFX IntArgb little endian), row 0 on top. */
spec_status spec_render_spectrogram(spec_ctx *ctx, const float *tile, int tile_on_device, uint32_t width,
                                    uint32_t nfft, uint32_t height, double fs, double min_db, double max_db,
                                    spec_colormap colormap, void *bgra_out, int out_on_device);

/* spec_waterfall (DB20_F32, fftshifted) followed by spec_render_spectrogram without the dB
 * tile ever leaving the device: one call per redraw of MainController.updateDisplay()
 * (MainController.java:962-1049); n_lines = canvas width. */
spec_status spec_waterfall_render(spec_ctx *ctx, const void *iq, int iq_on_device, uint64_t n_bytes,
                                  uint64_t start_byte, spec_dtype dt, uint32_t nfft, uint32_t hop,
                                  uint32_t n_lines, spec_window window, uint32_t height, double fs, double min_db,
                                  double max_db, spec_colormap colormap, void *bgra_out, int out_on_device);

/* ---- Welch PSD ----------------------------------------------------------- */

/* Replacement for the call
 *   PowerSpectralDensity.calculatePsdWelch(double[][] data, double fs, int nfft)
 * at AnalysisDialogController.java:308-312 (JDSP v1.3.1; source not in the
 * reference tree, so window / overlap / scaling are explicit parameters here).
 * n_psd independent PSDs are computed in one call: PSD b uses n_seg segments
 * of nfft samples, hop apart, starting at byte start_byte + b*psd_stride_bytes.
 * Any datatype; nfft is ANY integer in 1 ... 65536 -- the reference's short-burst call passes
 * the burst length itself (AnalysisDialogController.java:303-307), so lengths that are not a
 * power of two are transformed by a plain fp64 DFT (O(nfft^2) per segment; the dialog's case is
 * one segment of < 8192 samples).  cf32 / ci16 / cu8 / ci8 with a power-of-two 256 <= nfft <=
 * 16384 take the fused fast path.
 * freq_out (may be NULL): nfft doubles, (k - nfft/2) fs / nfft, HOST memory.
 * psd_out: n_psd x nfft floats (fftshifted: out[(k + nfft/2) % nfft] = P[k], also for odd nfft;
 * 10 log10(P + 1e-20) when db != 0), device memory when out_on_device != 0. */
spec_status spec_welch_psd(spec_ctx *ctx, const void *iq, int iq_on_device, uint64_t n_bytes,
                           uint64_t start_byte, uint64_t psd_stride_bytes, uint32_t n_psd,
                           spec_dtype dt, uint32_t nfft, uint32_t hop, uint32_t n_seg,
                           spec_window window, spec_psd_scaling scaling, double fs, int db,
                           double *freq_out, float *psd_out, int out_on_device);

/* The batch of spec_welch_psd over several contexts -- what a single-process host (the reference is one JVM,
 * JavaFxApplication.java:31-54) uses to spread a batch of PSDs over the GPUs of a node; SURVEY 8e, Welch side.
 * The PSDs of a batch are independent: context r computes the contiguous range spec_shard_lines(n_psd, n_ctx, r)
 * of them on its own device, one host thread per context inside the call, no exchange on the data path (the
 * segments of ONE PSD are not split: at the dialog's sizes a PSD is microseconds of work).  Argument meaning and
 * error behaviour are spec_welch_psd's; the result is what ONE context returns for the same batch -- bit for bit while
 * both take the same form of the kernel, to the rounding of the fp32 segment sums otherwise (a context sums a batch
 * of >= two PSDs per CU in one pass, smaller ones in runs of <= 64 segments; both within the 5e-6 of the parity tests).
 * iq_on_device == 0: iq[0] is the host buffer (n_bytes[0] bytes, start_byte as in spec_welch_psd); every context
 *   stages the span of its own PSDs.  iq_on_device != 0: iq[r] is memory on ctx[r]'s device whose byte 0 is the first
 *   byte of shard r's first PSD (n_bytes[r] bytes; start_byte is ignored); NULL for a shard without PSDs.
 * psd_out: n_psd x nfft floats in host memory, or (out_on_device != 0) on ctx[0]'s device -- the peers send their
 *   rows there with hipMemcpyPeerAsync.  freq_out as in spec_welch_psd.  Complete on return. */
spec_status spec_welch_psd_multi(spec_ctx *const *ctx, uint32_t n_ctx, const void *const *iq, int iq_on_device,
                                 const uint64_t *n_bytes, uint64_t start_byte, uint64_t psd_stride_bytes,
                                 uint32_t n_psd, spec_dtype dt, uint32_t nfft, uint32_t hop, uint32_t n_seg,
                                 spec_window window, spec_psd_scaling scaling, double fs, int db,
                                 double *freq_out, float *psd_out, int out_on_device);

/* The same estimate for the exact argument shape of the reference call
 *   PowerSpectralDensity.calculatePsdWelch(double[][] data, double fs, int nfft)
 * (AnalysisDialogController.java:308-312): planar doubles data[0] = I, data[1] = Q as produced by
 * ExtractDownConvertService -- in host memory, or (in_on_device != 0) still on the device where
 * spec_down_convert left them.  Every whole segment of the signal is used
 * (n_seg = (n_samples - nfft)/hop + 1); fp64 pipeline end to end, any nfft in 1 ... 65536 (the
 * dialog passes nfft = data[0].length for bursts shorter than 8192 samples, ADC:303-307).
 * freq_out / psd_out: nfft doubles each, host memory -- the two rows the reference returns. */
spec_status spec_welch_psd_planar_f64(spec_ctx *ctx, const double *re, const double *im, int in_on_device,
                                      uint64_t n_samples, uint32_t nfft, uint32_t hop, spec_window window,
                                      spec_psd_scaling scaling, double fs, int db, double *freq_out, double *psd_out);

/* ---- burst analysis (the Analysis dialog) ---------- */

typedef enum spec_downconv_mode {
    SPEC_DC_FAST = 0, /* fast == true  (EDC:104-107): boxcar of `down` taps, then decimate */
    SPEC_DC_LPF = 1   /* fast == false (EDC:108-113): low-pass FIR (8 down + 1 Hamming-sinc taps), then decimate */
} spec_downconv_mode;

/* Replaces: the reader half of
 *   double[][] ExtractDownConvertService.extractAndDownConvert(MappedByteBuffer buffer,
 *       long startSample, int count, String datatype, double freqOff, int down, boolean fast)
 * (services/ExtractDownConvertService.java:54-97): `count` IQ pairs from sample start_sample of
 * the recording, decoded to planar doubles re[count], im[count] with the reader's own table --
 * ci16 /32768, cu8 (b - 127.5)/128, ci8 /128, cf64 doubles, everything else (SPEC_DT_UNKNOWN
 * included) read as float pairs (EDC:94-96).  Bit-exact with the reference arithmetic.  Offsets
 * are 64-bit (the reference casts to int, EDC:80-96).  A span that leaves the buffer is
 * SPEC_ERANGE (the buffer getters throw IndexOutOfBoundsException). */
spec_status spec_extract_iq(spec_ctx *ctx, const void *buffer, int buffer_on_device, uint64_t capacity,
                            uint64_t start_sample, uint64_t count, spec_dtype dt, double *re, double *im,
                            int out_on_device);

/* Replaces: extractAndDownConvert as a whole (EDC:54-117): reader, frequency shift by freq_off
 * cycles per input sample (the reference hands the resampler a sample rate of 1.0, EDC:106,112),
 * filter, decimation by `down`.  Output: floor(count / down) samples, planar doubles.
 * JDSP's Resampler is not in the reference tree, so the filter is this library's own
 * specification (parity unpinned; oracle/spec_oracle.c so_down_convert):
 *   xm[n] = x[n] exp(-2 pi i frac(freq_off n));  y[m] = sum_k h[k] xm[m down + c - k]
 * with (h, c) = (boxcar 1/down of `down` taps, down - 1) or (Hamming-windowed sinc of 8 down + 1
 * taps, cut-off 0.5/down, unit DC gain, 4 down); samples outside the burst are zero. */
spec_status spec_down_convert(spec_ctx *ctx, const void *buffer, int buffer_on_device, uint64_t capacity,
                              uint64_t start_sample, uint64_t count, spec_dtype dt, double freq_off,
                              uint32_t down, spec_downconv_mode mode, double *re_out, double *im_out,
                              int out_on_device);

/* Replaces: the loop of AnalysisDialogController.updateMagnitudeChart
 * (controllers/AnalysisDialogController.java:219-246): db_out[i] = 20 log10(v[i]),
 * v[0] = hypot(re[0], im[0]), v[i] = alpha hypot(re[i], im[i]) + (1 - alpha) v[i-1].
 * n values; the reference plots only the finite ones (ADC:239-242) -- left to the caller.
 * The recurrence is evaluated as a parallel scan: equal to the serial loop to rounding. */
spec_status spec_magnitude_trace(spec_ctx *ctx, const double *re, const double *im, int in_on_device,
                                 uint64_t n, double alpha, double *db_out, int out_on_device);

/* Replaces: the loop of AnalysisDialogController.updateFrequencyChart (ADC:256-284):
 * hz_out[i-1] = center_freq + v[i], i = 1 .. n-1, where f[i] = wrap(atan2(im[i], re[i]) -
 * atan2(im[i-1], re[i-1])) / (2 pi) * fs, v[1] = f[1], v[i] = alpha f[i] + (1 - alpha) v[i-1].
 * n - 1 values. */
spec_status spec_inst_freq_trace(spec_ctx *ctx, const double *re, const double *im, int in_on_device,
                                 uint64_t n, double alpha, double fs, double center_freq,
                                 double *hz_out, int out_on_device);

/* ---- synthetic input (bench / tests) -------------------------- */

/* Fill device memory with the counter-based synthetic IQ recording: samples
 * [first_sample, first_sample + n_samples) in the byte layout of dt. */
spec_status spec_synth_iq(spec_ctx *ctx, void *dev_out, spec_dtype dt, uint64_t seed,
                          uint64_t first_sample, uint64_t n_samples);

#ifdef __cplusplus
}
#endif
#endif /* SPECGPU_H */
