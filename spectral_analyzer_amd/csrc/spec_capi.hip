// spec_capi.hip -- implementation of the C ABI declared in include/specgpu.h.
// Host code only: argument checking (the reference's error behaviour), plan /
// table cache, host<->device staging, launch sequencing.  No CPU compute path.
#include "../../include/specgpu.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "spec_fft.h"
#include "spec_internal.h"
#include "spec_dispatch_table.h"

using namespace specgpu;

struct spec_ctx {
    // every entry point that takes a context holds this lock for the whole call: a context may be shared
    // by host threads (the reference runs ExtractDownConvertService on a pool, AsyncExtractDownConvertService
    // .java:27-35,52-55), their calls are serialised.  Recursive: entry points call each other.
    std::recursive_mutex mu;
    uint64_t gen = 0;  // unique per spec_create: the per-thread error cache is keyed on it, not on the address alone
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t flags = 0;
    std::string err;
    // tables keyed by (log2n << 1 | f64) and (log2n << 4 | window << 1 | f64)
    std::map<uint32_t, void *> twiddles;
    std::map<uint32_t, void *> windows;
    std::map<uint32_t, std::pair<double, double>> window_sums;  // sum w, sum w^2
    // tables of arbitrary (non power-of-two) lengths, keyed by the length: plain-DFT Welch path
    std::map<uint32_t, void *> twiddles_n, windows_n;
    std::map<uint32_t, std::pair<double, double>> window_sums_n;
    // grow-only device scratch
    void *stage_in = nullptr;  size_t stage_in_bytes = 0;
    void *stage_out = nullptr; size_t stage_out_bytes = 0;
    void *scratch = nullptr;   size_t scratch_bytes = 0;   // large-N transposes, Welch slabs
    void *scratch2 = nullptr;  size_t scratch2_bytes = 0;  // fallback Welch: power lines + accumulator
    void *planar = nullptr;    size_t planar_bytes = 0;
    // pinned host ring of the recording reader (spec_waterfall_recording): two slots the file is pread into,
    // each copied to the device by an asynchronous (truly overlapped) transfer
    void *pin_in = nullptr;    size_t pin_in_bytes = 0;
    hipEvent_t ev_pin[2] = {nullptr, nullptr};  // slot's host->device copy has left the pinned buffer
    // spec_waterfall_multi: a peer context's two-slot buffer of finished pieces, its copy stream and the events
    // "piece computed" / "piece has left the slot"
    void *multi_buf = nullptr; size_t multi_buf_bytes = 0;
    hipStream_t s_peer = nullptr;
    hipEvent_t ev_mdone[2] = {nullptr, nullptr}, ev_mcopied[2] = {nullptr, nullptr};
    // "multi_verify": checksums of every piece on this (peer) device and of the rows it landed in on the consumer's
    // device (a stream and an array over there), and what the last multi call on this context found out
    int64_t opt_multi_verify = 0, opt_multi_verify_corrupt = 0;  // ..._corrupt (tests): damage one landed word before verifying
    void *cks_src = nullptr; size_t cks_src_bytes = 0;
    void *cks_dst = nullptr; size_t cks_dst_bytes = 0; int cks_dst_dev = -1;
    hipStream_t s_verify = nullptr; int s_verify_dev = -1;
    int64_t multi_peer_access = -1;   // -1 no peer copy yet; 0 staged by the runtime (no peer access); 1 direct (peer access
                                      // enabled); 2 consumer on the same device
    int64_t multi_verified = 0;       // pieces whose checksums agreed in the last verified call
    void *team_scratch = nullptr; size_t team_scratch_bytes = 0;  // spec_k_team.hip: ring slots of every team
    void *team_sync = nullptr;    size_t team_sync_bytes = 0;     //                   tickets, ring counters, abort word    // spec_welch_psd_planar_f64: interleaved copy of the burst
    // tuning / testing knobs (spec_set_option)
    int64_t opt_force_generic = 0, opt_lines_per_wg = 0, opt_large_chunk_mb = 1024, opt_stage_chunk_mb = 64;
    int64_t opt_large_team = 1, opt_large_ring = 0, opt_large_wg = 512, opt_large_block = 0, opt_rec_pread = 0;
    int64_t opt_large_single = 1;  // 32768-point fp32 lines in one workgroup (spec_k_v2h.hip); 0: the four-step path
    int64_t opt_pair_interleave = 1;  // the paired kernel's line order: the pairs of an XCD walk one block of lines together (spec_k_v2q.hip)
    int64_t opt_large_pair = 1;    // 65536-point fp32 lines by pairs of single-workgroup kernels (spec_k_v2q.hip); 0: the four-step team kernel
    int64_t opt_coop_256 = 2;  // "coop_256": 256-point lines through the wave-cooperative kernel of spec_k_v2n.hip: 2 where measured faster (coop_256_rule), 1 always, 0 never
    int64_t opt_small_single = 2;  // 8192-point fp32 lines through the same kernel (16 points per thread and half): 2 where measured faster, 1 always, 0 never
    int64_t opt_mid_single = 2;    // 16384-point fp32 lines through the same kernel: 2 where measured faster (run_lines), 1 always, 0 never
    int64_t opt_debug_twiddle_bits = 0;  // test hook: twiddle tables built from now on lose this many mantissa bits (tests/test_gpu_parity.py mutation test)
    int64_t opt_welch_two_pass = 0;
    int64_t opt_welch_rows = 0;  // experiment library only (build.py --variant v2rows): 16384-point Welch segments through the plan 16 x (32 x 32)
    // the persistent large-N kernel: a launch whose abort word has not been looked at yet, and the verdict once a
    // bounded wait did time out on this context (shared / partitioned GPU): it is not tried again
    bool team_check_pending = false, team_disabled = false;
    int64_t opt_team_fake_abort = 0;  // tests: the next default-mode call behaves as if its team kernel had timed out
    int n_cu = 256;
    // host-buffer pipeline (spec_waterfall): copy-in / copy-out streams and the events that order
    // them against the compute stream, created on first use
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    // read-ahead of spec_compute_magnitudes (the reference calls it once per slice, MC:982-993): when a
    // call continues the previous one (next slice of the same buffer), the following slices are computed
    // in the same launch and kept here with a copy of their input bytes; a later call is served from the
    // cache only if its input bytes still compare equal
    struct ReadAhead {
        const void *buf = nullptr; uint64_t capacity = 0, first = 0, stride = 0; int dt = -1; uint32_t nfft = 0, n = 0;
        std::vector<uint8_t> in; std::vector<double> out;
        const void *last_buf = nullptr; uint64_t last_start = 0; int last_dt = -1; uint32_t last_nfft = 0;
    } ra;
    int64_t opt_readahead_lines = 256, opt_render_fused = 1;
    // spec_waterfall_render: device table "pixel row of bin k" of the last (nfft, height) pair
    int32_t *sel_dev = nullptr; uint32_t sel_nfft = 0, sel_height = 0;
};

// a recording on disk (SigMfHelper.load's data file, SMH:49-94): descriptor, payload offset and size
struct spec_recording {
    int fd = -1;
    uint64_t header = 0;  // core:header_bytes (SMH:60-67)
    uint64_t bytes = 0;   // payload bytes after the header -- no 2 GiB cap (SMH:78-82)
    void *map = nullptr;  // the whole file mapped read-only (64-bit length), or nullptr when mmap refused
    size_t map_len = 0;
    std::string path;
};

static thread_local std::string g_create_err;
// the calling thread's own last failure (a context shared by threads has one `err` for all of them)
static thread_local std::string t_err;
static thread_local std::string t_other;  // spec_last_error: this thread's copy of ANOTHER thread's failure text
static thread_local const spec_ctx *t_err_ctx = nullptr;
static thread_local uint64_t t_err_gen = 0;
static std::atomic<uint64_t> g_ctx_gen{0};

static spec_status fail(spec_ctx *c, spec_status st, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) {
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        c->err = buf;
        t_err = buf;
        t_err_ctx = c;
        t_err_gen = c->gen;
    } else {
        g_create_err = buf;
    }
    return st;
}

// Scope of one entry point: serialises the context and makes its device current for the calling thread,
// restoring the caller's device on the way out (a context bound to device k must not leave a PyTorch
// thread on device k).
struct Enter {
    spec_ctx *c;
    int prev = -1;
    bool switched = false;
    explicit Enter(spec_ctx *ctx) : c(ctx) {
        if (!c) return;
        c->mu.lock();
        if (hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    void leave() {
        if (!c) return;
        if (switched) (void)hipSetDevice(prev);
        c->mu.unlock();
        c = nullptr;
    }
    ~Enter() { leave(); }
    Enter(const Enter &) = delete;
    Enter &operator=(const Enter &) = delete;
};

// overflow-checked a * b + c in uint64 (false on wrap): every byte count derived from caller-supplied counts
static bool mul_add_u64(uint64_t a, uint64_t b, uint64_t c, uint64_t *out) {
    uint64_t p;
    if (__builtin_mul_overflow(a, b, &p)) return false;
    return !__builtin_add_overflow(p, c, out);
}

#define HIP_TRY(c, expr)                                                                     \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(c, e__ == hipErrorOutOfMemory ? SPEC_ENOMEM : SPEC_EDEVICE, "%s: %s", #expr, \
                        hipGetErrorString(e__));                                             \
    } while (0)

static int kind_of(spec_dtype dt, uint32_t flags) {
    switch (dt) {
    case SPEC_DT_CU8: return K_CU8;
    case SPEC_DT_CI8: return K_CI8;
    case SPEC_DT_CI16_LE: case SPEC_DT_CI16_BE: return K_CI16;
    case SPEC_DT_CF32_LE: case SPEC_DT_CF32_BE: return K_CF32;
    case SPEC_DT_CF64_LE: case SPEC_DT_CF64_BE: return (flags & SPEC_FLAG_REF_CF64_ZERO) ? K_ZERO : K_CF64;
    default: return K_ZERO;
    }
}
static int is_be(spec_dtype dt) { return dt == SPEC_DT_CI16_BE || dt == SPEC_DT_CF32_BE || dt == SPEC_DT_CF64_BE; }
static uint32_t component_bytes(spec_dtype dt) {
    switch (dt) {
    case SPEC_DT_CU8: case SPEC_DT_CI8: return 1;
    case SPEC_DT_CI16_LE: case SPEC_DT_CI16_BE: return 2;
    case SPEC_DT_CF64_LE: case SPEC_DT_CF64_BE: return 8;
    default: return 4;
    }
}
static bool dtype_valid(int dt) { return dt >= SPEC_DT_UNKNOWN && dt <= SPEC_DT_CF64_BE; }
static int ilog2(uint32_t n) { int l = 0; while ((1u << l) < n) ++l; return l; }

extern "C" {

const char *spec_status_string(spec_status st) {
    switch (st) {
    case SPEC_OK: return "SPEC_OK";
    case SPEC_EINVAL: return "SPEC_EINVAL";
    case SPEC_ERANGE: return "SPEC_ERANGE";
    case SPEC_EDEVICE: return "SPEC_EDEVICE";
    case SPEC_ENOMEM: return "SPEC_ENOMEM";
    case SPEC_EUNSUPPORTED: return "SPEC_EUNSUPPORTED";
    default: return "SPEC_?";
    }
}

const char *spec_last_error(const spec_ctx *ctx) {
    if (!ctx) return g_create_err.c_str();
    // this thread's own failure on a (possibly shared) context -- of THIS context, not of a destroyed one whose
    // address a later spec_create was handed again
    if (t_err_ctx == ctx && t_err_gen == ctx->gen) return t_err.c_str();
    // another thread's failure: copied under the context's lock (that thread may be inside fail() right now), and the
    // pointer handed out is this thread's own copy
    // pointer handed out is a copy of this thread's own -- a SECOND one: asking about context B must neither overwrite
    // this thread's recorded failure on context A nor invalidate the text an earlier spec_last_error(A) returned
    spec_ctx *c = const_cast<spec_ctx *>(ctx);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    t_other = c->err;
    return t_other.c_str();
}

spec_dtype spec_dtype_from_sigmf(const char *s) {
    if (!s) return SPEC_DT_UNKNOWN;
    const size_t n = strlen(s);
    const bool le = n >= 3 && strcmp(s + n - 3, "_le") == 0;  // SigMfHelper.java:87-91
    auto sw = [&](const char *p) { return strncmp(s, p, strlen(p)) == 0; };
    if (sw("ci16")) return le ? SPEC_DT_CI16_LE : SPEC_DT_CI16_BE;  // SS:35
    if (sw("cf32")) return le ? SPEC_DT_CF32_LE : SPEC_DT_CF32_BE;  // SS:36
    if (sw("cu8")) return SPEC_DT_CU8;                              // SS:37
    if (sw("ci8")) return SPEC_DT_CI8;                              // SS:38
    if (sw("cf64")) return le ? SPEC_DT_CF64_LE : SPEC_DT_CF64_BE;  // EDC:79
    return SPEC_DT_UNKNOWN;
}

uint32_t spec_bytes_per_sample(spec_dtype dt) {  // Global.java:67-79
    switch (dt) {
    case SPEC_DT_CU8: case SPEC_DT_CI8: return 2;
    case SPEC_DT_CI16_LE: case SPEC_DT_CI16_BE: return 4;
    case SPEC_DT_CF64_LE: case SPEC_DT_CF64_BE: return 16;
    default: return 8;
    }
}

uint64_t spec_count_lines(uint64_t n_bytes, uint64_t start_byte, spec_dtype dt, uint32_t nfft, uint32_t hop) {
    if (hop == 0 || nfft == 0 || start_byte >= n_bytes) return 0;
    const uint64_t s = (n_bytes - start_byte) / spec_bytes_per_sample(dt);
    return s < nfft ? 0 : (s - nfft) / hop + 1;
}

spec_status spec_create(int device, void *hip_stream, uint32_t flags, spec_ctx **out) {
    if (!out) return fail(nullptr, SPEC_EINVAL, "spec_create: out is NULL");
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return fail(nullptr, SPEC_EDEVICE, "spec_create: no HIP device (%s); there is no CPU backend",
                    e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    if (device < 0 || device >= n_dev) return fail(nullptr, SPEC_EINVAL, "spec_create: device %d of %d", device, n_dev);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, SPEC_EDEVICE, "spec_create: device %d is %s; this library carries gfx950 code only", device,
                    prop.gcnArchName);
    spec_ctx *c = new (std::nothrow) spec_ctx;
    if (!c) return fail(nullptr, SPEC_ENOMEM, "spec_create: out of host memory");
    c->gen = ++g_ctx_gen;
    c->device = device;
    c->flags = flags;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        Enter g(c);  // the caller's current device is restored on the way out
        if (g.prev != device && !g.switched) { g.leave(); delete c; return fail(nullptr, SPEC_EDEVICE, "hipSetDevice(%d) failed", device); }
        if (hip_stream) {
            c->stream = static_cast<hipStream_t>(hip_stream);
        } else if (flags & SPEC_FLAG_NULL_STREAM) {
            c->stream = nullptr;  // the default stream: ordered with everything else the process queues on it
        } else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess) {
            c->own_stream = true;
        } else {
            g.leave(); delete c;
            return fail(nullptr, SPEC_EDEVICE, "hipStreamCreate failed");
        }
    }
    *out = c;
    return SPEC_OK;
}

void spec_destroy(spec_ctx *c) {
    if (!c) return;
    Enter g(c);
    (void)hipStreamSynchronize(c->stream);
    for (auto &kv : c->twiddles) (void)hipFree(kv.second);
    for (auto &kv : c->windows) (void)hipFree(kv.second);
    for (auto &kv : c->twiddles_n) (void)hipFree(kv.second);
    for (auto &kv : c->windows_n) (void)hipFree(kv.second);
    (void)hipFree(c->stage_in);
    (void)hipFree(c->stage_out);
    (void)hipFree(c->scratch);
    (void)hipFree(c->scratch2);
    (void)hipFree(c->planar);
    if (c->pin_in) (void)hipHostFree(c->pin_in);
    for (int i = 0; i < 2; ++i) if (c->ev_pin[i]) (void)hipEventDestroy(c->ev_pin[i]);
    (void)hipFree(c->team_scratch);
    (void)hipFree(c->team_sync);
    (void)hipFree(c->sel_dev);
    (void)hipFree(c->multi_buf);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_mdone[i]) (void)hipEventDestroy(c->ev_mdone[i]);
        if (c->ev_mcopied[i]) (void)hipEventDestroy(c->ev_mcopied[i]);
    }
    if (c->s_peer) (void)hipStreamDestroy(c->s_peer);
    if (c->s_verify) (void)hipStreamDestroy(c->s_verify);
    (void)hipFree(c->cks_src);
    (void)hipFree(c->cks_dst);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
        if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
    }
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    g.leave();
    delete c;
}

spec_status spec_sync(spec_ctx *c) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the host is synchronised anyway: did the last default-mode large-N call's persistent launch give up?  (Otherwise the
    // word is looked at lazily, at the next large-N call that finds the stream idle -- a host that queues such calls back
    // to back and only ever calls spec_sync would never learn.)
    if (c->team_check_pending && c->team_sync) {
        uint32_t aborted = 0;
        if (hipMemcpyAsync(&aborted, static_cast<uint32_t *>(c->team_sync) + large_team_abort_word(), 4, hipMemcpyDeviceToHost,
                           c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess) {
            c->team_check_pending = false;
            if (aborted) c->team_disabled = true;
        }
    }
    return SPEC_OK;
}

void *spec_stream(const spec_ctx *c) { return c ? static_cast<void *>(c->stream) : nullptr; }

// 256-point lines: the packed family's kernel (register reuse of the overlap, 16 lanes per line) or the wave-cooperative kernel
// of the 64- / 128-point lines (a wave reads the span of four lines with 16 bytes per lane into LDS and stores them the same way).
// Measured cell by cell on one box (tools/bench_coop.py, profiles/r05_coop256.txt; fractions of 8 TB/s, family -> cooperative):
//   cu8 / ci8 (2-byte samples: the family's lane groups read 32 bytes at a time)   every hop and window   1.17x ... 1.37x
//   cf32 / ci16, either byte order, at a hop the family has no register-reuse variant for (not N, N/2, N/4)   1.07x ... 1.17x
//   cf32 / ci16 at hop = N, N/2, N/4                                                                          0.85x ... 1.05x: the family stays
static bool coop_256_rule(int kind, int be, uint32_t hop, bool win) {
    if (kind == K_CU8 || kind == K_CI8) return true;
    (void)be; (void)win;  // (big-endian files have had the family's register-reuse variants since round 5: the same rule)
    return hop != 256 && hop != 128 && hop != 64;
}

spec_status spec_set_option(spec_ctx *c, const char *key, int64_t value) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!key) return fail(c, SPEC_EINVAL, "spec_set_option: null key");
    if (!strcmp(key, "force_generic")) c->opt_force_generic = value;
    else if (!strcmp(key, "lines_per_wg")) c->opt_lines_per_wg = value < 0 ? 0 : value;
    else if (!strcmp(key, "large_chunk_mb")) c->opt_large_chunk_mb = value < 1 ? 1 : value;
    else if (!strcmp(key, "stage_chunk_mb")) c->opt_stage_chunk_mb = value < 1 ? 1 : value;
    else if (!strcmp(key, "rec_pread")) c->opt_rec_pread = value != 0;
    else if (!strcmp(key, "welch_two_pass")) c->opt_welch_two_pass = value != 0;
    else if (!strcmp(key, "debug_twiddle_bits")) {
        // a deliberately degraded transform for the suite's mutation test: only on a context that has built no table yet
        if (!c->twiddles.empty()) return fail(c, SPEC_EINVAL, "debug_twiddle_bits must be set before the first transform of a context");
        c->opt_debug_twiddle_bits = value < 0 ? 0 : (value > 20 ? 20 : value);
    }
    else if (!strcmp(key, "welch_rows")) {
#ifdef SPEC_V2_ROWS
        c->opt_welch_rows = value < 0 ? 0 : (value > 2 ? 2 : value);  // 1: 16 x (32 x 32); 2: 1024 threads x 16 points (four waves per SIMD)
#else
        if (value != 0) return fail(c, SPEC_EUNSUPPORTED, "welch_rows is an experiment: build the variant library (python -m spectral_analyzer_amd.build --variant v2rows)");
#endif
    }
    else if (!strcmp(key, "large_team")) {
        c->opt_large_team = value < 0 ? 0 : (value > 3 ? 3 : value);
        c->team_disabled = false;  // setting the knob gives the persistent launch another chance
        c->team_check_pending = false;
    }
    else if (!strcmp(key, "large_ring")) c->opt_large_ring = value < 0 ? 0 : (value > 4 ? 4 : value);
    else if (!strcmp(key, "large_wg")) c->opt_large_wg = value == 256 ? 256 : (value == 1024 ? 1024 : 512);
    else if (!strcmp(key, "large_block")) c->opt_large_block = value < 0 ? 0 : (value > 65536 ? 65536 : value);
    else if (!strcmp(key, "large_team_fake_abort")) c->opt_team_fake_abort = value != 0;
    else if (!strcmp(key, "large_single")) c->opt_large_single = value != 0;
    else if (!strcmp(key, "large_pair")) c->opt_large_pair = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(key, "pair_interleave")) c->opt_pair_interleave = value != 0;
    else if (!strcmp(key, "mid_single")) c->opt_mid_single = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(key, "small_single")) c->opt_small_single = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(key, "coop_256")) c->opt_coop_256 = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(key, "multi_verify")) c->opt_multi_verify = value != 0;
    else if (!strcmp(key, "multi_verify_corrupt")) c->opt_multi_verify_corrupt = value != 0;
    else if (!strcmp(key, "render_fused")) c->opt_render_fused = value != 0;
    else if (!strcmp(key, "readahead_lines")) { c->opt_readahead_lines = value < 0 ? 0 : value; c->ra.n = 0; }
    else return fail(c, SPEC_EINVAL, "spec_set_option: unknown key '%s'", key);
    return SPEC_OK;
}

spec_status spec_get_option(spec_ctx *c, const char *key, int64_t *value) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!key || !value) return fail(c, SPEC_EINVAL, "spec_get_option: null argument");
    struct { const char *k; int64_t v; } tab[] = {
        {"force_generic", c->opt_force_generic}, {"lines_per_wg", c->opt_lines_per_wg}, {"large_chunk_mb", c->opt_large_chunk_mb},
        {"stage_chunk_mb", c->opt_stage_chunk_mb}, {"rec_pread", c->opt_rec_pread}, {"welch_two_pass", c->opt_welch_two_pass}, {"debug_twiddle_bits", c->opt_debug_twiddle_bits}, {"welch_rows", c->opt_welch_rows},
        {"large_team", c->opt_large_team}, {"large_ring", c->opt_large_ring}, {"large_wg", c->opt_large_wg}, {"large_single", c->opt_large_single}, {"large_pair", c->opt_large_pair}, {"pair_interleave", c->opt_pair_interleave}, {"mid_single", c->opt_mid_single}, {"small_single", c->opt_small_single}, {"coop_256", c->opt_coop_256},
        {"large_block", c->opt_large_block}, {"render_fused", c->opt_render_fused}, {"readahead_lines", c->opt_readahead_lines},
        {"large_team_fake_abort", c->opt_team_fake_abort}, {"large_team_disabled", c->team_disabled ? 1 : 0},
        {"multi_verify", c->opt_multi_verify}, {"multi_verify_corrupt", c->opt_multi_verify_corrupt}, {"multi_peer_access", c->multi_peer_access}, {"multi_verified", c->multi_verified},
    };
    for (const auto &e : tab)
        if (!strcmp(key, e.k)) { *value = e.v; return SPEC_OK; }
    return fail(c, SPEC_EINVAL, "spec_get_option: unknown key '%s'", key);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------
// W_N^m = exp(-2 pi i m / N), evaluated in long double, rounded once.
static spec_status get_twiddles(spec_ctx *c, int log2n, bool f64, const void **out) {
    const uint32_t key = ((uint32_t)log2n << 1) | (f64 ? 1u : 0u);
    auto it = c->twiddles.find(key);
    if (it != c->twiddles.end()) { *out = it->second; return SPEC_OK; }
    const size_t n = (size_t)1 << log2n, esz = f64 ? 16 : 8;
    std::vector<unsigned char> host(n * esz);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (size_t m = 0; m < n; ++m) {
        // exact at the eight octant points, symmetric elsewhere
        const long double a = -two_pi * (long double)m / (long double)n;
        const long double cr = cosl(a), ci = sinl(a);
        if (f64) { double *p = reinterpret_cast<double *>(host.data()) + 2 * m; p[0] = (double)cr; p[1] = (double)ci; }
        else { float *p = reinterpret_cast<float *>(host.data()) + 2 * m; p[0] = (float)cr; p[1] = (float)ci; }
    }
    if (c->opt_debug_twiddle_bits > 0) {  // "debug_twiddle_bits": drop the low mantissa bits of every entry (truncation)
        const int k = (int)c->opt_debug_twiddle_bits;
        if (f64) {
            uint64_t *q = reinterpret_cast<uint64_t *>(host.data());
            for (size_t i = 0; i < 2 * n; ++i) q[i] &= ~((1ull << (k + 29)) - 1);  // (a double carries 29 bits more than a float)
        } else {
            uint32_t *q = reinterpret_cast<uint32_t *>(host.data());
            for (size_t i = 0; i < 2 * n; ++i) q[i] &= ~((1u << k) - 1);
        }
    }
    void *dev = nullptr;
    HIP_TRY(c, hipMalloc(&dev, n * esz));
    hipError_t e = hipMemcpyAsync(dev, host.data(), n * esz, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // host vector dies at return
    if (e != hipSuccess) { (void)hipFree(dev); return fail(c, SPEC_EDEVICE, "twiddle upload: %s", hipGetErrorString(e)); }
    c->twiddles[key] = dev;
    *out = dev;
    return SPEC_OK;
}

static spec_status get_window(spec_ctx *c, int log2n, bool f64, spec_window w, const void **out, double *s1,
                              double *s2, bool table_for_rect = false) {
    const size_t n = (size_t)1 << log2n;
    if (w == SPEC_WIN_RECT && !table_for_rect) {
        *out = nullptr;
        if (s1) *s1 = (double)n;
        if (s2) *s2 = (double)n;
        return SPEC_OK;
    }
    const uint32_t key = ((uint32_t)log2n << 4) | ((uint32_t)w << 1) | (f64 ? 1u : 0u);
    auto it = c->windows.find(key);
    if (it == c->windows.end()) {
        const size_t esz = f64 ? 8 : 4;
        std::vector<unsigned char> host(n * esz);
        const long double two_pi = 6.283185307179586476925286766559005768L;
        double a1 = 0, a2 = 0;
        for (size_t i = 0; i < n; ++i) {
            const double v = w == SPEC_WIN_RECT ? 1.0 : (double)(0.5L - 0.5L * cosl(two_pi * (long double)i / (long double)n));
            a1 += v; a2 += v * v;
            if (f64) reinterpret_cast<double *>(host.data())[i] = v;
            else reinterpret_cast<float *>(host.data())[i] = (float)v;
        }
        void *dev = nullptr;
        HIP_TRY(c, hipMalloc(&dev, n * esz));
        hipError_t e = hipMemcpyAsync(dev, host.data(), n * esz, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipFree(dev); return fail(c, SPEC_EDEVICE, "window upload: %s", hipGetErrorString(e)); }
        c->windows[key] = dev;
        c->window_sums[key] = {a1, a2};
        it = c->windows.find(key);
    }
    *out = it->second;
    if (s1) *s1 = c->window_sums[key].first;
    if (s2) *s2 = c->window_sums[key].second;
    return SPEC_OK;
}

// Tables of an arbitrary length N (the plain-DFT Welch path): W_N^m as cx<double>[N] and the periodic Hann
// window as double[N] (nullptr for the rectangular one), long double on the host, rounded once.
static spec_status get_tables_n(spec_ctx *c, uint32_t n, spec_window w, const void **tw, const void **win, double *s1,
                                double *s2) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    auto upload = [&](const std::vector<double> &host, void **dev) -> spec_status {
        HIP_TRY(c, hipMalloc(dev, host.size() * sizeof(double)));
        hipError_t e = hipMemcpyAsync(*dev, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // host vector dies at return
        if (e != hipSuccess) { (void)hipFree(*dev); *dev = nullptr; return fail(c, SPEC_EDEVICE, "table upload: %s", hipGetErrorString(e)); }
        return SPEC_OK;
    };
    auto it = c->twiddles_n.find(n);
    if (it == c->twiddles_n.end()) {
        std::vector<double> host;
        try { host.resize(2 * (size_t)n); } catch (...) { return fail(c, SPEC_ENOMEM, "out of host memory"); }
        for (uint32_t m = 0; m < n; ++m) {
            const long double a = -two_pi * (long double)m / (long double)n;
            host[2 * m] = (double)cosl(a);
            host[2 * m + 1] = (double)sinl(a);
        }
        void *dev = nullptr;
        spec_status st = upload(host, &dev);
        if (st != SPEC_OK) return st;
        it = c->twiddles_n.emplace(n, dev).first;
    }
    *tw = it->second;
    if (w == SPEC_WIN_RECT) { *win = nullptr; *s1 = *s2 = (double)n; return SPEC_OK; }
    auto iw = c->windows_n.find(n);
    if (iw == c->windows_n.end()) {
        std::vector<double> host;
        try { host.resize(n); } catch (...) { return fail(c, SPEC_ENOMEM, "out of host memory"); }
        double a1 = 0, a2 = 0;
        for (uint32_t i = 0; i < n; ++i) {
            host[i] = (double)(0.5L - 0.5L * cosl(two_pi * (long double)i / (long double)n));
            a1 += host[i]; a2 += host[i] * host[i];
        }
        void *dev = nullptr;
        spec_status st = upload(host, &dev);
        if (st != SPEC_OK) return st;
        iw = c->windows_n.emplace(n, dev).first;
        c->window_sums_n[n] = {a1, a2};
    }
    *win = iw->second;
    *s1 = c->window_sums_n[n].first;
    *s2 = c->window_sums_n[n].second;
    return SPEC_OK;
}

static spec_status grow(spec_ctx *c, void **buf, size_t *have, size_t need) {
    if (*have >= need) return SPEC_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *have = 0; }
    HIP_TRY(c, hipMalloc(buf, need));
    *have = need;
    return SPEC_OK;
}

// choose how many consecutive lines one workgroup walks
static uint32_t pick_lines_per_wg(uint64_t n_lines, int lpw) {
    const uint64_t target_wgs = 256ull * 48;  // ~10 rounds of resident workgroups: small tail, some L2 reuse
    uint64_t per = (n_lines + target_wgs - 1) / target_wgs;
    per = (per + lpw - 1) / lpw * lpw;
    if (per < (uint64_t)lpw) per = lpw;
    const uint64_t cap = 32ull * lpw;
    if (per > cap) per = cap;
    return (uint32_t)per;
}

// launch the spectrogram kernel(s) for lines that lie fully inside the buffer
// d_sel != nullptr: "selected bins" mode -- rows of `row` floats, bin k at column d_sel[k] (callers check
// v2_sel_applicable first)
static spec_status run_lines(spec_ctx *c, const uint8_t *d_first, spec_dtype dt, int log2n, uint32_t hop,
                             uint64_t n_lines, spec_window window, spec_out_fmt fmt, void *d_out,
                             const int32_t *d_sel = nullptr, uint32_t row = 0) {
    if (n_lines == 0) return SPEC_OK;
    const bool f64 = fmt >= SPEC_OUT_DB20_F64 || dt == SPEC_DT_CF64_LE || dt == SPEC_DT_CF64_BE;
    int l1 = 0, l2 = 0;
    const bool large = !plan_supported(log2n, f64) && large_split(log2n, f64, &l1, &l2);
    if (!plan_supported(log2n, f64) && !large)
        return fail(c, SPEC_EUNSUPPORTED, "nfft = 2^%d is not supported in %s", log2n, f64 ? "fp64" : "fp32");
    WfArgs a{};
    a.iq = d_first;
    a.hop = hop;
    a.bps = spec_bytes_per_sample(dt);
    a.kind = kind_of(dt, c->flags);
    a.be = is_be(dt);
    a.out_fmt = (int)fmt;
    if (d_sel && (f64 || fmt != SPEC_OUT_DB20_F32 || c->opt_force_generic || !v2_sel_applicable(log2n, a.kind, a.be, n_lines, hop)))
        return fail(c, SPEC_EUNSUPPORTED, "selected-bin output is not available for this configuration");
    // the four-step path takes its inter-step twiddles from the fp64 W_N table whatever the precision
    spec_status st = get_twiddles(c, log2n, large ? true : f64, &a.tw);
    if (st != SPEC_OK) return st;
    st = get_window(c, log2n, f64, window, &a.win, nullptr, nullptr);
    if (st != SPEC_OK) return st;
    a.win_hann = window == SPEC_WIN_HANN;
    const uint64_t out_esz = fmt >= SPEC_OUT_DB20_F64 ? 8 : 4, nfft = 1ull << log2n;
    // 8192- and 16384-point lines have two kernels: the family's (spec_v2.h) and the half-line kernel of spec_k_v2h.hip (smaller
    // workgroups, two or three per CU).  Which one is faster is MEASURED cell by cell -- format x byte order x hop class x window --
    // by tools/tune_dispatch.py, which generates spec_dispatch_table.h; "mid_single" (16384) / "small_single" (8192) = 2 (default)
    // follow the table, 1 always takes the half-line kernel, 0 never.  (Rounds 3-4 transcribed the rule by hand from such tables.)
    bool mid_single = false;
    if ((log2n == 14 || log2n == 13) && !f64 && !d_sel && !c->opt_force_generic) {
        const int64_t knob = log2n == 14 ? c->opt_mid_single : c->opt_small_single;
        mid_single = knob == 1 || (knob == 2 && dispatch_half_line(log2n, a.kind, a.be != 0, hop, window != SPEC_WIN_RECT));
    }
    if (((large && c->opt_large_single && c->opt_large_team == 1) || mid_single) && !f64 && !d_sel &&
        v2h_applicable(log2n, a.kind, a.out_fmt, n_lines, hop)) {
        // 32768-point fp32 lines: ONE workgroup per line -- a radix-2 step in registers, then two 16384-point transforms of
        // the packed family through the same LDS buffer (spec_k_v2h.hip); no hand-off between workgroups.  The default
        // dispatch only: a caller who sets "large_team" to 0 / 2 / 3 asks for one of the four-step paths by name.
        const void *tw_half = nullptr;
        if ((st = get_twiddles(c, log2n, false, &a.tw)) != SPEC_OK) return st;
        if ((st = get_twiddles(c, log2n - 1, false, &tw_half)) != SPEC_OK) return st;
        const void *tw_full64 = nullptr;  // the window's cosine is formed in fp64
        if (a.win && (st = get_twiddles(c, log2n, true, &tw_full64)) != SPEC_OK) return st;
        uint64_t done = 0;
        while (done < n_lines) {
            const uint64_t rem = n_lines - done;
            // one workgroup per CU at a time and a set-up (twiddle tables, first line not prefetched) worth about a line: long
            // runs, two workgroups per CU when the recording is short (measured: runs of 32 against runs of 8, 44.6 % / 42.5 %)
            const uint64_t wgs_wanted = (uint64_t)c->n_cu * (log2n == 15 ? 2 : log2n == 14 ? 4 : 8);
            uint64_t run = c->opt_lines_per_wg > 0 ? (uint64_t)c->opt_lines_per_wg : (rem + wgs_wanted - 1) / wgs_wanted;
            if (run < 1) run = 1;
            if (run > 32) run = 32;  // (runs of 64: 42.8 %, of 32: 43.8 % on the same box)
            while (run > 1 && run * ((uint64_t)hop * a.bps + nfft * out_esz) >= (1ull << 31)) run /= 2;  // 32-bit offsets in a span
            a.n_lines = rem < 0x7FFFFFFFull ? rem : 0x7FFFFFFFull;
            a.iq = d_first + done * (uint64_t)hop * a.bps;
            a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
            hipError_t e = launch_v2h_spectro(a, log2n, tw_half, tw_full64, (uint32_t)run, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "32768-point launch: %s", hipGetErrorString(e));
            done += a.n_lines;
        }
        return SPEC_OK;
    }
    if (large && !f64 && c->opt_large_pair && c->opt_large_team == 1 && !d_sel && v2q_applicable(log2n, a.kind, a.out_fmt, n_lines, hop)) {
        // 65536-point fp32 lines: a PAIR of workgroups per line, each a single-workgroup kernel on a radix-4 step's even or odd
        // outputs (spec_k_v2q.hip); no hand-off, no waiting.  The default dispatch only, as above.
        const void *tw_q = nullptr, *tw_full64 = nullptr;
        if ((st = get_twiddles(c, log2n, false, &a.tw)) != SPEC_OK) return st;
        if ((st = get_twiddles(c, log2n - 2, false, &tw_q)) != SPEC_OK) return st;
        if (a.win && (st = get_twiddles(c, log2n, true, &tw_full64)) != SPEC_OK) return st;
        uint64_t done = 0;
        while (done < n_lines) {
            const uint64_t rem = n_lines - done;
            // one workgroup per CU: n_cu / 2 pairs at a time; long runs (the set-up is worth about a line), two rounds when short
            const uint64_t pairs_wanted = (uint64_t)c->n_cu;
            uint64_t run = c->opt_lines_per_wg > 0 ? (uint64_t)c->opt_lines_per_wg : (rem + pairs_wanted - 1) / pairs_wanted;
            if (run < 1) run = 1;
            if (run > 32) run = 32;
            const uint64_t ilv = c->opt_pair_interleave ? 16 : 1;  // a pair's lines are `ilv` apart
            while (run > 1 && run * ilv * ((uint64_t)hop * a.bps + nfft * out_esz) >= (1ull << 31)) run /= 2;  // 32-bit offsets in a span
            a.n_lines = rem < 0x7FFFFFFFull ? rem : 0x7FFFFFFFull;
            a.iq = d_first + done * (uint64_t)hop * a.bps;
            a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
            hipError_t e = launch_v2q_spectro(a, tw_q, tw_full64, (uint32_t)run, (int)c->opt_pair_interleave, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "65536-point launch: %s", hipGetErrorString(e));
            done += a.n_lines;
        }
        return SPEC_OK;
    }
    // 32768-point fp64 lines: the fp64 twin of the paired kernel above (spec_k_v3h.hip v3q_kernel: two 8192-point fp64 transforms per
    // workgroup).  Measured against the team kernel (tools/bench_v3q.py, profiles/r05_v3q.txt): 1.2x ... 1.8x faster for every format
    // but little-endian cf64 -- the team kernel's best case (0.30 of 8 TB/s against 0.22: the pair kernel's cf64 variants spill ~80
    // registers) -- which "large_pair" = 1 (default) therefore leaves where it was; 2 = always the pair.
    if (large && f64 && c->opt_large_team == 1 && !d_sel && v3q_applicable(log2n, a.kind, n_lines, hop) &&
        (c->opt_large_pair == 2 || (c->opt_large_pair == 1 && !(a.kind == K_CF64 && !a.be)))) {
        const void *tw_q = nullptr;
        if ((st = get_twiddles(c, log2n - 2, true, &tw_q)) != SPEC_OK) return st;
        uint64_t done = 0;
        while (done < n_lines) {
            const uint64_t rem = n_lines - done;
            uint64_t run = c->opt_lines_per_wg > 0 ? (uint64_t)c->opt_lines_per_wg : (rem + (uint64_t)c->n_cu - 1) / (uint64_t)c->n_cu;
            if (run < 1) run = 1;
            if (run > 32) run = 32;
            const uint64_t ilv = c->opt_pair_interleave ? 16 : 1;
            while (run > 1 && run * ilv * ((uint64_t)hop * a.bps + nfft * out_esz) >= (1ull << 31)) run /= 2;  // 32-bit offsets in a span
            a.n_lines = rem < 0x7FFFFFFFull ? rem : 0x7FFFFFFFull;
            a.iq = d_first + done * (uint64_t)hop * a.bps;
            a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
            hipError_t e = launch_v3q_spectro(a, tw_q, (uint32_t)run, (int)c->opt_pair_interleave, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "32768-point fp64 launch: %s", hipGetErrorString(e));
            done += a.n_lines;
        }
        return SPEC_OK;
    }
    if (large && f64 && c->opt_large_single && c->opt_large_team == 1 && !d_sel && v3h_applicable(log2n, a.kind, n_lines, hop)) {
        // 16384-point fp64 lines (256 KiB, as the 32768-point fp32 line above): one workgroup per line, a radix-2 step in
        // registers and two 8192-point transforms of the fp64 family (spec_k_v3h.hip)
        const void *tw_half = nullptr;
        if ((st = get_twiddles(c, log2n - 1, true, &tw_half)) != SPEC_OK) return st;
        uint64_t done = 0;
        while (done < n_lines) {
            const uint64_t rem = n_lines - done;
            const uint64_t wgs_wanted = (uint64_t)c->n_cu * 2;
            uint64_t run = c->opt_lines_per_wg > 0 ? (uint64_t)c->opt_lines_per_wg : (rem + wgs_wanted - 1) / wgs_wanted;
            if (run < 1) run = 1;
            if (run > 32) run = 32;
            while (run > 1 && run * ((uint64_t)hop * a.bps + nfft * out_esz) >= (1ull << 31)) run /= 2;  // 32-bit offsets in a span
            a.n_lines = rem < 0x7FFFFFFFull ? rem : 0x7FFFFFFFull;
            a.iq = d_first + done * (uint64_t)hop * a.bps;
            a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
            hipError_t e = launch_v3h_spectro(a, tw_half, (uint32_t)run, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "16384-point fp64 launch: %s", hipGetErrorString(e));
            done += a.n_lines;
        }
        return SPEC_OK;
    }
    if (large) {
        const void *tw1 = nullptr, *tw2 = nullptr;
        if ((st = get_twiddles(c, l1, f64, &tw1)) != SPEC_OK) return st;
        if ((st = get_twiddles(c, l2, f64, &tw2)) != SPEC_OK) return st;
        const size_t per_line = large_scratch_bytes_per_line(log2n, f64);
        // One persistent launch with the intermediate kept in each XCD's L2 (spec_k_team.hip); behind it, unless
        // "large_team" = 2, ONE guarded launch of the self-contained fall-back (spec_k_large.hip large_solo_kernel) that
        // runs only if a bounded wait of the team kernel timed out (grid not co-resident: a shared GPU).
        // "large_team" = 3 (tests): the team kernel is skipped and the abort word set, so the fall-back does the work.
        if (c->team_check_pending && c->team_sync && hipStreamQuery(c->stream) == hipSuccess) {
            // the previous default-mode call has finished: did its team kernel give up?  (Looked at lazily -- the call
            // itself stays asynchronous; its result was produced by the guarded fall-back either way.)  A context on a
            // GPU where the persistent grid is not co-resident would otherwise spin to the 2 s limit in every call.
            uint32_t aborted = 0;  // on the context's own (idle) stream: no null-stream synchronisation with the host's other streams
            if (hipMemcpyAsync(&aborted, static_cast<uint32_t *>(c->team_sync) + large_team_abort_word(), 4, hipMemcpyDeviceToHost,
                               c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess) {
                c->team_check_pending = false;
                if (aborted) c->team_disabled = true;
            }
        }
        const bool team = c->opt_large_team >= 2 || (c->opt_large_team == 1 && n_lines >= 64 && !c->team_disabled);
        if (team) {
            uint32_t teams_max = 0;
            // line-sized slots of intermediate per team: they share the XCD's 4 MiB L2 with the input and output streams.
            // fp64 lines (1 MiB slots at 65536 points): two -- a third is written back before it is reused (measured 5 %
            // slower); fp32 lines (half the size): three fit, and the extra slack between the two sides is worth 3-4 %
            const uint32_t ring = c->opt_large_ring > 0 ? (uint32_t)c->opt_large_ring : (f64 ? 2u : 3u);
            hipError_t e = launch_spectro_team(a, log2n, f64, tw1, tw2, nullptr, ring, nullptr, c->n_cu, &teams_max, true, c->stream,
                                               (int)c->opt_large_wg, (uint32_t)c->opt_large_block);
            if (e == hipErrorNotSupported)
                return fail(c, SPEC_EUNSUPPORTED, "large_wg = %lld is an experiment geometry: build the variant library "
                            "(python -m spectral_analyzer_amd.build --variant teamvar)", (long long)c->opt_large_wg);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "large-N team query: %s", hipGetErrorString(e));
            if ((st = grow(c, &c->team_scratch, &c->team_scratch_bytes, (size_t)(teams_max ? teams_max : 1) * ring * per_line)) != SPEC_OK) return st;
            if ((st = grow(c, &c->team_sync, &c->team_sync_bytes, large_team_sync_bytes())) != SPEC_OK) return st;
            // the fall-back's intermediates: one line per workgroup (round 2 kept a 1 GiB chunk scratch for the same purpose)
            // (a quarter of the CUs: the fall-back almost never runs, and its scratch is allocated with the first large-N call
            // whether it does or not -- 64 MiB instead of 256 MiB for 65536-point fp64 lines)
            // ... unless the caller ASKED for the fall-back ("large_team" = 3): then it does all the work and gets every CU
            const uint32_t solo_grid = c->opt_large_team == 3 ? (uint32_t)c->n_cu : (uint32_t)(c->n_cu >= 4 ? c->n_cu / 4 : 1);
            if (c->opt_large_team != 2 && (st = grow(c, &c->scratch, &c->scratch_bytes, (size_t)solo_grid * per_line)) != SPEC_OK) return st;
            for (uint64_t done = 0; done < n_lines;) {
                const uint64_t nl = n_lines - done < 0x40000000ull ? n_lines - done : 0x40000000ull;  // 32-bit line index
                uint32_t *sync = static_cast<uint32_t *>(c->team_sync);
                HIP_TRY(c, hipMemsetAsync(sync, 0, large_team_sync_bytes(), c->stream));
                a.n_lines = nl;
                a.iq = d_first + done * (uint64_t)hop * a.bps;
                a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
                const bool fake_abort = c->opt_large_team == 3 || (c->opt_large_team == 1 && c->opt_team_fake_abort);
                c->opt_team_fake_abort = 0;
                if (fake_abort) {
                    HIP_TRY(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(sync + large_team_abort_word()), 1, 1, c->stream));
                } else {
                    e = launch_spectro_team(a, log2n, f64, tw1, tw2, c->team_scratch, ring, sync, c->n_cu,
                                            &teams_max, false, c->stream, (int)c->opt_large_wg, (uint32_t)c->opt_large_block);
                    if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "large-N team launch: %s", hipGetErrorString(e));
                }
#ifdef SPEC_TEAM_PROF
                if (const char *path = getenv("SPEC_TEAM_PROF_OUT")) {  // development aid: dump the per-workgroup wait cycles
                    std::vector<unsigned long long> pf(16 * 1024 + 64 * 32 * 8);  // per-workgroup words + the event trace
                    HIP_TRY(c, hipMemcpyAsync(pf.data(), static_cast<uint8_t *>(c->team_sync) + large_team_prof_offset_bytes(), pf.size() * 8,
                                              hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    if (FILE *f = fopen(path, "wb")) { fwrite(pf.data(), 8, pf.size(), f); fclose(f); }
                }
#endif
                if (c->opt_large_team == 2) {  // no fall-back wanted: a timed-out wait is this call's error
                    uint32_t aborted = 0;
                    HIP_TRY(c, hipMemcpyAsync(&aborted, sync + large_team_abort_word(), 4, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    if (aborted) return fail(c, SPEC_EDEVICE, "large-N team kernel: a bounded wait timed out (grid not co-resident)");
                } else {  // one guarded launch: returns at once unless the abort word is set
                    c->team_check_pending = c->opt_large_team == 1;
                    e = launch_spectro_large_solo(a, log2n, f64, tw1, tw2, c->scratch, solo_grid, c->stream, sync + large_team_abort_word());
                    if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "large-N fall-back launch: %s", hipGetErrorString(e));
                }
                done += nl;
            }
            return SPEC_OK;
        }
        // four-step path as two launches per chunk of lines ("large_team" = 0, and calls of fewer than 64 lines)
        const uint32_t *run_if = nullptr;
        uint64_t chunk = ((uint64_t)c->opt_large_chunk_mb << 20) / per_line;
        if (chunk == 0) chunk = 1;
        if (chunk > n_lines) chunk = n_lines;
        if ((st = grow(c, &c->scratch, &c->scratch_bytes, (size_t)chunk * per_line)) != SPEC_OK) return st;
        for (uint64_t done = 0; done < n_lines; done += chunk) {
            a.n_lines = n_lines - done < chunk ? n_lines - done : chunk;
            a.iq = d_first + done * (uint64_t)hop * a.bps;
            a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
            hipError_t e = launch_spectro_large(a, log2n, f64, tw1, tw2, c->scratch, c->stream, run_if);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "large-N launch: %s", hipGetErrorString(e));
        }
        return SPEC_OK;
    }
    if (!f64 && !d_sel && !c->opt_force_generic && v2n_applicable(log2n, a.kind, a.out_fmt, n_lines, hop, d_first,
                                                                    c->opt_coop_256 == 1 || (c->opt_coop_256 == 2 && coop_256_rule(a.kind, a.be, hop, a.win != nullptr)) ? 8 : 7)) {
        // 64 / 128 points: wave-cooperative I/O around the packed FFT core (spec_k_v2n.hip)
        uint64_t done = 0;
        while (done < n_lines) {
            const uint64_t rem = n_lines - done;
            a.n_lines = rem < 0x40000000ull ? rem : 0x40000000ull;
            a.iq = d_first + done * (uint64_t)hop * a.bps;
            a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
            a.lines_per_wg = c->opt_lines_per_wg > 0 ? (uint32_t)c->opt_lines_per_wg : 0;  // here: consecutive BLOCKS per wave and chunk
            hipError_t e = launch_v2n_spectro(a, log2n, c->n_cu, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "spectrogram launch: %s", hipGetErrorString(e));
            done += a.n_lines;
        }
        return SPEC_OK;
    }
    const int lpw = plan_lpw(log2n);
    const bool v2 = !f64 && !c->opt_force_generic &&
                    v2_applicable(log2n, a.kind, a.be, a.out_fmt, n_lines, hop);
    if (f64 && !d_sel && !c->opt_force_generic && a.kind != K_ZERO && v3d_applicable(log2n, a.kind, n_lines, hop)) {
        // fp64 member of the packed family (strict-parity pipeline), same launch geometry as below
        const uint64_t sub = log2n == 13 ? 1 : (uint64_t)v2_lpw(log2n);  // fp64 8192 points: one line per workgroup
        uint64_t done = 0;
        while (done < n_lines) {
            const uint64_t rem = n_lines - done;
            uint64_t run = c->opt_lines_per_wg > 0 ? (uint64_t)c->opt_lines_per_wg
                                                    : (rem + sub * c->n_cu * 8 - 1) / (sub * c->n_cu * 8);
            if (run < 1) run = 1;
            if (run > 32) run = 32;
            // (as for the fp32 family below: without an overlap short runs keep neighbouring workgroups on neighbouring lines --
            // cf64 lines of 1024 / 2048 points +10 % / +9 % with runs of 4, 4096 points +2 % with 8; cf32 / ci16 -> f64 lines are
            // bound by their arithmetic and level: profiles/r05_run_len.txt)
            if (c->opt_lines_per_wg <= 0 && hop >= nfft && log2n >= 10 && log2n <= 12) {
                const uint64_t cap = log2n == 12 ? 8 : 4;
                if (run > cap) run = cap;
            }
            while (run > 1 && sub * run * ((uint64_t)hop * a.bps + nfft * out_esz) >= (1ull << 31)) run /= 2;
            a.n_lines = rem < 0x7FFFFFFFull ? rem : 0x7FFFFFFFull;
            a.iq = d_first + done * (uint64_t)hop * a.bps;
            a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
            hipError_t e = launch_v3d_spectro(a, log2n, (uint32_t)run, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "fp64 spectrogram launch: %s", hipGetErrorString(e));
            done += a.n_lines;
        }
        return SPEC_OK;
    }
    if (v2) {
        // packed-fp32 family: every sub-line (T threads) walks its own run of consecutive lines
        const uint64_t sub = (uint64_t)v2_lpw(log2n);
        uint64_t done = 0;
        while (done < n_lines) {
            const uint64_t rem = n_lines - done;
            uint64_t run = c->opt_lines_per_wg > 0 ? (uint64_t)c->opt_lines_per_wg
                                                    : (rem + sub * c->n_cu * 8 - 1) / (sub * c->n_cu * 8);
            if (run < 1) run = 1;
            if (run > 32) run = 32;
            // Without an overlap to keep in registers (the reference's own hop = nfft, MC:984-985) a long run buys nothing, and short
            // ones let neighbouring workgroups work on neighbouring lines: 1024 / 2048 / 4096 points +2 ... +9 % with runs of 8 / 4 / 4
            // (cf32 and ci16, two sweeps on one box: tools/bench_run_len.py, profiles/r05_run_len.txt; 256 / 512 points and the
            // 32-point-per-thread sizes are level or mixed and keep the long runs)
            if (c->opt_lines_per_wg <= 0 && hop >= nfft && log2n >= 10 && log2n <= 12) {
                const uint64_t cap = log2n == 10 ? 8 : 4;
                if (run > cap) run = cap;
            }
            // (with 50 % overlap long runs are right -- each run start re-reads half a line -- except for cf32 lines of 256 and 1024
            // points: runs of 16 there, +6 ... +12 % and +3 ... +5 %; ci16 and the other sizes are level or lose: same log)
            if (c->opt_lines_per_wg <= 0 && hop * 2 == (uint32_t)nfft && a.kind == K_CF32 && (log2n == 8 || log2n == 10) && run > 16) run = 16;
            // (and cf32 lines of 512 points at hop = nfft: runs of 16, +1.5 ... +6 % at 2^26 ... 2^30 samples; 256 points and ci16 are mixed)
            if (c->opt_lines_per_wg <= 0 && hop >= nfft && a.kind == K_CF32 && log2n == 9 && run > 16) run = 16;
            // 32-bit byte offsets inside a workgroup's span (input and output side)
            while (run > 1 && sub * run * ((uint64_t)hop * a.bps + nfft * out_esz) >= (1ull << 31)) run /= 2;
            const uint64_t max_lines = 0x7FFFFFFFull;  // 32-bit line index inside one launch
            a.n_lines = rem < max_lines ? rem : max_lines;
            a.iq = d_first + done * (uint64_t)hop * a.bps;
            a.out = static_cast<uint8_t *>(d_out) + done * (d_sel ? row : nfft) * out_esz;
            hipError_t e = d_sel ? launch_v2_spectro_sel(a, log2n, (uint32_t)run, d_sel, row, c->stream)
                                 : launch_v2_spectro(a, log2n, (uint32_t)run, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "spectrogram launch: %s", hipGetErrorString(e));
            done += a.n_lines;
        }
        return SPEC_OK;
    }
    // one launch covers at most 2^31 - 1 workgroups; split very long recordings
    uint64_t done = 0;
    while (done < n_lines) {
        const uint64_t rem = n_lines - done;
        a.lines_per_wg = c->opt_lines_per_wg > 0 ? (uint32_t)((c->opt_lines_per_wg + lpw - 1) / lpw * lpw)
                                                  : pick_lines_per_wg(rem, lpw);
        const uint64_t max_lines = (uint64_t)a.lines_per_wg * 0x7FFFFFFFull;
        a.n_lines = rem < max_lines ? rem : max_lines;
        a.iq = d_first + done * (uint64_t)hop * a.bps;
        a.out = static_cast<uint8_t *>(d_out) + done * nfft * out_esz;
        hipError_t e = f64 ? launch_spectro_f64(a, log2n, c->stream) : launch_spectro_f32(a, log2n, c->stream);
        if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "spectrogram launch: %s", hipGetErrorString(e));
        done += a.n_lines;
    }
    return SPEC_OK;
}

static spec_status check_common(spec_ctx *c, const void *iq, const void *out, int dt, uint32_t nfft, uint32_t hop,
                                int window, int *log2n) {
    if (!c) return SPEC_EINVAL;
    if (!iq || !out) return fail(c, SPEC_EINVAL, "null buffer");
    if (!dtype_valid(dt)) return fail(c, SPEC_EINVAL, "bad spec_dtype %d", dt);
    if (nfft == 0 || (nfft & (nfft - 1)) != 0)  // commons-math3 throws for these (SS:29)
        return fail(c, SPEC_EINVAL, "nfft = %u is not a power of two", nfft);
    if (hop == 0) return fail(c, SPEC_EINVAL, "hop must be >= 1");
    if (window != SPEC_WIN_RECT && window != SPEC_WIN_HANN) return fail(c, SPEC_EINVAL, "bad window %d", window);
    *log2n = ilog2(nfft);
    return SPEC_OK;
}

// read [off, off + len) of the file into dst; holes of a sparse file and bytes past EOF read as zero
static bool pread_all(int fd, uint8_t *dst, uint64_t len, uint64_t off) {
    while (len) {
        const ssize_t r = pread(fd, dst, len > (1u << 30) ? (1u << 30) : (size_t)len, (off_t)off);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) { memset(dst, 0, len); return true; }
        dst += r; off += (uint64_t)r; len -= (uint64_t)r;
    }
    return true;
}
// the same with the range cut over a few threads: one thread copies out of the page cache at 5-10 GB/s,
// PCIe Gen5 x16 takes 60
static bool pread_parallel(int fd, uint8_t *dst, uint64_t len, uint64_t off) {
    const uint64_t min_part = 8ull << 20;
    unsigned parts = (unsigned)(len / min_part);
    if (parts > 6) parts = 6;
    if (parts < 2) return pread_all(fd, dst, len, off);
    std::vector<std::thread> th;
    std::vector<char> ok(parts, 0);
    const uint64_t per = (len / parts + 4095) & ~4095ull;
    for (unsigned i = 0; i < parts; ++i) {
        const uint64_t a = (uint64_t)i * per, b = i + 1 == parts ? len : (a + per < len ? a + per : len);
        if (a >= b) { ok[i] = 1; continue; }
        try { th.emplace_back([=, &ok] { ok[i] = pread_all(fd, dst + a, b - a, off + a); }); }
        catch (...) { ok[i] = pread_all(fd, dst + a, b - a, off + a); }
    }
    for (auto &t : th) t.join();
    for (char k : ok) if (!k) return false;
    return true;
}
static spec_status grow_pinned(spec_ctx *c, size_t need) {
    if (c->pin_in_bytes >= need) return SPEC_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->s_in) HIP_TRY(c, hipStreamSynchronize(c->s_in));
    if (c->pin_in) { (void)hipHostFree(c->pin_in); c->pin_in = nullptr; c->pin_in_bytes = 0; }
    HIP_TRY(c, hipHostMalloc(&c->pin_in, need, hipHostMallocDefault));
    c->pin_in_bytes = need;
    for (int i = 0; i < 2; ++i)
        if (!c->ev_pin[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pin[i], hipEventDisableTiming));
    return SPEC_OK;
}

extern "C" {

// rec != nullptr: the input is the payload of a recording on disk (iq is ignored, iq_on_device must be 0)
static spec_status waterfall_impl(spec_ctx *c, const void *iq, int iq_on_device, uint64_t n_bytes, uint64_t start_byte,
                                  spec_dtype dt, uint32_t nfft, uint32_t hop, uint64_t n_lines, spec_window window,
                                  spec_out_fmt out_fmt, double eof_fill, void *out, int out_on_device,
                                  const int32_t *d_sel, uint32_t sel_row, const spec_recording *rec = nullptr) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    int log2n = 0;
    if (rec) {
        iq_on_device = 0;
        n_bytes = rec->bytes;
        struct stat now;  // a file that SHRANK since it was opened: touching the mapping past its end is SIGBUS for the
                          // whole host process (a JVM included); pread zero-fills instead, so take that path for this call
        const bool shrunk = fstat(rec->fd, &now) != 0 || (uint64_t)now.st_size < rec->header + rec->bytes;
        if (rec->map && !c->opt_rec_pread && !shrunk) {
            // the file is mapped: its pages are ordinary pageable host memory and take the staged pipeline of
            // spec_waterfall as they are (measured: 93 GB/s over PCIe, both directions together, against 52 for
            // pread into the pinned ring, which pays one more copy out of the page cache)
            iq = static_cast<const uint8_t *>(rec->map) + rec->header;
            rec = nullptr;
        } else {
            iq = rec;  // only has to be non-null from here on
        }
    }
    spec_status st = check_common(c, iq, out, dt, nfft, hop, window, &log2n);
    if (st != SPEC_OK) return st;
    if (out_fmt < SPEC_OUT_DB20_F32 || out_fmt > SPEC_OUT_POW_F64) return fail(c, SPEC_EINVAL, "bad out_fmt %d", out_fmt);
    if (n_lines == 0) return SPEC_OK;
    const uint64_t bps = spec_bytes_per_sample(dt), out_esz = out_fmt >= SPEC_OUT_DB20_F64 ? 8 : 4;
    const uint64_t row = d_sel ? sel_row : nfft;  // elements per output line
    {
        uint64_t tile_bytes = 0;
        if (!mul_add_u64(n_lines, row * out_esz, 0, &tile_bytes))
            return fail(c, SPEC_ERANGE, "%llu lines of %u bins do not fit a 64-bit byte count", (unsigned long long)n_lines, nfft);
    }
    // MainController.java:987 -- lines whose last byte is inside the buffer
    uint64_t n_valid = spec_count_lines(n_bytes, start_byte, dt, nfft, hop);
    if (n_valid > n_lines) n_valid = n_lines;
    const bool f64 = out_fmt >= SPEC_OUT_DB20_F64 || dt == SPEC_DT_CF64_LE || dt == SPEC_DT_CF64_BE;
    {
        int l1 = 0, l2 = 0;
        if (n_valid && !plan_supported(log2n, f64) && !large_split(log2n, f64, &l1, &l2))
            return fail(c, SPEC_EUNSUPPORTED, "nfft = %u is not supported in %s (64 ... 65536)", nfft,
                        f64 ? "fp64" : "fp32");
    }

    if (iq_on_device && out_on_device) {
        const uint8_t *first = static_cast<const uint8_t *>(iq) + start_byte;
        if (n_valid && (reinterpret_cast<uintptr_t>(first) % component_bytes(dt)) != 0)
            return fail(c, SPEC_EINVAL, "device input is not aligned to its %u-byte components", component_bytes(dt));
        st = run_lines(c, first, dt, log2n, hop, n_valid, window, out_fmt, out, d_sel, sel_row);
        if (st != SPEC_OK) return st;
        if (n_valid < n_lines) {  // MC:994-998
            hipError_t e = launch_fill(static_cast<uint8_t *>(out) + n_valid * row * out_esz, (n_lines - n_valid) * row,
                                       eof_fill, out_esz == 8, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "fill launch: %s", hipGetErrorString(e));
        }
        return SPEC_OK;
    }

    // staged path: walk the valid lines in chunks.  Host buffers are pageable (the Java side hands over a
    // mapped file and a heap array), so a copy blocks the thread that issues it; the copies of chunk i+1
    // (host -> device, this thread) and of chunk i-1 (device -> host, a helper thread) therefore run on
    // their own streams around the kernels of chunk i, and PCIe carries both directions at once.
    const uint64_t chunk_cap = (uint64_t)c->opt_stage_chunk_mb << 20;
    uint64_t lines_per_chunk = chunk_cap / (row * out_esz);
    const uint64_t by_in = chunk_cap > nfft * bps ? (chunk_cap - nfft * bps) / ((uint64_t)hop * bps) + 1 : 1;
    if (by_in < lines_per_chunk) lines_per_chunk = by_in;
    if (lines_per_chunk == 0) lines_per_chunk = 1;
    if (lines_per_chunk > n_valid && n_valid) lines_per_chunk = n_valid;
    // slot strides of the two-deep staging buffers (256-byte multiples keep every slot aligned)
    const uint64_t in_chunk_bytes = ((((lines_per_chunk - 1) * hop + nfft) * bps) + 255) & ~255ull,
                   out_chunk_bytes = (lines_per_chunk * row * out_esz + 255) & ~255ull;
    if (iq_on_device && n_valid &&
        reinterpret_cast<uintptr_t>(static_cast<const uint8_t *>(iq) + start_byte) % component_bytes(dt) != 0)
        return fail(c, SPEC_EINVAL, "device input is not aligned to its %u-byte components", component_bytes(dt));
    const uint64_t n_chunks = n_valid ? (n_valid + lines_per_chunk - 1) / lines_per_chunk : 0;
    const uint64_t slots = n_chunks > 1 ? 2 : 1;
    if (n_chunks == 1) {  // everything fits one chunk (every interactive call): one stream, no helper, least latency
        const uint64_t in_len = ((n_valid - 1) * hop + nfft) * bps;
        const uint8_t *d_in = static_cast<const uint8_t *>(iq) + start_byte;
        if (!iq_on_device) {
            st = grow(c, &c->stage_in, &c->stage_in_bytes, in_len);
            if (st != SPEC_OK) return st;
            if (rec) {  // file -> pinned buffer -> device
                if ((st = grow_pinned(c, in_len)) != SPEC_OK) return st;
                if (!pread_parallel(rec->fd, static_cast<uint8_t *>(c->pin_in), in_len, rec->header + start_byte))
                    return fail(c, SPEC_EDEVICE, "reading %s: %s", rec->path.c_str(), strerror(errno));
                d_in = static_cast<const uint8_t *>(c->pin_in);
            }
            HIP_TRY(c, hipMemcpyAsync(c->stage_in, d_in, in_len, hipMemcpyHostToDevice, c->stream));
            d_in = static_cast<const uint8_t *>(c->stage_in);
        }
        void *d_out = out;
        if (!out_on_device) {
            st = grow(c, &c->stage_out, &c->stage_out_bytes, n_valid * row * out_esz);
            if (st != SPEC_OK) return st;
            d_out = c->stage_out;
        }
        st = run_lines(c, d_in, dt, log2n, hop, n_valid, window, out_fmt, d_out, d_sel, sel_row);
        if (st != SPEC_OK) return st;
        if (!out_on_device)
            HIP_TRY(c, hipMemcpyAsync(out, d_out, n_valid * row * out_esz, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (n_chunks > 1) {
        if (!iq_on_device) { st = grow(c, &c->stage_in, &c->stage_in_bytes, slots * in_chunk_bytes); if (st != SPEC_OK) return st; }
        if (rec) { st = grow_pinned(c, slots * in_chunk_bytes); if (st != SPEC_OK) return st; }
        if (!out_on_device) { st = grow(c, &c->stage_out, &c->stage_out_bytes, slots * out_chunk_bytes); if (st != SPEC_OK) return st; }
        if (!c->s_in) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
            HIP_TRY(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                HIP_TRY(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
                HIP_TRY(c, hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
            }
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // earlier work on the staging buffers
    }
    struct OutJob { uint64_t l0, nl; int slot; };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<OutJob> jobs;
    bool closing = false, slot_busy[2] = {false, false};
    hipError_t worker_err = hipSuccess;
    bool threaded = !out_on_device && n_chunks > 1;
    uint8_t *const h_out = static_cast<uint8_t *>(out);
    auto copy_out = [&](const OutJob &j) -> hipError_t {  // device -> host of one finished chunk
        hipError_t e = hipStreamWaitEvent(c->s_out, c->ev_done[j.slot], 0);
        if (e == hipSuccess)
            e = hipMemcpyAsync(h_out + j.l0 * row * out_esz, static_cast<uint8_t *>(c->stage_out) + j.slot * out_chunk_bytes,
                               j.nl * row * out_esz, hipMemcpyDeviceToHost, c->s_out);
        if (e == hipSuccess) e = hipStreamSynchronize(c->s_out);
        return e;
    };
    std::thread worker;
    auto worker_main = [&] {
            (void)hipSetDevice(c->device);
            for (;;) {
                OutJob j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !jobs.empty() || closing; });
                    if (jobs.empty()) return;
                    j = jobs.front();
                    jobs.pop_front();
                }
                const hipError_t e = copy_out(j);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (e != hipSuccess && worker_err == hipSuccess) worker_err = e;
                    slot_busy[j.slot] = false;
                }
                cv.notify_all();
            }
        };
    if (threaded) {
        try { worker = std::thread(worker_main); }
        catch (...) { threaded = false; }  // no thread to be had: copy out from this thread, without the overlap
    }
    spec_status pst = SPEC_OK;
    hipError_t perr = hipSuccess;
    uint64_t chunk = 0;
    for (uint64_t l0 = 0; l0 < n_valid && n_chunks > 1; l0 += lines_per_chunk, ++chunk) {
        const uint64_t nl = (n_valid - l0 < lines_per_chunk) ? n_valid - l0 : lines_per_chunk;
        const uint64_t in_off = start_byte + l0 * hop * bps, in_len = ((nl - 1) * hop + nfft) * bps;
        const int slot = (int)(chunk & 1);
        if (threaded) {  // the output slot must have been drained (chunk - 2)
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !slot_busy[slot]; });
            if (worker_err != hipSuccess) break;
        }
        const uint8_t *d_in;
        if (iq_on_device) {
            d_in = static_cast<const uint8_t *>(iq) + in_off;
        } else {
            uint8_t *dst = static_cast<uint8_t *>(c->stage_in) + slot * in_chunk_bytes;
            const uint8_t *h_src = static_cast<const uint8_t *>(iq) + in_off;
            if (rec) {  // pread the chunk into its pinned slot (free once the copy of chunk - 2 has left it)
                uint8_t *pin = static_cast<uint8_t *>(c->pin_in) + slot * in_chunk_bytes;
                if (chunk >= 2) perr = hipEventSynchronize(c->ev_pin[slot]);
                if (perr == hipSuccess && !pread_parallel(rec->fd, pin, in_len, rec->header + in_off)) {
                    pst = fail(c, SPEC_EDEVICE, "reading %s: %s", rec->path.c_str(), strerror(errno));
                    break;
                }
                h_src = pin;
            }
            if (perr == hipSuccess && chunk >= 2) perr = hipStreamWaitEvent(c->s_in, c->ev_done[slot], 0);  // kernels of chunk - 2 have read the slot
            if (perr == hipSuccess) perr = hipMemcpyAsync(dst, h_src, in_len, hipMemcpyHostToDevice, c->s_in);
            if (perr == hipSuccess && rec) perr = hipEventRecord(c->ev_pin[slot], c->s_in);
            if (perr == hipSuccess) perr = hipEventRecord(c->ev_in[slot], c->s_in);
            if (perr == hipSuccess) perr = hipStreamWaitEvent(c->stream, c->ev_in[slot], 0);
            if (perr != hipSuccess) break;
            d_in = dst;
        }
        void *d_out = out_on_device ? static_cast<void *>(h_out + l0 * row * out_esz)
                                    : static_cast<void *>(static_cast<uint8_t *>(c->stage_out) + slot * out_chunk_bytes);
        pst = run_lines(c, d_in, dt, log2n, hop, nl, window, out_fmt, d_out, d_sel, sel_row);
        if (pst != SPEC_OK) break;
        perr = hipEventRecord(c->ev_done[slot], c->stream);
        if (perr != hipSuccess) break;
        if (!out_on_device) {
            const OutJob j{l0, nl, slot};
            if (threaded) {
                { std::lock_guard<std::mutex> lk(mu); slot_busy[slot] = true; jobs.push_back(j); }
                cv.notify_all();
            } else {
                perr = copy_out(j);
                if (perr != hipSuccess) break;
            }
        }
    }
    if (threaded) {  // always join the helper before this frame goes away
        { std::lock_guard<std::mutex> lk(mu); closing = true; }
        cv.notify_all();
        worker.join();
    }
    if (n_chunks > 1) {  // the staging buffers are free again, the events consumed
        (void)hipStreamSynchronize(c->s_in);
        (void)hipStreamSynchronize(c->stream);
    }
    if (pst != SPEC_OK) return pst;
    if (perr == hipSuccess) perr = worker_err;
    if (perr != hipSuccess)
        return fail(c, perr == hipErrorOutOfMemory ? SPEC_ENOMEM : SPEC_EDEVICE, "staging pipeline: %s", hipGetErrorString(perr));
    if (n_valid < n_lines) {  // MC:994-998
        const uint64_t n = (n_lines - n_valid) * row;
        if (out_on_device) {
            hipError_t e = launch_fill(static_cast<uint8_t *>(out) + n_valid * row * out_esz, n, eof_fill, out_esz == 8,
                                       c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "fill launch: %s", hipGetErrorString(e));
        } else if (out_esz == 8) {
            double *p = static_cast<double *>(out) + n_valid * row;
            for (uint64_t i = 0; i < n; ++i) p[i] = eof_fill;
        } else {
            float *p = static_cast<float *>(out) + n_valid * row;
            for (uint64_t i = 0; i < n; ++i) p[i] = (float)eof_fill;
        }
    }
    return SPEC_OK;
}

spec_status spec_waterfall(spec_ctx *c, const void *iq, int iq_on_device, uint64_t n_bytes, uint64_t start_byte,
                           spec_dtype dt, uint32_t nfft, uint32_t hop, uint64_t n_lines, spec_window window,
                           spec_out_fmt out_fmt, double eof_fill, void *out, int out_on_device) {
    return waterfall_impl(c, iq, iq_on_device, n_bytes, start_byte, dt, nfft, hop, n_lines, window, out_fmt, eof_fill, out,
                          out_on_device, nullptr, 0);
}

// ---- one waterfall across several devices --------------------------------------------------
void spec_shard_lines(uint64_t n_lines, uint32_t n_shards, uint32_t shard, uint64_t *first_line, uint64_t *end_line) {
    uint64_t a = 0, b = 0;
    if (n_shards && shard < n_shards) {  // r L / n without overflowing 64 bits
        a = (uint64_t)(((unsigned __int128)n_lines * shard) / n_shards);
        b = (uint64_t)(((unsigned __int128)n_lines * (shard + 1)) / n_shards);
    }
    if (first_line) *first_line = a;
    if (end_line) *end_line = b;
}

void spec_shard_span(uint64_t first_line, uint64_t end_line, spec_dtype dt, uint32_t nfft, uint32_t hop,
                     uint64_t *first_byte, uint64_t *n_bytes) {
    const uint64_t bps = spec_bytes_per_sample(dt);
    if (first_byte) *first_byte = first_line * hop * bps;
    if (n_bytes) *n_bytes = end_line > first_line ? ((end_line - first_line - 1) * hop + nfft) * bps : 0;
}

// ---- peer copies of the multi-context entries: which path they take, and the "multi_verify" self-check ------------
static const char *peer_path_name(const spec_ctx *c) {
    return c->multi_peer_access == 2 ? "same device" : c->multi_peer_access == 1 ? "direct: peer access enabled"
         : c->multi_peer_access == 0 ? "staged by the runtime: no peer access" : "not established";
}
// direct xGMI writes where the platform allows them; staged by the runtime otherwise.  Remembered on the peer context
// ("multi_peer_access" of spec_get_option) and named in every message about a peer copy.
static spec_status peer_path(spec_ctx *c, const spec_ctx *root) {
    if (c->device == root->device) { c->multi_peer_access = 2; return SPEC_OK; }
    int can = 0;
    c->multi_peer_access = 0;
    if (hipDeviceCanAccessPeer(&can, c->device, root->device) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(root->device, 0);
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) c->multi_peer_access = 1;
        if (e != hipSuccess) (void)hipGetLastError();
    }
    return SPEC_OK;
}
// After the copies have completed: the checksum of every piece where it LANDED (a kernel on the consumer's device, on a
// stream this context keeps there) against the checksum taken on this device before it left (cks_src, device memory
// of this context, one word per piece).
static spec_status verify_landed(spec_ctx *c, const spec_ctx *root, const std::vector<std::pair<const void *, uint64_t>> &pieces,
                                 const unsigned long long *cks_src, const char *who) {
    const size_t n = pieces.size();
    if (n == 0) return SPEC_OK;
    std::vector<unsigned long long> src(n), dst(n);
    HIP_TRY(c, hipMemcpyAsync(src.data(), cks_src, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipError_t e = hipSetDevice(root->device);
    if (e == hipSuccess && (c->s_verify == nullptr || c->s_verify_dev != root->device)) {
        if (c->s_verify) { (void)hipStreamDestroy(c->s_verify); c->s_verify = nullptr; }
        e = hipStreamCreateWithFlags(&c->s_verify, hipStreamNonBlocking);
        c->s_verify_dev = root->device;
    }
    if (e == hipSuccess && (c->cks_dst_bytes < n * 8 || c->cks_dst_dev != root->device)) {
        (void)hipFree(c->cks_dst); c->cks_dst = nullptr; c->cks_dst_bytes = 0;
        e = hipMalloc(&c->cks_dst, n * 8);
        if (e == hipSuccess) { c->cks_dst_bytes = n * 8; c->cks_dst_dev = root->device; }
    }
    unsigned long long *d = static_cast<unsigned long long *>(c->cks_dst);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0, n * 8, c->s_verify);
    for (size_t j = 0; j < n && e == hipSuccess; ++j) e = launch_checksum(pieces[j].first, pieces[j].second, d + j, c->s_verify);
    if (e == hipSuccess) e = hipMemcpyAsync(dst.data(), d, n * 8, hipMemcpyDeviceToHost, c->s_verify);
    if (e == hipSuccess) e = hipStreamSynchronize(c->s_verify);
    (void)hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "%s: multi_verify on device %d: %s", who, root->device, hipGetErrorString(e));
    for (size_t j = 0; j < n; ++j)
        if (src[j] != dst[j])
            return fail(c, SPEC_EDEVICE, "%s: multi_verify: piece %zu of %zu (%llu bytes) sent from device %d does not match what "
                        "landed on device %d (checksum %016llx sent, %016llx landed; %s)", who, j, n,
                        (unsigned long long)pieces[j].second, c->device, root->device, src[j], dst[j], peer_path_name(c));
    c->multi_verified = (int64_t)n;
    return SPEC_OK;
}

// shard [l0, l1) of a multi-device waterfall on context c.  `src` / `src_bytes` / `src_off`: the buffer this
// context reads (the whole host recording, or its own device span) and the byte its first line starts at.
static spec_status multi_shard(spec_ctx *c, spec_ctx *root, bool is_root, const void *src, int src_on_device,
                               uint64_t src_bytes, uint64_t src_off, spec_dtype dt, uint32_t nfft, uint32_t hop,
                               uint64_t l0, uint64_t l1, spec_window window, spec_out_fmt out_fmt, void *out,
                               int out_on_device, uint32_t n_chunks) {
    // what spec_get_option reports is about THIS call: a context that sends nothing now (host tile, consumer, empty shard) must
    // not keep the previous call's "verified" count or copy path
    c->multi_verified = 0;
    c->multi_peer_access = -1;
    if (l1 <= l0) return SPEC_OK;
    const uint64_t bps = spec_bytes_per_sample(dt), out_esz = out_fmt >= SPEC_OUT_DB20_F64 ? 8 : 4;
    const uint64_t row_bytes = (uint64_t)nfft * out_esz;
    uint8_t *const out_rows = static_cast<uint8_t *>(out) + l0 * row_bytes;
    if (!out_on_device || is_root) {  // host tile: own rows, own copy-out; consumer: in place
        spec_status st = waterfall_impl(c, src, src_on_device, src_bytes, src_off, dt, nfft, hop, l1 - l0, window, out_fmt,
                                        -150.0, out_rows, out_on_device, nullptr, 0);
        if (st != SPEC_OK) return st;
        return out_on_device ? spec_sync(c) : SPEC_OK;
    }
    // a peer of a device-resident tile: pieces through a two-slot buffer, each sent behind its own kernels
    Enter g(c);
    if (n_chunks == 0) n_chunks = 8;
    if ((uint64_t)n_chunks > l1 - l0) n_chunks = (uint32_t)(l1 - l0);
    const uint64_t per = (l1 - l0 + n_chunks - 1) / n_chunks;  // lines per piece (the last one may be shorter)
    const uint32_t n_pieces = (uint32_t)((l1 - l0 + per - 1) / per);
    const uint64_t slot_bytes = (per * row_bytes + 255) & ~255ull;
    spec_status st = grow(c, &c->multi_buf, &c->multi_buf_bytes, 2 * slot_bytes);
    if (st != SPEC_OK) return st;
    if (!c->s_peer) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->s_peer, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_mdone[i], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_mcopied[i], hipEventDisableTiming));
        }
    }
    st = peer_path(c, root);
    if (st != SPEC_OK) return st;
    const bool verify = root->opt_multi_verify || c->opt_multi_verify;
    c->multi_verified = 0;
    unsigned long long *cks_src = nullptr;
    if (verify) {
        if ((st = grow(c, &c->cks_src, &c->cks_src_bytes, (size_t)n_pieces * 8)) != SPEC_OK) return st;
        cks_src = static_cast<unsigned long long *>(c->cks_src);
        HIP_TRY(c, hipMemsetAsync(cks_src, 0, (size_t)n_pieces * 8, c->stream));
    }
    // (Any failure inside the loop leaves it by `break`: the two synchronisations below must run before this function
    // returns -- peer copies into `out` may still be in flight, and the slots and events are reused by the next call.)
    hipError_t le = hipSuccess;
    uint32_t j = 0;
    for (uint64_t a = l0; a < l1 && st == SPEC_OK && le == hipSuccess; a += per, ++j) {
        const uint64_t b = a + per < l1 ? a + per : l1;
        const int slot = (int)(j & 1u);
        uint8_t *buf = static_cast<uint8_t *>(c->multi_buf) + slot * slot_bytes;
        if (j >= 2 && (le = hipStreamWaitEvent(c->stream, c->ev_mcopied[slot], 0)) != hipSuccess) break;  // piece j - 2 has left the slot
        st = waterfall_impl(c, src, src_on_device, src_bytes, src_off + (a - l0) * hop * bps, dt, nfft, hop, b - a, window,
                            out_fmt, -150.0, buf, 1, nullptr, 0);
        if (st != SPEC_OK) break;
        if (verify && (le = launch_checksum(buf, (b - a) * row_bytes, cks_src + j, c->stream)) != hipSuccess) break;
        if ((le = hipEventRecord(c->ev_mdone[slot], c->stream)) != hipSuccess) break;
        if ((le = hipStreamWaitEvent(c->s_peer, c->ev_mdone[slot], 0)) != hipSuccess) break;
        if ((le = hipMemcpyPeerAsync(static_cast<uint8_t *>(out) + a * row_bytes, root->device, buf, c->device,
                                     (b - a) * row_bytes, c->s_peer)) != hipSuccess) break;
        if ((le = hipEventRecord(c->ev_mcopied[slot], c->s_peer)) != hipSuccess) break;
    }
    const hipError_t e1 = hipStreamSynchronize(c->s_peer), e2 = hipStreamSynchronize(c->stream);
    if (st != SPEC_OK) return st;
    if (le != hipSuccess) return fail(c, SPEC_EDEVICE, "peer copy (%s): %s", peer_path_name(c), hipGetErrorString(le));
    if (e1 != hipSuccess || e2 != hipSuccess)
        return fail(c, SPEC_EDEVICE, "peer copy (%s): %s", peer_path_name(c), hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    if (verify) {
        std::vector<std::pair<const void *, uint64_t>> pieces;
        for (uint64_t a = l0; a < l1; a += per)
            pieces.emplace_back(static_cast<const uint8_t *>(out) + a * row_bytes, ((a + per < l1 ? a + per : l1) - a) * row_bytes);
        if (c->opt_multi_verify_corrupt) {  // tests: one word of the LAST piece is damaged where it landed
            c->opt_multi_verify_corrupt = 0;
            const uint32_t bad = 0x7FC0DEADu;
            HIP_TRY(c, hipMemcpy(const_cast<void *>(pieces.back().first), &bad, 4, hipMemcpyHostToDevice));
        }
        return verify_landed(c, root, pieces, cks_src, "spec_waterfall_multi");
    }
    return SPEC_OK;
}

spec_status spec_waterfall_multi(spec_ctx *const *ctx, uint32_t n_ctx, const void *const *iq, int iq_on_device,
                                 uint64_t n_bytes, uint64_t start_byte, spec_dtype dt, uint32_t nfft, uint32_t hop,
                                 uint64_t n_lines, spec_window window, spec_out_fmt out_fmt, double eof_fill, void *out,
                                 int out_on_device, uint32_t n_chunks) {
    if (!ctx || n_ctx == 0 || !ctx[0]) return SPEC_EINVAL;
    spec_ctx *root = ctx[0];
    if (n_ctx > 64) return fail(root, SPEC_EINVAL, "spec_waterfall_multi: %u contexts (at most 64)", n_ctx);
    for (uint32_t r = 0; r < n_ctx; ++r) {
        if (!ctx[r]) return fail(root, SPEC_EINVAL, "spec_waterfall_multi: context %u is NULL", r);
        for (uint32_t q = 0; q < r; ++q)
            if (ctx[q] == ctx[r]) return fail(root, SPEC_EINVAL, "spec_waterfall_multi: context %u appears twice", r);
    }
    if (!iq || (!iq_on_device && !iq[0])) return fail(root, SPEC_EINVAL, "null buffer");
    int log2n = 0;  // device shards: an empty shard (fewer lines than contexts) has no buffer, checked per shard below
    spec_status st = check_common(root, iq_on_device ? static_cast<const void *>(iq) : iq[0], out, dt, nfft, hop, window, &log2n);
    if (st != SPEC_OK) return st;
    if (out_fmt < SPEC_OUT_DB20_F32 || out_fmt > SPEC_OUT_POW_F64) return fail(root, SPEC_EINVAL, "bad out_fmt %d", out_fmt);
    if (n_lines == 0) return SPEC_OK;
    const uint64_t out_esz = out_fmt >= SPEC_OUT_DB20_F64 ? 8 : 4;
    {
        uint64_t tile_bytes = 0;
        if (!mul_add_u64(n_lines, (uint64_t)nfft * out_esz, 0, &tile_bytes))
            return fail(root, SPEC_ERANGE, "%llu lines of %u bins do not fit a 64-bit byte count", (unsigned long long)n_lines, nfft);
    }
    uint64_t n_valid = spec_count_lines(n_bytes, start_byte, dt, nfft, hop);  // MainController.java:987
    if (n_valid > n_lines) n_valid = n_lines;
    if (iq_on_device)
        for (uint32_t r = 0; r < n_ctx; ++r) {
            uint64_t a, b;
            spec_shard_lines(n_valid, n_ctx, r, &a, &b);
            if (b > a && !iq[r]) return fail(root, SPEC_EINVAL, "spec_waterfall_multi: shard %u has lines but no buffer", r);
        }
    std::vector<spec_status> status(n_ctx, SPEC_OK);
    auto run = [&](uint32_t r) {
        uint64_t l0, l1, first = 0, span = 0;
        spec_shard_lines(n_valid, n_ctx, r, &l0, &l1);
        spec_shard_span(l0, l1, dt, nfft, hop, &first, &span);
        if (iq_on_device)  // the shard's own span: byte 0 of iq[r] is the first byte of line l0
            status[r] = multi_shard(ctx[r], root, r == 0, iq[r], 1, span, 0, dt, nfft, hop, l0, l1, window, out_fmt, out,
                                    out_on_device, n_chunks);
        else
            status[r] = multi_shard(ctx[r], root, r == 0, iq[0], 0, n_bytes, start_byte + first, dt, nfft, hop, l0, l1, window,
                                    out_fmt, out, out_on_device, n_chunks);
    };
    std::vector<std::thread> th;
    for (uint32_t r = 1; r < n_ctx; ++r) {
        try { th.emplace_back(run, r); }
        catch (...) { run(r); }  // no thread to be had: this shard runs here, without the overlap
    }
    run(0);
    for (auto &t : th) t.join();
    for (uint32_t r = 0; r < n_ctx; ++r)
        if (status[r] != SPEC_OK) {
            if (r) {
                std::string msg;
                { std::lock_guard<std::recursive_mutex> lk(ctx[r]->mu); msg = ctx[r]->err; }
                return fail(root, status[r], "shard %u of %u: %s", r, n_ctx, msg.c_str());
            }
            return status[r];
        }
    if (n_valid < n_lines) {  // MC:994-998
        const uint64_t n = (n_lines - n_valid) * nfft;
        if (out_on_device) {
            Enter g(root);
            hipError_t e = launch_fill(static_cast<uint8_t *>(out) + n_valid * nfft * out_esz, n, eof_fill, out_esz == 8, root->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(root->stream);
            if (e != hipSuccess) return fail(root, SPEC_EDEVICE, "fill launch: %s", hipGetErrorString(e));
        } else if (out_esz == 8) {
            double *p = static_cast<double *>(out) + n_valid * nfft;
            for (uint64_t i = 0; i < n; ++i) p[i] = eof_fill;
        } else {
            float *p = static_cast<float *>(out) + n_valid * nfft;
            for (uint64_t i = 0; i < n; ++i) p[i] = (float)eof_fill;
        }
    }
    return SPEC_OK;
}

// ---- recordings on disk (SURVEY 8f "next" #3; SigMfHelper.java:49-94) ---------------------------------
spec_status spec_open_recording(spec_ctx *c, const char *data_path, uint64_t header_bytes, spec_recording **out) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!data_path || !out) return fail(c, SPEC_EINVAL, "spec_open_recording: null argument");
    *out = nullptr;
    const int fd = open(data_path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return fail(c, SPEC_EINVAL, "cannot open %s: %s", data_path, strerror(errno));
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        close(fd);
        return fail(c, SPEC_EINVAL, "%s is not a regular file", data_path);
    }
    spec_recording *r = new (std::nothrow) spec_recording;
    if (!r) { close(fd); return fail(c, SPEC_ENOMEM, "out of host memory"); }
    r->fd = fd;
    r->header = header_bytes;
    const uint64_t size = (uint64_t)sb.st_size;
    r->bytes = size > header_bytes ? size - header_bytes : 0;  // SMH:74-76: max(0, channelSize - headerBytes)
    if (size) {  // FileChannel.map without the Integer.MAX_VALUE cap (SMH:78-84); pread remains when it is refused
        void *m = mmap(nullptr, (size_t)size, PROT_READ, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) { r->map = m; r->map_len = (size_t)size; }
    }
    try { r->path = data_path; } catch (...) {}
    *out = r;
    return SPEC_OK;
}

uint64_t spec_recording_bytes(const spec_recording *rec) { return rec ? rec->bytes : 0; }

void spec_close_recording(spec_recording *rec) {
    if (!rec) return;
    if (rec->map) munmap(rec->map, rec->map_len);
    if (rec->fd >= 0) close(rec->fd);
    delete rec;
}

spec_status spec_waterfall_recording(spec_ctx *c, const spec_recording *rec, uint64_t start_byte, spec_dtype dt,
                                     uint32_t nfft, uint32_t hop, uint64_t n_lines, spec_window window,
                                     spec_out_fmt out_fmt, double eof_fill, void *out, int out_on_device) {
    if (!c) return SPEC_EINVAL;
    if (!rec) return fail(c, SPEC_EINVAL, "null recording");
    return waterfall_impl(c, nullptr, 0, 0, start_byte, dt, nfft, hop, n_lines, window, out_fmt, eof_fill, out, out_on_device,
                          nullptr, 0, rec);
}

// computeMagnitudes with the 64-bit offset the reference cannot express (SS:33 takes an int startByte,
// MC:985 casts to int): one slice of a recording of any size, fp64 pipeline
spec_status spec_compute_magnitudes_recording(spec_ctx *c, const spec_recording *rec, uint64_t start_byte, uint32_t nfft,
                                              const char *datatype, int big_endian, double *out) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!rec || !out || !datatype) return fail(c, SPEC_EINVAL, "null argument");
    if (nfft == 0 || (nfft & (nfft - 1)) != 0) return fail(c, SPEC_EINVAL, "nfft = %u is not a power of two", nfft);
    spec_dtype dt = spec_dtype_from_sigmf(datatype);
    switch (dt) {
    case SPEC_DT_CI16_LE: case SPEC_DT_CI16_BE: dt = big_endian ? SPEC_DT_CI16_BE : SPEC_DT_CI16_LE; break;
    case SPEC_DT_CF32_LE: case SPEC_DT_CF32_BE: dt = big_endian ? SPEC_DT_CF32_BE : SPEC_DT_CF32_LE; break;
    case SPEC_DT_CF64_LE: case SPEC_DT_CF64_BE: dt = big_endian ? SPEC_DT_CF64_BE : SPEC_DT_CF64_LE; break;
    default: break;
    }
    if (kind_of(dt, c->flags) == K_ZERO) {  // SS:60-63: nothing is read, flat 20 log10(1e-10)
        for (uint32_t i = 0; i < nfft; ++i) out[i] = -200.0;
        return SPEC_OK;
    }
    const uint64_t span = (uint64_t)nfft * spec_bytes_per_sample(dt);
    if (start_byte > rec->bytes || span > rec->bytes - start_byte)
        return fail(c, SPEC_ERANGE, "IndexOutOfBounds: bytes [%llu, +%llu) of %llu", (unsigned long long)start_byte,
                    (unsigned long long)span, (unsigned long long)rec->bytes);
    return waterfall_impl(c, nullptr, 0, 0, start_byte, dt, nfft, nfft, 1, SPEC_WIN_RECT, SPEC_OUT_DB20_F64, -150.0, out, 0,
                          nullptr, 0, rec);
}

spec_status spec_compute_magnitudes(spec_ctx *c, const void *buffer, uint64_t capacity, int64_t start_byte,
                                    uint32_t nfft, const char *datatype, int big_endian, double *out) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!buffer || !out || !datatype) return fail(c, SPEC_EINVAL, "null argument");
    if (nfft == 0 || (nfft & (nfft - 1)) != 0) return fail(c, SPEC_EINVAL, "nfft = %u is not a power of two", nfft);
    // datatype: startsWith rules of SS:35-38; byte order is the buffer's (SMH:87-91)
    spec_dtype dt = spec_dtype_from_sigmf(datatype);
    switch (dt) {
    case SPEC_DT_CI16_LE: case SPEC_DT_CI16_BE: dt = big_endian ? SPEC_DT_CI16_BE : SPEC_DT_CI16_LE; break;
    case SPEC_DT_CF32_LE: case SPEC_DT_CF32_BE: dt = big_endian ? SPEC_DT_CF32_BE : SPEC_DT_CF32_LE; break;
    case SPEC_DT_CF64_LE: case SPEC_DT_CF64_BE: dt = big_endian ? SPEC_DT_CF64_BE : SPEC_DT_CF64_LE; break;
    default: break;
    }
    // the reference's service reads only what its decode branch touches: nothing
    // for an unknown datatype (SS:60-63), else nfft * bytes-per-IQ from startByte
    const bool reads = kind_of(dt, c->flags) != K_ZERO;
    const uint64_t span = (uint64_t)nfft * spec_bytes_per_sample(dt);
    if (reads && (start_byte < 0 || (uint64_t)start_byte + span > capacity))
        return fail(c, SPEC_ERANGE, "IndexOutOfBounds: bytes [%lld, +%llu) of %llu", (long long)start_byte,
                    (unsigned long long)span, (unsigned long long)capacity);
    if (!reads) {  // flat 20 log10(1e-10) = -200 dB, no bytes touched
        for (uint32_t i = 0; i < nfft; ++i) out[i] = -200.0;
        return SPEC_OK;
    }
    if (c->opt_readahead_lines > 1) {
        spec_ctx::ReadAhead &ra = c->ra;
        const uint64_t sb = (uint64_t)start_byte;
        const uint8_t *bytes = static_cast<const uint8_t *>(buffer);
        if (ra.n && ra.buf == buffer && ra.capacity == capacity && ra.dt == (int)dt && ra.nfft == nfft && sb >= ra.first &&
            (sb - ra.first) % span == 0) {
            const uint64_t j = (sb - ra.first) / span;
            if (j < ra.n && memcmp(bytes + sb, ra.in.data() + j * span, span) == 0) {  // same bytes in -> same line out
                memcpy(out, ra.out.data() + j * nfft, (size_t)nfft * sizeof(double));
                ra.last_start = sb;
                return SPEC_OK;
            }
        }
        const bool sequential = ra.last_buf == buffer && ra.last_dt == (int)dt && ra.last_nfft == nfft &&
                                sb == ra.last_start + span;
        ra.last_buf = buffer; ra.last_dt = (int)dt; ra.last_nfft = nfft; ra.last_start = sb;
        if (sequential) {
            uint64_t b = (capacity - sb) / span;  // whole slices from here to the end of the buffer (>= 1)
            if (b > (uint64_t)c->opt_readahead_lines) b = (uint64_t)c->opt_readahead_lines;
            const uint64_t by_mem = (32ull << 20) / (span + (uint64_t)nfft * sizeof(double));
            if (b > by_mem) b = by_mem;
            if (b > 1) {
                ra.n = 0;
                try { ra.in.resize(b * span); ra.out.resize(b * nfft); } catch (...) { b = 1; }
            }
            if (b > 1) {
                memcpy(ra.in.data(), bytes + sb, b * span);
                spec_status st = spec_waterfall(c, ra.in.data(), 0, b * span, 0, dt, nfft, nfft, b, SPEC_WIN_RECT,
                                                SPEC_OUT_DB20_F64, -150.0, ra.out.data(), 0);
                if (st != SPEC_OK) return st;
                ra.buf = buffer; ra.capacity = capacity; ra.dt = (int)dt; ra.nfft = nfft; ra.first = sb; ra.stride = span;
                ra.n = (uint32_t)b;
                memcpy(out, ra.out.data(), (size_t)nfft * sizeof(double));
                return SPEC_OK;
            }
        }
    }
    return spec_waterfall(c, buffer, 0, capacity, (uint64_t)start_byte, dt, nfft, nfft, 1, SPEC_WIN_RECT,
                          SPEC_OUT_DB20_F64, -150.0, out, 0);
}

// Shared body of spec_welch_psd (float PSDs) and spec_welch_psd_planar_f64 (double PSDs, out_f64).
// nfft may be ANY positive integer: the reference's short-burst call passes the burst length itself
// (AnalysisDialogController.java:303-307); lengths that are not a power of two take the plain fp64 DFT.
static spec_status welch_impl(spec_ctx *c, const void *iq, int iq_on_device, uint64_t n_bytes, uint64_t start_byte,
                              uint64_t psd_stride_bytes, uint32_t n_psd, spec_dtype dt, uint32_t nfft, uint32_t hop,
                              uint32_t n_seg, spec_window window, spec_psd_scaling scaling, double fs, int db,
                              double *freq_out, void *psd_out_v, int out_on_device, bool out_f64) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!iq || !psd_out_v) return fail(c, SPEC_EINVAL, "null buffer");
    if (!dtype_valid(dt)) return fail(c, SPEC_EINVAL, "bad spec_dtype %d", dt);
    if (nfft == 0) return fail(c, SPEC_EINVAL, "nfft must be >= 1");
    if (hop == 0) return fail(c, SPEC_EINVAL, "hop must be >= 1");
    if (window != SPEC_WIN_RECT && window != SPEC_WIN_HANN) return fail(c, SPEC_EINVAL, "bad window %d", window);
    if (n_seg == 0 || n_psd == 0) return fail(c, SPEC_EINVAL, "n_seg and n_psd must be >= 1");
    if (!(fs > 0)) return fail(c, SPEC_EINVAL, "fs must be positive");
    if (scaling != SPEC_PSD_DENSITY && scaling != SPEC_PSD_SPECTRUM) return fail(c, SPEC_EINVAL, "bad scaling");
    if (nfft > 65536) return fail(c, SPEC_EUNSUPPORTED, "Welch PSD: nfft = %u is not supported (1 ... 65536)", nfft);
    const bool f64 = out_f64 || dt == SPEC_DT_CF64_LE || dt == SPEC_DT_CF64_BE;
    // lengths the FFT kernels have a plan for; everything else (not a power of two, or nfft = 1) is a plain DFT
    bool pow2 = (nfft & (nfft - 1)) == 0;
    const int log2n = pow2 ? ilog2(nfft) : 0;
    if (pow2) {
        int l1 = 0, l2 = 0;
        pow2 = plan_supported(log2n, f64) || large_split(log2n, f64, &l1, &l2);
    }
    spec_status st = SPEC_OK;
    float *psd_out = static_cast<float *>(psd_out_v);  // the fused fp32 path (never taken with out_f64)
    const size_t out_esz = out_f64 ? 8 : 4;
    const uint64_t bps = spec_bytes_per_sample(dt);
    uint64_t span = 0, total = 0;
    if (!mul_add_u64((uint64_t)(n_seg - 1) * hop + nfft, bps, 0, &span) ||
        !mul_add_u64((uint64_t)(n_psd - 1), psd_stride_bytes, span, &total))
        return fail(c, SPEC_ERANGE, "Welch PSD: %u PSDs x %u segments do not fit a 64-bit byte count", n_psd, n_seg);
    if (start_byte > n_bytes || total > n_bytes - start_byte)
        return fail(c, SPEC_ERANGE, "Welch PSD needs bytes [%llu, +%llu) of %llu", (unsigned long long)start_byte,
                    (unsigned long long)total, (unsigned long long)n_bytes);
    if (n_psd > 1 && psd_stride_bytes % component_bytes(dt) != 0)
        return fail(c, SPEC_EINVAL, "psd_stride_bytes must be a multiple of the component size");

    const uint8_t *d_in;
    if (iq_on_device) {
        d_in = static_cast<const uint8_t *>(iq) + start_byte;
        if (reinterpret_cast<uintptr_t>(d_in) % component_bytes(dt) != 0)
            return fail(c, SPEC_EINVAL, "device input is not aligned to its %u-byte components", component_bytes(dt));
    } else {
        st = grow(c, &c->stage_in, &c->stage_in_bytes, total);
        if (st != SPEC_OK) return st;
        HIP_TRY(c, hipMemcpyAsync(c->stage_in, static_cast<const uint8_t *>(iq) + start_byte, total,
                                  hipMemcpyHostToDevice, c->stream));
        d_in = static_cast<const uint8_t *>(c->stage_in);
    }
    if (!pow2) {
        // plain fp64 DFT with the exact W_N table (spec_misc.hip welch_dft_kernel)
        const void *twn = nullptr, *winn = nullptr;
        double s1 = 0, s2 = 0;
        st = get_tables_n(c, nfft, window, &twn, &winn, &s1, &s2);
        if (st != SPEC_OK) return st;
        if (!(s2 > 0)) return fail(c, SPEC_EINVAL, "the window of %u point(s) sums to zero", nfft);
        void *d_out = psd_out_v;
        if (!out_on_device) {
            st = grow(c, &c->stage_out, &c->stage_out_bytes, (size_t)n_psd * nfft * out_esz);
            if (st != SPEC_OK) return st;
            d_out = c->stage_out;
        }
        const double norm = (scaling == SPEC_PSD_DENSITY ? 1.0 / (fs * s2) : 1.0 / (s1 * s1)) / (double)n_seg;
        hipError_t e = launch_welch_dft(d_in, psd_stride_bytes, n_psd, n_seg, hop, (uint32_t)bps, kind_of(dt, c->flags),
                                        is_be(dt), nfft, twn, winn, norm, db, d_out, out_f64, c->stream);
        if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "welch DFT launch: %s", hipGetErrorString(e));
        if (!out_on_device) {
            HIP_TRY(c, hipMemcpyAsync(psd_out_v, d_out, (size_t)n_psd * nfft * out_esz, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        } else if (!iq_on_device) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        if (freq_out)
            for (uint32_t k = 0; k < nfft; ++k) freq_out[k] = ((double)k - (double)(nfft / 2)) * fs / (double)nfft;
        return SPEC_OK;
    }
    WelchArgs a{};
    a.iq = d_in;
    a.psd_stride_bytes = psd_stride_bytes;
    a.n_psd = n_psd;
    a.n_seg = n_seg;
    a.hop = hop;
    a.bps = (uint32_t)bps;
    a.kind = kind_of(dt, c->flags);
    a.be = is_be(dt);
    st = get_twiddles(c, log2n, false, &a.tw);
    if (st != SPEC_OK) return st;
    double s1 = 0, s2 = 0;
    st = get_window(c, log2n, false, window, &a.win, &s1, &s2);
    if (st != SPEC_OK) return st;
    if (!out_f64 && !c->opt_force_generic && v2_applicable(log2n, a.kind, a.be, OUT_POW_F32, n_seg, hop)) {
        // packed-fp32 family: sub-lines accumulate |X|^2 over runs of segments, one slab each
        st = get_window(c, log2n, false, window, &a.win, &s1, &s2, /*table_for_rect=*/true);
        if (st != SPEC_OK) return st;
        a.win_hann = window == SPEC_WIN_HANN ? 1 : 2;  // (2: the table of ones the rectangular Welch multiplies by)
        const uint32_t sub = (uint32_t)v2_lpw(log2n);
        // enough sub-lines to fill the chip first (a single 256-segment PSD gets one segment per
        // sub-line), long runs (register reuse, fewer slabs) once there is plenty of work
        uint64_t run = c->opt_lines_per_wg > 0 ? (uint64_t)c->opt_lines_per_wg
                                                : ((uint64_t)n_seg * n_psd) / ((uint64_t)c->n_cu * 8 * sub);
        if (run < 1) run = 1;
        if (run > 64) run = 64;
        if (run > n_seg) run = n_seg;
        // plenty of PSDs (>= two per CU) and whole-workgroup lines: one workgroup walks all segments of a PSD and
        // finishes it itself -- no slabs, no second launch
        // (a workgroup addresses its span through one buffer descriptor: 32-bit byte range).  The kernel sums |X|^2 in
        // fp32 registers over its whole run: bounded to 512 segments here (worst-case 512 eps/2 = 3e-5 relative for a
        // sum of positive terms, ~1e-6 typical); longer PSDs take the slab form -- runs of <= 64 segments summed in
        // double by the reduction -- so that the 5e-6 of the parity tests holds for any n_seg
        const bool fused = sub == 1 && c->opt_lines_per_wg <= 0 && !c->opt_welch_two_pass && (uint64_t)n_psd >= 2ull * (uint64_t)c->n_cu &&
                           n_seg <= 512 && ((uint64_t)n_seg - 1) * hop * bps + (uint64_t)nfft * bps < (1ull << 31);
        if (fused) run = n_seg;
        const uint32_t wgs = (uint32_t)((n_seg + run * sub - 1) / (run * sub));
        const uint32_t n_slabs = wgs * sub;
        if (!fused) {
            st = grow(c, &c->scratch, &c->scratch_bytes, (size_t)n_psd * n_slabs * nfft * sizeof(float));
            if (st != SPEC_OK) return st;
        }
        a.partial = c->scratch;
        float *d_out = psd_out;
        if (!out_on_device) {
            st = grow(c, &c->stage_out, &c->stage_out_bytes, (size_t)n_psd * nfft * sizeof(float));
            if (st != SPEC_OK) return st;
            d_out = static_cast<float *>(c->stage_out);
        }
        const double norm = (scaling == SPEC_PSD_DENSITY ? 1.0 / (fs * s2) : 1.0 / (s1 * s1)) / (double)n_seg;
        a.final_out = fused ? d_out : nullptr;
        a.norm = norm;
        a.db = db;
        a.rows = (int)c->opt_welch_rows;
        hipError_t e = launch_v2_welch(a, log2n, (uint32_t)run, wgs, c->stream);
        if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "welch launch: %s", hipGetErrorString(e));
        if (!fused) {
            e = launch_welch_finalize(a.partial, 0, n_psd, n_slabs, nfft, norm, db, d_out, 0, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "welch finalize launch: %s", hipGetErrorString(e));
        }
        if (!out_on_device) {
            HIP_TRY(c, hipMemcpyAsync(psd_out, d_out, (size_t)n_psd * nfft * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        } else if (!iq_on_device) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        if (freq_out)
            for (uint32_t k = 0; k < nfft; ++k) freq_out[k] = ((double)k - (double)(nfft / 2)) * fs / (double)nfft;
        return SPEC_OK;
    }
    if (f64 && !c->opt_force_generic && a.kind != K_ZERO && v3d_applicable(log2n, a.kind, n_seg, hop)) {
        // fp64 member of the family (the dialog's calculatePsdWelch call, cf64 recordings, fp64 output): |X|^2 summed
        // in registers, one double slab per sub-line, the same finalize -- one launch + finalize for any batch
        st = get_twiddles(c, log2n, true, &a.tw);
        if (st != SPEC_OK) return st;
        st = get_window(c, log2n, true, window, &a.win, &s1, &s2, /*table_for_rect=*/true);
        if (st != SPEC_OK) return st;
        a.win_hann = window == SPEC_WIN_HANN ? 1 : 2;
        const uint32_t sub = (uint32_t)v3d_lpw(log2n);
        uint64_t run = c->opt_lines_per_wg > 0 ? (uint64_t)c->opt_lines_per_wg
                                                : ((uint64_t)n_seg * n_psd) / ((uint64_t)c->n_cu * 8 * sub);
        if (run < 1) run = 1;
        if (run > 64) run = 64;
        if (run > n_seg) run = n_seg;
        const uint32_t wgs = (uint32_t)((n_seg + run * sub - 1) / (run * sub));
        const uint32_t n_slabs = wgs * sub;
        st = grow(c, &c->scratch, &c->scratch_bytes, (size_t)n_psd * n_slabs * nfft * sizeof(double));
        if (st != SPEC_OK) return st;
        a.partial = c->scratch;
        void *d_out = psd_out_v;
        if (!out_on_device) {
            st = grow(c, &c->stage_out, &c->stage_out_bytes, (size_t)n_psd * nfft * out_esz);
            if (st != SPEC_OK) return st;
            d_out = c->stage_out;
        }
        hipError_t e = launch_v3d_welch(a, log2n, (uint32_t)run, wgs, c->stream);
        if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "fp64 welch launch: %s", hipGetErrorString(e));
        const double norm = (scaling == SPEC_PSD_DENSITY ? 1.0 / (fs * s2) : 1.0 / (s1 * s1)) / (double)n_seg;
        e = launch_welch_finalize(a.partial, 1, n_psd, n_slabs, nfft, norm, db, d_out, out_f64, c->stream);
        if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "welch finalize launch: %s", hipGetErrorString(e));
        if (!out_on_device) {
            HIP_TRY(c, hipMemcpyAsync(psd_out_v, d_out, (size_t)n_psd * nfft * out_esz, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        } else if (!iq_on_device) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        if (freq_out)
            for (uint32_t k = 0; k < nfft; ++k) freq_out[k] = ((double)k - (double)(nfft / 2)) * fs / (double)nfft;
        return SPEC_OK;
    }
    // Fallback for what neither family takes (nfft < 256, 16384 points and more in fp64, 32768 and more in
    // fp32): |X|^2 lines of the spectrogram path (fp64 for cf64) summed per bin in a fixed order.
    const spec_out_fmt pfmt = f64 ? SPEC_OUT_POW_F64 : SPEC_OUT_POW_F32;
    const size_t esz = f64 ? 8 : 4;
    uint64_t seg_chunk = (128ull << 20) / ((uint64_t)nfft * esz);
    if (seg_chunk == 0) seg_chunk = 1;
    if (seg_chunk > n_seg) seg_chunk = n_seg;
    st = grow(c, &c->scratch2, &c->scratch2_bytes, (size_t)seg_chunk * nfft * esz + (size_t)nfft * sizeof(double));
    if (st != SPEC_OK) return st;
    double *acc = reinterpret_cast<double *>(static_cast<uint8_t *>(c->scratch2) + (size_t)seg_chunk * nfft * esz);
    uint8_t *d_out = static_cast<uint8_t *>(psd_out_v);
    if (!out_on_device) {
        st = grow(c, &c->stage_out, &c->stage_out_bytes, (size_t)n_psd * nfft * out_esz);
        if (st != SPEC_OK) return st;
        d_out = static_cast<uint8_t *>(c->stage_out);
    }
    st = get_window(c, log2n, f64, window, &a.win, &s1, &s2);
    if (st != SPEC_OK) return st;
    const double norm = (scaling == SPEC_PSD_DENSITY ? 1.0 / (fs * s2) : 1.0 / (s1 * s1)) / (double)n_seg;
    for (uint32_t b = 0; b < n_psd; ++b) {
        HIP_TRY(c, hipMemsetAsync(acc, 0, (size_t)nfft * sizeof(double), c->stream));
        for (uint64_t s0 = 0; s0 < n_seg; s0 += seg_chunk) {
            const uint64_t ns = n_seg - s0 < seg_chunk ? n_seg - s0 : seg_chunk;
            st = run_lines(c, d_in + (uint64_t)b * psd_stride_bytes + s0 * hop * bps, dt, log2n, hop, ns, window, pfmt,
                           c->scratch2);
            if (st != SPEC_OK) return st;
            hipError_t e = launch_welch_accum(c->scratch2, f64, ns, nfft, acc, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "welch accumulate launch: %s", hipGetErrorString(e));
        }
        hipError_t e = launch_welch_scale(acc, nfft, norm, db, d_out + (size_t)b * nfft * out_esz, out_f64, c->stream);
        if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "welch scale launch: %s", hipGetErrorString(e));
    }
    if (!out_on_device) {
        HIP_TRY(c, hipMemcpyAsync(psd_out_v, d_out, (size_t)n_psd * nfft * out_esz, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else if (!iq_on_device) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // staging buffer may be reused by the next call
    }
    if (freq_out)  // AnalysisDialogController.java:324-328 adds centerFreq to this axis
        for (uint32_t k = 0; k < nfft; ++k) freq_out[k] = ((double)k - (double)(nfft / 2)) * fs / (double)nfft;
    return SPEC_OK;
}

spec_status spec_welch_psd(spec_ctx *c, const void *iq, int iq_on_device, uint64_t n_bytes, uint64_t start_byte,
                           uint64_t psd_stride_bytes, uint32_t n_psd, spec_dtype dt, uint32_t nfft, uint32_t hop,
                           uint32_t n_seg, spec_window window, spec_psd_scaling scaling, double fs, int db,
                           double *freq_out, float *psd_out, int out_on_device) {
    return welch_impl(c, iq, iq_on_device, n_bytes, start_byte, psd_stride_bytes, n_psd, dt, nfft, hop, n_seg, window, scaling,
                      fs, db, freq_out, psd_out, out_on_device, false);
}

// The batch of spec_welch_psd over several contexts (SURVEY 8e at the C ABI, Welch side): the PSDs of a batch are
// independent, so context r computes the contiguous range spec_shard_lines(n_psd, n_ctx, r) of them -- one host thread per
// context, no exchange on the data path; a device-resident result is gathered on ctx[0]'s device with hipMemcpyPeerAsync.
spec_status spec_welch_psd_multi(spec_ctx *const *ctx, uint32_t n_ctx, const void *const *iq, int iq_on_device,
                                 const uint64_t *n_bytes, uint64_t start_byte, uint64_t psd_stride_bytes, uint32_t n_psd,
                                 spec_dtype dt, uint32_t nfft, uint32_t hop, uint32_t n_seg, spec_window window,
                                 spec_psd_scaling scaling, double fs, int db, double *freq_out, float *psd_out, int out_on_device) {
    if (!ctx || n_ctx == 0 || !ctx[0]) return SPEC_EINVAL;
    spec_ctx *root = ctx[0];
    if (n_ctx > 64) return fail(root, SPEC_EINVAL, "spec_welch_psd_multi: %u contexts (at most 64)", n_ctx);
    for (uint32_t r = 0; r < n_ctx; ++r) {
        if (!ctx[r]) return fail(root, SPEC_EINVAL, "spec_welch_psd_multi: context %u is NULL", r);
        for (uint32_t q = 0; q < r; ++q)
            if (ctx[q] == ctx[r]) return fail(root, SPEC_EINVAL, "spec_welch_psd_multi: context %u appears twice", r);
    }
    if (!iq || !n_bytes || !psd_out || (!iq_on_device && !iq[0])) return fail(root, SPEC_EINVAL, "null buffer");
    if (n_psd == 0 || n_seg == 0) return fail(root, SPEC_EINVAL, "n_seg and n_psd must be >= 1");
    if (nfft == 0) return fail(root, SPEC_EINVAL, "nfft must be >= 1");
    if (!(fs > 0)) return fail(root, SPEC_EINVAL, "fs must be positive");
    std::vector<spec_status> status(n_ctx, SPEC_OK);
    auto run = [&](uint32_t r) {
        uint64_t a = 0, b = 0;
        spec_shard_lines(n_psd, n_ctx, r, &a, &b);
        if (b <= a) return;
        spec_ctx *c = ctx[r];
        const uint32_t np = (uint32_t)(b - a);
        const size_t bytes = (size_t)np * nfft * sizeof(float);
        float *dest = psd_out + a * nfft;
        const bool via_peer = out_on_device && r != 0;  // a peer of a device-resident result computes into a buffer of its own
        if (via_peer) {
            Enter g(c);
            if ((status[r] = grow(c, &c->multi_buf, &c->multi_buf_bytes, bytes)) != SPEC_OK) return;
            dest = static_cast<float *>(c->multi_buf);
        }
        if (iq_on_device) {
            if (!iq[r]) { status[r] = fail(c, SPEC_EINVAL, "spec_welch_psd_multi: shard %u has PSDs but no buffer", r); return; }
            status[r] = welch_impl(c, iq[r], 1, n_bytes[r], 0, psd_stride_bytes, np, dt, nfft, hop, n_seg, window, scaling, fs, db,
                                   nullptr, dest, out_on_device, false);
        } else {
            status[r] = welch_impl(c, iq[0], 0, n_bytes[0], start_byte + a * psd_stride_bytes, psd_stride_bytes, np, dt, nfft, hop,
                                   n_seg, window, scaling, fs, db, nullptr, dest, out_on_device, false);
        }
        c->multi_verified = 0;        // (about THIS call, as in multi_shard)
        c->multi_peer_access = -1;
        if (status[r] != SPEC_OK || !out_on_device) return;
        Enter g(c);
        hipError_t e = hipSuccess;
        const bool verify = via_peer && (root->opt_multi_verify || c->opt_multi_verify);
        if (via_peer) {
            if ((status[r] = peer_path(c, root)) != SPEC_OK) return;
            c->multi_verified = 0;
            if (verify) {
                if ((status[r] = grow(c, &c->cks_src, &c->cks_src_bytes, 8)) != SPEC_OK) return;
                e = hipMemsetAsync(c->cks_src, 0, 8, c->stream);
                if (e == hipSuccess) e = launch_checksum(dest, bytes, static_cast<unsigned long long *>(c->cks_src), c->stream);
            }
            if (e == hipSuccess) e = hipMemcpyPeerAsync(psd_out + a * nfft, root->device, dest, c->device, bytes, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { status[r] = fail(c, SPEC_EDEVICE, "peer copy (%s): %s", peer_path_name(c), hipGetErrorString(e)); return; }
        if (verify)
            status[r] = verify_landed(c, root, {{psd_out + a * nfft, (uint64_t)bytes}}, static_cast<unsigned long long *>(c->cks_src),
                                      "spec_welch_psd_multi");
    };
    std::vector<std::thread> th;
    for (uint32_t r = 1; r < n_ctx; ++r) {
        try { th.emplace_back(run, r); }
        catch (...) { run(r); }  // no thread to be had: this shard runs here
    }
    run(0);
    for (auto &t : th) t.join();
    for (uint32_t r = 0; r < n_ctx; ++r)
        if (status[r] != SPEC_OK) {
            if (r) {
                std::string msg;
                { std::lock_guard<std::recursive_mutex> lk(ctx[r]->mu); msg = ctx[r]->err; }
                return fail(root, status[r], "shard %u of %u: %s", r, n_ctx, msg.c_str());
            }
            return status[r];
        }
    if (freq_out)  // AnalysisDialogController.java:324-328 adds centerFreq to this axis
        for (uint32_t k = 0; k < nfft; ++k) freq_out[k] = ((double)k - (double)(nfft / 2)) * fs / (double)nfft;
    return SPEC_OK;
}

// Exact shape of the call at AnalysisDialogController.java:308-312:
//   PowerSpectralDensity.calculatePsdWelch(double[][] data, double fs, int nfft)
// with data[0] = I, data[1] = Q (planar doubles, the output of the down-converter).
spec_status spec_welch_psd_planar_f64(spec_ctx *c, const double *re, const double *im, int in_on_device,
                                      uint64_t n_samples, uint32_t nfft, uint32_t hop, spec_window window,
                                      spec_psd_scaling scaling, double fs, int db, double *freq_out, double *psd_out) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!re || !im || !psd_out) return fail(c, SPEC_EINVAL, "null buffer");
    if (nfft == 0) return fail(c, SPEC_EINVAL, "nfft must be >= 1");
    if (hop == 0) return fail(c, SPEC_EINVAL, "hop must be >= 1");
    if (n_samples < nfft) return fail(c, SPEC_ERANGE, "signal of %llu samples is shorter than nfft = %u",
                                      (unsigned long long)n_samples, nfft);
    const uint64_t n_seg = (n_samples - nfft) / hop + 1, used = (n_seg - 1) * hop + nfft;
    if (n_seg > 0xFFFFFFFFull) return fail(c, SPEC_EINVAL, "too many segments");
    if (used > (1ull << 40)) return fail(c, SPEC_ERANGE, "signal of %llu samples is out of range", (unsigned long long)used);
    // the two planes go up as they are and are interleaved on the device (a host loop over 2 x 20 M doubles
    // cost 90 of the call's 99 ms)
    spec_status st = grow(c, &c->planar, &c->planar_bytes, 2 * used * sizeof(double));
    if (st != SPEC_OK) return st;
    const double *d_re = re, *d_im = im;
    if (!in_on_device) {
        st = grow(c, &c->stage_in, &c->stage_in_bytes, 2 * used * sizeof(double));
        if (st != SPEC_OK) return st;
        double *sr = static_cast<double *>(c->stage_in);
        HIP_TRY(c, hipMemcpyAsync(sr, re, used * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(sr + used, im, used * sizeof(double), hipMemcpyHostToDevice, c->stream));
        d_re = sr;
        d_im = sr + used;
    }
    hipError_t e = launch_interleave(d_re, d_im, c->planar, used, c->stream);
    if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "interleave launch: %s", hipGetErrorString(e));
    return welch_impl(c, c->planar, 1, used * 16, 0, 0, 1, SPEC_DT_CF64_LE, nfft, hop, (uint32_t)n_seg, window, scaling, fs,
                      db, freq_out, psd_out, 0, true);
}

spec_status spec_render_spectrogram(spec_ctx *c, const float *tile, int tile_on_device, uint32_t width,
                                    uint32_t nfft, uint32_t height, double fs, double min_db, double max_db,
                                    spec_colormap colormap, void *bgra_out, int out_on_device) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!tile || !bgra_out) return fail(c, SPEC_EINVAL, "null buffer");
    if (nfft == 0 || !(fs > 0)) return fail(c, SPEC_EINVAL, "nfft and fs must be positive");
    if (colormap != SPEC_CMAP_GRAYSCALE && colormap != SPEC_CMAP_HEATMAP) return fail(c, SPEC_EINVAL, "bad colormap");
    if (width == 0 || height == 0) return SPEC_OK;
    if (nfft > (1u << 24) || height > (1u << 20) || width > (1u << 24))
        return fail(c, SPEC_ERANGE, "image of %u x %u pixels over %u bins is out of range", width, height, nfft);
    const size_t in_bytes = (size_t)width * nfft * sizeof(float), out_bytes = (size_t)width * height * 4;
    const float *d_tile = tile;
    if (!tile_on_device) {
        spec_status st = grow(c, &c->stage_in, &c->stage_in_bytes, in_bytes);
        if (st != SPEC_OK) return st;
        HIP_TRY(c, hipMemcpyAsync(c->stage_in, tile, in_bytes, hipMemcpyHostToDevice, c->stream));
        d_tile = static_cast<const float *>(c->stage_in);
    }
    void *d_out = bgra_out;
    if (!out_on_device) {
        spec_status st = grow(c, &c->stage_out, &c->stage_out_bytes, out_bytes);
        if (st != SPEC_OK) return st;
        d_out = c->stage_out;
    }
    const double conversion = 10 * std::log10(fs / nfft) + 20 * std::log10((double)nfft);  // MC:1273-1274
    hipError_t e = launch_render(d_tile, width, nfft, height, conversion, min_db, max_db, (int)colormap, 0, d_out, c->stream);
    if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "render launch: %s", hipGetErrorString(e));
    if (!out_on_device) {
        HIP_TRY(c, hipMemcpyAsync(bgra_out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else if (!tile_on_device) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return SPEC_OK;
}

spec_status spec_waterfall_render(spec_ctx *c, const void *iq, int iq_on_device, uint64_t n_bytes,
                                  uint64_t start_byte, spec_dtype dt, uint32_t nfft, uint32_t hop,
                                  uint32_t n_lines, spec_window window, uint32_t height, double fs, double min_db,
                                  double max_db, spec_colormap colormap, void *bgra_out, int out_on_device) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!bgra_out) return fail(c, SPEC_EINVAL, "null buffer");
    if (n_lines == 0 || height == 0) return SPEC_OK;
    // Fused form: the image samples ONE bin per pixel row (MC:1280), so the FFT kernel stores only those
    // `height` bins of every line (a compact [n_lines][height] tile) and the colour kernel reads that.
    // Needs a kernel of the packed family with 16-point threads and a one-to-one bin -> row map.
    if (c->opt_render_fused && !c->opt_force_generic && nfft >= 256 && nfft <= 4096 && (nfft & (nfft - 1)) == 0 &&
        height <= nfft && dtype_valid(dt) && hop >= 1 && fs > 0 &&
        (colormap == SPEC_CMAP_GRAYSCALE || colormap == SPEC_CMAP_HEATMAP) &&
        v2_sel_applicable(ilog2(nfft), kind_of(dt, c->flags), is_be(dt), n_lines, hop)) {
        std::vector<int32_t> sel;
        try { sel.assign(nfft, -1); } catch (...) { return fail(c, SPEC_ENOMEM, "out of host memory"); }
        bool one_to_one = true;
        for (uint32_t f = 0; f < height && one_to_one; ++f) {
            const int bin = (int)((double)f / (double)height * (double)nfft);  // MC:1280, fftshifted index
            const uint32_t k = ((uint32_t)bin + nfft / 2) & (nfft - 1);        // SS:78 undone
            if (bin < 0 || (uint32_t)bin >= nfft || sel[k] >= 0) one_to_one = false; else sel[k] = (int32_t)f;
        }
        if (one_to_one) {
            if (c->sel_nfft != nfft || c->sel_height != height) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (c->sel_nfft != nfft) {
                    if (c->sel_dev) { (void)hipFree(c->sel_dev); c->sel_dev = nullptr; }
                    c->sel_nfft = 0;
                    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->sel_dev), nfft * sizeof(int32_t)));
                }
                HIP_TRY(c, hipMemcpy(c->sel_dev, sel.data(), nfft * sizeof(int32_t), hipMemcpyHostToDevice));
                c->sel_nfft = nfft; c->sel_height = height;
            }
            spec_status st = grow(c, &c->scratch2, &c->scratch2_bytes, (size_t)n_lines * height * sizeof(float));
            if (st != SPEC_OK) return st;
            st = waterfall_impl(c, iq, iq_on_device, n_bytes, start_byte, dt, nfft, hop, n_lines, window, SPEC_OUT_DB20_F32,
                                -150.0, c->scratch2, 1, c->sel_dev, height);
            if (st != SPEC_OK) return st;
            const size_t out_bytes = (size_t)n_lines * height * 4;
            void *d_out = bgra_out;
            if (!out_on_device) {
                st = grow(c, &c->stage_out, &c->stage_out_bytes, out_bytes);
                if (st != SPEC_OK) return st;
                d_out = c->stage_out;
            }
            const double conversion = 10 * std::log10(fs / nfft) + 20 * std::log10((double)nfft);  // MC:1273-1274
            hipError_t e = launch_render(static_cast<const float *>(c->scratch2), n_lines, nfft, height, conversion, min_db,
                                         max_db, (int)colormap, 1, d_out, c->stream);
            if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "render launch: %s", hipGetErrorString(e));
            if (!out_on_device) {
                HIP_TRY(c, hipMemcpyAsync(bgra_out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
            return SPEC_OK;
        }
    }
    // dB tile lives in the context's scratch2 and never crosses PCIe
    spec_status st = grow(c, &c->scratch2, &c->scratch2_bytes, (size_t)n_lines * nfft * sizeof(float));
    if (st != SPEC_OK) return st;
    st = spec_waterfall(c, iq, iq_on_device, n_bytes, start_byte, dt, nfft, hop, n_lines, window, SPEC_OUT_DB20_F32,
                        -150.0, c->scratch2, 1);
    if (st != SPEC_OK) return st;
    return spec_render_spectrogram(c, static_cast<const float *>(c->scratch2), 1, n_lines, nfft, height, fs, min_db,
                                   max_db, colormap, bgra_out, out_on_device);
}

// ---- burst analysis ---------------------------------------------------

// reader geometry of EDC:60-96: decode kind, byte stride between samples, bytes one sample touches
static void edc_layout(const spec_ctx *c, spec_dtype dt, int *kind, uint32_t *stride, uint32_t *width) {
    switch (dt) {
    case SPEC_DT_CI16_LE: case SPEC_DT_CI16_BE: *kind = K_CI16; *stride = *width = 4; break;
    case SPEC_DT_CU8: *kind = K_CU8; *stride = *width = 2; break;
    case SPEC_DT_CI8: *kind = K_CI8; *stride = *width = 2; break;
    case SPEC_DT_CF64_LE: case SPEC_DT_CF64_BE:
        *kind = K_CF64; *width = 16; *stride = (c->flags & SPEC_FLAG_REF_EDC_CF64_STRIDE8) ? 8 : 16; break;
    default: *kind = K_CF32; *stride = *width = 8; break;  // EDC:94-96: everything else reads as float pairs
    }
}

// the range test alone (no staging, no launch): entry points run it before they size anything from `count`
static spec_status burst_check(spec_ctx *c, uint64_t capacity, uint64_t start_sample, uint64_t count, spec_dtype dt) {
    int kind; uint32_t stride, width;
    edc_layout(c, dt, &kind, &stride, &width);
    uint64_t start_byte = 0, span = 0;
    if (count == 0 || !mul_add_u64(start_sample, stride, 0, &start_byte) || !mul_add_u64(count - 1, stride, width, &span) ||
        start_byte > capacity || span > capacity - start_byte)
        return fail(c, SPEC_ERANGE, "samples [%llu, +%llu) leave the %llu-byte buffer", (unsigned long long)start_sample,
                    (unsigned long long)count, (unsigned long long)capacity);
    return SPEC_OK;
}

// range-check the burst and make its raw bytes device-resident: *d_raw points at sample start_sample
static spec_status burst_locate(spec_ctx *c, const void *buffer, int on_device, uint64_t capacity, uint64_t start_sample,
                                uint64_t count, spec_dtype dt, const uint8_t **d_raw, int *kind, uint32_t *stride) {
    uint32_t width;
    edc_layout(c, dt, kind, stride, &width);
    // phrased without wrap-around: a huge count must be SPEC_ERANGE, never a launch
    uint64_t start_byte = 0, span = 0;
    if (count == 0 || !mul_add_u64(start_sample, *stride, 0, &start_byte) || !mul_add_u64(count - 1, *stride, width, &span) ||
        start_byte > capacity || span > capacity - start_byte)
        return fail(c, SPEC_ERANGE, "samples [%llu, +%llu) leave the %llu-byte buffer", (unsigned long long)start_sample,
                    (unsigned long long)count, (unsigned long long)capacity);
    if (on_device) {
        *d_raw = static_cast<const uint8_t *>(buffer) + start_byte;
        if (reinterpret_cast<uintptr_t>(*d_raw) % component_bytes(dt) != 0)
            return fail(c, SPEC_EINVAL, "device input is not aligned to its %u-byte components", component_bytes(dt));
    } else {
        spec_status st = grow(c, &c->stage_in, &c->stage_in_bytes, span);
        if (st != SPEC_OK) return st;
        HIP_TRY(c, hipMemcpyAsync(c->stage_in, static_cast<const uint8_t *>(buffer) + start_byte, span,
                                  hipMemcpyHostToDevice, c->stream));
        *d_raw = static_cast<const uint8_t *>(c->stage_in);
    }
    return SPEC_OK;
}

// decode (+ mix) `count` samples into device doubles d_re / d_im
static spec_status burst_read(spec_ctx *c, const void *buffer, int on_device, uint64_t capacity, uint64_t start_sample,
                              uint64_t count, spec_dtype dt, double freq_off, double *d_re, double *d_im) {
    const uint8_t *d_raw; int kind; uint32_t stride;
    spec_status st = burst_locate(c, buffer, on_device, capacity, start_sample, count, dt, &d_raw, &kind, &stride);
    if (st != SPEC_OK) return st;
    hipError_t e = launch_extract_mix(d_raw, kind, is_be(dt), stride, count, freq_off, d_re, d_im, c->stream);
    if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "reader launch: %s", hipGetErrorString(e));
    return SPEC_OK;
}

spec_status spec_extract_iq(spec_ctx *c, const void *buffer, int buffer_on_device, uint64_t capacity,
                            uint64_t start_sample, uint64_t count, spec_dtype dt, double *re, double *im,
                            int out_on_device) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!dtype_valid(dt)) return fail(c, SPEC_EINVAL, "bad datatype %d", (int)dt);
    if (count == 0) return SPEC_OK;
    if (!buffer || !re || !im) return fail(c, SPEC_EINVAL, "null buffer");
    {
        spec_status rst = burst_check(c, capacity, start_sample, count, dt);
        if (rst != SPEC_OK) return rst;
    }
    double *d_re = re, *d_im = im;
    if (!out_on_device) {
        spec_status st = grow(c, &c->stage_out, &c->stage_out_bytes, 2 * count * sizeof(double));
        if (st != SPEC_OK) return st;
        d_re = static_cast<double *>(c->stage_out);
        d_im = d_re + count;
    }
    spec_status st = burst_read(c, buffer, buffer_on_device, capacity, start_sample, count, dt, 0.0, d_re, d_im);
    if (st != SPEC_OK) return st;
    if (!out_on_device) {
        HIP_TRY(c, hipMemcpyAsync(re, d_re, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(im, d_im, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    if (!out_on_device || !buffer_on_device) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPEC_OK;
}

spec_status spec_down_convert(spec_ctx *c, const void *buffer, int buffer_on_device, uint64_t capacity,
                              uint64_t start_sample, uint64_t count, spec_dtype dt, double freq_off, uint32_t down,
                              spec_downconv_mode mode, double *re_out, double *im_out, int out_on_device) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!dtype_valid(dt)) return fail(c, SPEC_EINVAL, "bad datatype %d", (int)dt);
    if (down == 0 || down > (1u << 20)) return fail(c, SPEC_EINVAL, "down = %u out of range", down);
    if (mode != SPEC_DC_FAST && mode != SPEC_DC_LPF) return fail(c, SPEC_EINVAL, "bad mode %d", (int)mode);
    if (!std::isfinite(freq_off)) return fail(c, SPEC_EINVAL, "freq_off must be finite");
    const uint64_t n_out = count / down;
    if (n_out == 0) return SPEC_OK;
    if (!buffer || !re_out || !im_out) return fail(c, SPEC_EINVAL, "null buffer");
    {
        spec_status rst = burst_check(c, capacity, start_sample, count, dt);
        if (rst != SPEC_OK) return rst;
    }
    double *d_or = re_out, *d_oi = im_out;
    if (!out_on_device) {
        spec_status st = grow(c, &c->stage_out, &c->stage_out_bytes, 2 * n_out * sizeof(double));
        if (st != SPEC_OK) return st;
        d_or = static_cast<double *>(c->stage_out);
        d_oi = d_or + n_out;
    }
    if (mode == SPEC_DC_FAST) {  // every sample feeds exactly one output: reader, mixer and boxcar in one pass
        const uint8_t *d_raw; int kind; uint32_t stride;
        spec_status st = burst_locate(c, buffer, buffer_on_device, capacity, start_sample, count, dt, &d_raw, &kind, &stride);
        if (st != SPEC_OK) return st;
        hipError_t e = launch_boxcar_decim(d_raw, kind, is_be(dt), stride, freq_off, down, d_or, d_oi, n_out, c->stream);
        if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "filter launch: %s", hipGetErrorString(e));
    } else {
        // taps of the stated specification (include/specgpu.h), fp64 on the host
        const uint32_t K = 8 * down + 1, centre = 4 * down;
        std::vector<double> h;
        try { h.resize(K); } catch (...) { return fail(c, SPEC_ENOMEM, "out of host memory for %u filter taps", K); }
        double sum = 0.0;
        for (uint32_t k = 0; k < K; ++k) {
            const double x = ((double)k - 4.0 * down) / (double)down;
            const double sinc = x == 0.0 ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
            h[k] = sinc * (0.54 - 0.46 * std::cos(2.0 * M_PI * (double)k / (double)(K - 1)));
            sum += h[k];
        }
        for (uint32_t k = 0; k < K; ++k) h[k] /= sum;
        const bool one_pass = mix_fir_applicable(K, down);  // the span of a workgroup's outputs fits the LDS
        // scratch: taps (+ the mixed samples, planar, when the filter is too long for the one-pass kernel)
        spec_status st = grow(c, &c->scratch, &c->scratch_bytes, ((one_pass ? 0 : 2 * count) + K) * sizeof(double));
        if (st != SPEC_OK) return st;
        double *d_h = static_cast<double *>(c->scratch), *d_mr = d_h + K, *d_mi = d_mr + count;
        HIP_TRY(c, hipMemcpyAsync(d_h, h.data(), K * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (one_pass) {
            const uint8_t *d_raw; int kind; uint32_t stride;
            st = burst_locate(c, buffer, buffer_on_device, capacity, start_sample, count, dt, &d_raw, &kind, &stride);
            if (st == SPEC_OK) {
                hipError_t e = launch_mix_fir(d_raw, kind, is_be(dt), stride, count, freq_off, d_h, K, centre, down, d_or, d_oi,
                                              n_out, c->stream);
                if (e != hipSuccess) st = fail(c, SPEC_EDEVICE, "filter launch: %s", hipGetErrorString(e));
            }
        } else {
            st = burst_read(c, buffer, buffer_on_device, capacity, start_sample, count, dt, freq_off, d_mr, d_mi);
            if (st == SPEC_OK) {
                hipError_t e = launch_fir_decim(d_mr, d_mi, count, d_h, K, centre, down, d_or, d_oi, n_out, c->stream);
                if (e != hipSuccess) st = fail(c, SPEC_EDEVICE, "filter launch: %s", hipGetErrorString(e));
            }
        }
        (void)hipStreamSynchronize(c->stream);  // `h` (pageable host memory) must outlive its copy
        if (st != SPEC_OK) return st;
    }
    if (!out_on_device) {
        HIP_TRY(c, hipMemcpyAsync(re_out, d_or, n_out * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(im_out, d_oi, n_out * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    if (!out_on_device || !buffer_on_device) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPEC_OK;
}

// shared body of the two traces: kind_trace 0 = magnitude (n outputs), 1 = frequency (n - 1 outputs)
static spec_status run_trace(spec_ctx *c, int kind_trace, const double *re, const double *im, int in_on_device,
                             uint64_t n, double alpha, double fs, double add, double *out, int out_on_device) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    const uint64_t n_out = kind_trace == 0 ? n : (n > 0 ? n - 1 : 0);
    if (n_out == 0) return SPEC_OK;
    if (!re || !im || !out) return fail(c, SPEC_EINVAL, "null buffer");
    if (std::isnan(alpha)) return fail(c, SPEC_EINVAL, "alpha is NaN");
    if (n > (1ull << 40)) return fail(c, SPEC_ERANGE, "trace of %llu samples is out of range", (unsigned long long)n);
    const double *d_re = re, *d_im = im;
    if (!in_on_device) {
        spec_status st = grow(c, &c->stage_in, &c->stage_in_bytes, 2 * n * sizeof(double));
        if (st != SPEC_OK) return st;
        double *s = static_cast<double *>(c->stage_in);
        HIP_TRY(c, hipMemcpyAsync(s, re, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(s + n, im, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        d_re = s;
        d_im = s + n;
    }
    double *d_out = out;
    if (!out_on_device) {
        spec_status st = grow(c, &c->stage_out, &c->stage_out_bytes, n_out * sizeof(double));
        if (st != SPEC_OK) return st;
        d_out = static_cast<double *>(c->stage_out);
    }
    spec_status st = grow(c, &c->scratch2, &c->scratch2_bytes, trace_scratch_bytes(n_out));
    if (st != SPEC_OK) return st;
    hipError_t e = launch_trace(kind_trace, d_re, d_im, n_out, alpha, fs, add, c->scratch2, d_out, c->stream);
    if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "trace launch: %s", hipGetErrorString(e));
    if (!out_on_device) HIP_TRY(c, hipMemcpyAsync(out, d_out, n_out * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (!out_on_device || !in_on_device) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPEC_OK;
}

spec_status spec_magnitude_trace(spec_ctx *c, const double *re, const double *im, int in_on_device, uint64_t n,
                                 double alpha, double *db_out, int out_on_device) {
    return run_trace(c, 0, re, im, in_on_device, n, alpha, 0.0, 0.0, db_out, out_on_device);
}

spec_status spec_inst_freq_trace(spec_ctx *c, const double *re, const double *im, int in_on_device, uint64_t n,
                                 double alpha, double fs, double center_freq, double *hz_out, int out_on_device) {
    return run_trace(c, 1, re, im, in_on_device, n, alpha, fs, center_freq, hz_out, out_on_device);
}

spec_status spec_synth_iq(spec_ctx *c, void *dev_out, spec_dtype dt, uint64_t seed, uint64_t first_sample,
                          uint64_t n_samples) {
    if (!c) return SPEC_EINVAL;
    Enter g(c);
    if (!dev_out) return fail(c, SPEC_EINVAL, "null buffer");
    if (!dtype_valid(dt) || dt == SPEC_DT_UNKNOWN) return fail(c, SPEC_EINVAL, "bad spec_dtype %d", dt);
    hipError_t e = launch_synth(dev_out, kind_of(dt, 0), is_be(dt), seed, first_sample, n_samples, c->stream);
    if (e != hipSuccess) return fail(c, SPEC_EDEVICE, "synth launch: %s", hipGetErrorString(e));
    return SPEC_OK;
}

}  // extern "C"
