// spec_misc.hip -- small helper kernels: EOF fill, Welch slab reduction,
// synthetic IQ generator.
#include "spec_fft.h"
#include "spec_internal.h"

namespace specgpu {

// MainController.java:994-998 -- lines past the end of the recording
template <typename T> __global__ void fill_kernel(T *out, uint64_t n, T value) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = value;
}

hipError_t launch_fill(void *out, uint64_t n, double value, int is_f64, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
    if (is_f64) hipLaunchKernelGGL(fill_kernel<double>, dim3(blocks), dim3(256), 0, s, static_cast<double *>(out), n, value);
    else hipLaunchKernelGGL(fill_kernel<float>, dim3(blocks), dim3(256), 0, s, static_cast<float *>(out), n, (float)value);
    return hipGetLastError();
}

// Position-weighted checksum of a run of 32-bit words, added (mod 2^64) into *out: sum_i word_i * (2 i + 1).  Used by
// the "multi_verify" option of spec_waterfall_multi / spec_welch_psd_multi: the same number over a piece before it
// leaves a peer's device and over the rows it landed in on the consumer's device (a moved, truncated or stale piece
// changes it; the order of the partial sums does not).
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t *__restrict__ w, uint64_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        acc += (unsigned long long)w[i] * (2ull * i + 1ull);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

hipError_t launch_checksum(const void *p, uint64_t n_bytes, unsigned long long *out, hipStream_t s) {
    const uint64_t n = n_bytes / 4;
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(checksum_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint32_t *>(p), n, out);
    return hipGetLastError();
}

// Sum the partial slabs of each PSD, scale, fftshift, optional 10 log10(P + 1e-20).  A block
// owns 32 bins; its 8 lanes per bin each sum every 8th slab and are combined in lane order, so
// the result does not depend on scheduling (bitwise reproducible, no atomics) while a single
// 256-slab PSD is no longer one long dependent chain per bin.
template <typename TS, typename TO>
__global__ __launch_bounds__(256) void welch_finalize_kernel(const TS *__restrict__ partial, uint32_t n_slabs,
                                                             uint32_t nfft, double norm, int db,
                                                             TO *__restrict__ psd_out) {
    __shared__ double part[8][32];
    const uint32_t psd = blockIdx.y, b = threadIdx.x & 31, lane = threadIdx.x >> 5;
    const uint32_t k = blockIdx.x * 32 + b;
    double acc = 0;
    if (k < nfft) {
        const TS *p = partial + (uint64_t)psd * n_slabs * nfft + k;
#pragma unroll 4
        for (uint32_t sl = lane; sl < n_slabs; sl += 8) acc += (double)p[(uint64_t)sl * nfft];
    }
    part[lane][b] = acc;
    __syncthreads();
    if (lane == 0 && k < nfft) {
        double t = part[0][b];
#pragma unroll
        for (int l = 1; l < 8; ++l) t += part[l][b];
        const double v = t * norm;
        const uint32_t ks = (k + nfft / 2) & (nfft - 1);
        psd_out[(uint64_t)psd * nfft + ks] = db ? (TO)(10.0 * log10(v + 1e-20)) : (TO)v;
    }
}

template <typename TS, typename TO>
static void finalize_launch(const void *partial, uint32_t n_psd, uint32_t n_slabs, uint32_t nfft, double norm, int db,
                            void *psd_out, hipStream_t s) {
    for (uint32_t p0 = 0; p0 < n_psd; p0 += 65535) {  // grid.y limit
        const uint32_t np = n_psd - p0 < 65535 ? n_psd - p0 : 65535;
        hipLaunchKernelGGL((welch_finalize_kernel<TS, TO>), dim3((nfft + 31) / 32, np), dim3(256), 0, s,
                           static_cast<const TS *>(partial) + (uint64_t)p0 * n_slabs * nfft, n_slabs, nfft, norm, db,
                           static_cast<TO *>(psd_out) + (uint64_t)p0 * nfft);
    }
}
hipError_t launch_welch_finalize(const void *partial, int slabs_f64, uint32_t n_psd, uint32_t n_slabs, uint32_t nfft,
                                 double norm, int db, void *psd_out, int out_f64, hipStream_t s) {
    if (slabs_f64) {
        if (out_f64) finalize_launch<double, double>(partial, n_psd, n_slabs, nfft, norm, db, psd_out, s);
        else finalize_launch<double, float>(partial, n_psd, n_slabs, nfft, norm, db, psd_out, s);
    } else {
        if (out_f64) finalize_launch<float, double>(partial, n_psd, n_slabs, nfft, norm, db, psd_out, s);
        else finalize_launch<float, float>(partial, n_psd, n_slabs, nfft, norm, db, psd_out, s);
    }
    return hipGetLastError();
}

// ---- fallback Welch (any datatype / size the packed family does not take): mean of power lines
// 32 bins x 8 line-lanes per workgroup: lane g adds the lines l = g, g + 8, ... of its bin in order, the eight
// partial sums are then added in lane order -- a fixed order (reproducible), eight times more waves and chains an
// eighth as long as one thread per bin walking every line (that form took 0.8 ms per 2048 lines of 8192 bins,
// six times the FFT that produced them)
template <typename T>
__global__ __launch_bounds__(256) void welch_accum_kernel(const T *__restrict__ lines, uint64_t n, uint32_t nfft,
                                                          double *__restrict__ acc) {
    __shared__ double part[8][33];
    const uint32_t b = threadIdx.x & 31, g = threadIdx.x >> 5;
    const uint32_t k = blockIdx.x * 32 + b;
    double a = 0.0;
    if (k < nfft)
        for (uint64_t l = g; l < n; l += 8) a += (double)lines[l * nfft + k];
    part[g][b] = a;
    __syncthreads();
    if (g == 0 && k < nfft) {
        double t = acc[k];
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][b];
        acc[k] = t;
    }
}
__global__ void welch_scale_kernel(const double *__restrict__ acc, uint32_t nfft, double norm, int db,
                                   void *__restrict__ out, int out_f64) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nfft) return;
    const double v = acc[k] * norm, r = db ? 10.0 * log10(v + 1e-20) : v;
    if (out_f64) static_cast<double *>(out)[k] = r;
    else static_cast<float *>(out)[k] = (float)r;
}
hipError_t launch_welch_accum(const void *lines, int lines_f64, uint64_t n, uint32_t nfft, double *acc, hipStream_t s) {
    if (lines_f64) hipLaunchKernelGGL(welch_accum_kernel<double>, dim3((nfft + 31) / 32), dim3(256), 0, s,
                                      static_cast<const double *>(lines), n, nfft, acc);
    else hipLaunchKernelGGL(welch_accum_kernel<float>, dim3((nfft + 31) / 32), dim3(256), 0, s,
                            static_cast<const float *>(lines), n, nfft, acc);
    return hipGetLastError();
}
hipError_t launch_welch_scale(const double *acc, uint32_t nfft, double norm, int db, void *psd_out, int out_f64,
                              hipStream_t s) {
    hipLaunchKernelGGL(welch_scale_kernel, dim3((nfft + 255) / 256), dim3(256), 0, s, acc, nfft, norm, db, psd_out, out_f64);
    return hipGetLastError();
}

// ---- Welch PSD of an arbitrary (non power-of-two) length: plain DFT in fp64 -------------------------
// AnalysisDialogController.java:303-307 hands calculatePsdWelch the burst length itself whenever the
// down-converted burst is shorter than 8192 samples -- any integer, one segment.  X[k] = sum_n x[n] w[n]
// W_N^(nk mod N) with the exact table W_N^m (long double on the host, rounded once) and the index kept
// by modular addition, so the only error is the fp64 summation (~sqrt(N) eps).  One thread per bin k; the
// segment's decoded, windowed samples pass through LDS 256 at a time.  O(N^2) per segment: meant for the
// dialog's short bursts (N <= 8191: 67 M complex multiply-adds), correct for any N the table fits.
__global__ __launch_bounds__(256) void welch_dft_kernel(const uint8_t *__restrict__ iq, uint64_t psd_stride_bytes,
                                                        uint32_t n_seg, uint32_t hop, uint32_t bps, int kind, int be,
                                                        uint32_t N, const cx<double> *__restrict__ tw,
                                                        const double *__restrict__ win, double norm, int db, void *out,
                                                        int out_f64) {
    __shared__ cx<double> xs[256];
    const uint32_t tid = threadIdx.x, k = blockIdx.x * 256 + tid, psd = blockIdx.y;
    const uint8_t *base = iq + (uint64_t)psd * psd_stride_bytes;
    double acc = 0.0;
    for (uint32_t seg = 0; seg < n_seg; ++seg) {
        const uint8_t *sb = base + (uint64_t)seg * hop * bps;
        double ar = 0.0, ai = 0.0;
        uint32_t m = 0;  // (n k) mod N
        for (uint32_t n0 = 0; n0 < N; n0 += 256) {
            __syncthreads();
            if (n0 + tid < N) {
                cx<double> x = decode_sample<double>(sb + (uint64_t)(n0 + tid) * bps, kind, be != 0);
                if (win) { const double w = win[n0 + tid]; x.x *= w; x.y *= w; }
                xs[tid] = x;
            }
            __syncthreads();
            const uint32_t cnt = N - n0 < 256 ? N - n0 : 256;
            if (k < N) {
                for (uint32_t j = 0; j < cnt; ++j) {
                    const cx<double> w = tw[m], x = xs[j];
                    ar += x.x * w.x - x.y * w.y;
                    ai += x.x * w.y + x.y * w.x;
                    m += k;
                    if (m >= N) m -= N;
                }
            }
        }
        acc += ar * ar + ai * ai;
    }
    if (k < N) {
        const double v = acc * norm;
        const double r = db ? 10.0 * log10(v + 1e-20) : v;
        const uint64_t o = (uint64_t)psd * N + (k + N / 2) % N;  // numpy.fft.fftshift for odd N too
        if (out_f64) static_cast<double *>(out)[o] = r;
        else static_cast<float *>(out)[o] = (float)r;
    }
}

hipError_t launch_welch_dft(const uint8_t *iq, uint64_t psd_stride_bytes, uint32_t n_psd, uint32_t n_seg, uint32_t hop,
                            uint32_t bps, int kind, int be, uint32_t nfft, const void *tw, const void *win, double norm,
                            int db, void *out, int out_f64, hipStream_t s) {
    const size_t esz = out_f64 ? 8 : 4;
    for (uint32_t p0 = 0; p0 < n_psd; p0 += 65535) {  // grid.y limit
        const uint32_t np = n_psd - p0 < 65535 ? n_psd - p0 : 65535;
        hipLaunchKernelGGL(welch_dft_kernel, dim3((nfft + 255) / 256, np), dim3(256), 0, s,
                           iq + (uint64_t)p0 * psd_stride_bytes, psd_stride_bytes, n_seg, hop, bps, kind, be, nfft,
                           static_cast<const cx<double> *>(tw), static_cast<const double *>(win), norm, db,
                           static_cast<uint8_t *>(out) + (uint64_t)p0 * nfft * esz, out_f64);
    }
    return hipGetLastError();
}

// ---- renderSpectrogram + getColorForMagnitude (MC:1261-1291, MC:926-957): dB tile -> BGRA8 ----
// The tile is [x][bin] (one line per column x), the image [y][x]: a 32 x 32 block of pixels is read along
// the bins (coalesced in the tile), coloured, turned in LDS and written along x (coalesced in the image) --
// one thread per pixel in image order read the tile with a stride of a whole line per lane and ran at a
// quarter of this rate.  The arithmetic follows the Java expressions operation by operation (double for
// the bin / normalisation, float for Color.interpolate, round-half-up to 8 bits); the _rn intrinsics keep
// hipcc from fusing them.
__device__ __forceinline__ uint32_t render_pixel(float tv, double conversion, double min_db, double max_db, int colormap) {
    const double db = __dsub_rn((double)tv, conversion);                           // MC:1283
    double n = __ddiv_rn(__dsub_rn(db, min_db), __dsub_rn(max_db, min_db));        // MC:929
    n = n < 0.0 ? 0.0 : (n > 1.0 ? 1.0 : n);                                       // MC:930
    float r, g, b;
    if (colormap == 1) {                                                           // Heatmap MC:943-953
        if (n < 0.2) { r = g = b = 0.0f; }
        else if (n < 0.5) {
            const double tt = __ddiv_rn(__dsub_rn(n, 0.2), 0.3);
            if (tt <= 0.0) { r = 0; g = 0; b = 1; } else if (tt >= 1.0) { r = 1; g = 0; b = 0; }
            else { const float ft = (float)tt; r = ft; g = 0.0f; b = __fadd_rn(1.0f, __fmul_rn(-1.0f, ft)); }
        } else {
            const double tt = __ddiv_rn(__dsub_rn(n, 0.5), 0.5);
            if (tt <= 0.0) { r = 1; g = 0; b = 0; } else if (tt >= 1.0) { r = 1; g = 1; b = 0; }
            else { r = 1.0f; g = (float)tt; b = 0.0f; }
        }
    } else {                                                                       // Grayscale MC:939-941
        r = n <= 0.0 ? 0.0f : (n >= 1.0 ? 1.0f : (float)n);
        g = b = r;
    }
    auto ch = [](float c) { return (uint32_t)(unsigned char)floor(__dadd_rn(__dmul_rn((double)c, 255.0), 0.5)); };
    return ch(b) | (ch(g) << 8) | (ch(r) << 16) | 0xFF000000u;                     // B, G, R, A bytes
}

__global__ __launch_bounds__(256) void render_kernel(const float *__restrict__ tile, uint32_t width, uint32_t nfft,
                                                     uint32_t height, double conversion, double min_db, double max_db,
                                                     int colormap, int compact, uint32_t *__restrict__ out) {
    __shared__ uint32_t px[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint32_t x0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // lanes along f: neighbouring bins of one line
        const uint32_t x = x0 + ty + 8 * j, f = f0 + tx;
        if (x < width && f < height) {
            const int bin = (int)__dmul_rn(__ddiv_rn((double)f, (double)height), (double)nfft);  // MC:1280
            // compact tile: the FFT kernel stored only the sampled bins, bin(f) at column f
            const float tv = compact ? tile[(uint64_t)x * height + f] : tile[(uint64_t)x * nfft + bin];
            px[ty + 8 * j][tx] = render_pixel(tv, conversion, min_db, max_db, colormap);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // lanes along x: neighbouring pixels of one image row
        const uint32_t x = x0 + tx, f = f0 + ty + 8 * j;
        if (x < width && f < height) out[(uint64_t)(height - 1 - f) * width + x] = px[tx][ty + 8 * j];  // MC:1288
    }
}

// planar doubles (the reference's double[2][N], ADC:216,298) -> interleaved cf64
__global__ void interleave_kernel(const double *__restrict__ re, const double *__restrict__ im,
                                  double2 *__restrict__ out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = make_double2(re[i], im[i]);
}

hipError_t launch_interleave(const double *re, const double *im, void *out, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t wgs = (n + 255) / 256;
    hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(256), 0, s, re, im,
                       static_cast<double2 *>(out), n);
    return hipGetLastError();
}

hipError_t launch_render(const float *tile, uint32_t width, uint32_t nfft, uint32_t height, double conversion,
                         double min_db, double max_db, int colormap, int compact, void *bgra, hipStream_t s) {
    if (width == 0 || height == 0) return hipSuccess;
    hipLaunchKernelGGL(render_kernel, dim3((width + 31) / 32, (height + 31) / 32), dim3(256), 0, s, tile, width, nfft, height,
                       conversion, min_db, max_db, colormap, compact, static_cast<uint32_t *>(bgra));
    return hipGetLastError();
}

// ---- synthetic IQ: two tones + Gaussian noise, counter based ----------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

__global__ void synth_kernel(uint8_t *out, int kind, int be, uint64_t seed, uint64_t first, uint64_t n) {
    constexpr uint32_t INC1 = 528280977u;   // round(0.123 * 2^32)
    constexpr uint32_t INC2 = 2963527434u;  // round((1 - 0.31) * 2^32)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = first + i;
        const uint64_t a = splitmix64(seed ^ (2 * s)), b = splitmix64(seed ^ (2 * s + 1));
        const float u1 = ((float)(uint32_t)(a >> 40) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(uint32_t)(b >> 40) + 0.5f) * (1.0f / 16777216.0f);
        const float r = 0.05f * sqrtf(-2.0f * logf(u1));
        const uint32_t p1 = (uint32_t)(s * (uint64_t)INC1), p2 = (uint32_t)(s * (uint64_t)INC2);
        float sn, cn, s1, c1, s2, c2;
        sincospif(2.0f * u2, &sn, &cn);
        sincospif((float)((double)p1 * (2.0 / 4294967296.0)), &s1, &c1);
        sincospif((float)((double)p2 * (2.0 / 4294967296.0)), &s2, &c2);
        float v[2] = {r * cn + 0.5f * c1 + 0.1f * c2, r * sn + 0.5f * s1 + 0.1f * s2};
        for (int c = 0; c < 2; ++c) {
            const float x = v[c], xc = fminf(1.0f, fmaxf(-1.0f, x));
            switch (kind) {
            case K_CF32: {
                uint32_t u = __float_as_uint(x);
                if (be) u = __builtin_bswap32(u);
                reinterpret_cast<uint32_t *>(out)[2 * i + c] = u;
                break;
            }
            case K_CF64: {
                uint64_t u = (uint64_t)__double_as_longlong((double)x);
                if (be) u = __builtin_bswap64(u);
                reinterpret_cast<uint64_t *>(out)[2 * i + c] = u;
                break;
            }
            case K_CI16: {
                uint16_t u = (uint16_t)(int16_t)__float2int_rn(32767.0f * xc);
                if (be) u = __builtin_bswap16(u);
                reinterpret_cast<uint16_t *>(out)[2 * i + c] = u;
                break;
            }
            case K_CI8: out[2 * i + c] = (uint8_t)(int8_t)__float2int_rn(127.0f * xc); break;
            case K_CU8: out[2 * i + c] = (uint8_t)__float2int_rn(127.5f + 127.0f * xc); break;
            default: break;
            }
        }
    }
}

hipError_t launch_synth(void *out, int kind, int be, uint64_t seed, uint64_t first, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
    hipLaunchKernelGGL(synth_kernel, dim3(blocks), dim3(256), 0, s, static_cast<uint8_t *>(out), kind, be, seed,
                       first, n);
    return hipGetLastError();
}

}  // namespace specgpu
