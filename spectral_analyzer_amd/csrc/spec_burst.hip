// spec_burst.hip -- the Analysis dialog's burst chain, fp64:
//   reader + mixer   ExtractDownConvertService.java:60-97 (+ the frequency shift of EDC:104-113)
//   FIR + decimate   the build's own down-converter specification (JDSP Resampler is absent)
//   magnitude trace  AnalysisDialogController.java:219-246   hypot -> EMA -> 20 log10
//   frequency trace  AnalysisDialogController.java:256-284   phase step -> wrap -> Hz -> EMA
// The exponential moving average v[i] = alpha x[i] + (1 - alpha) v[i-1] is a first-order linear
// recurrence: every element is the affine map v -> a v + b, maps compose associatively, so the
// traces are three small kernels (tile aggregates, carries, apply) instead of one serial loop.
#include "spec_fft.h"
#include "spec_internal.h"

namespace specgpu {

namespace {

constexpr int TR_THREADS = 256, TR_ITEMS = 8, TR_TILE = TR_THREADS * TR_ITEMS;
constexpr double kPi = 3.14159265358979323846;

// ---- reader (EDC:60-97) + optional mixer ------------------------------------------------------
__device__ __forceinline__ uint64_t ld_u64(const uint8_t *p, bool be) {
    const uint64_t u = *reinterpret_cast<const uint64_t *>(p);
    return be ? __builtin_bswap64(u) : u;
}

// sample i of the burst, decoded (EDC:79-96) and shifted by freq_off cycles per sample
__device__ __forceinline__ cx<double> read_mixed(const uint8_t *__restrict__ raw, int kind, int be, uint32_t stride,
                                                 uint64_t i, double freq_off) {
    const uint8_t *p = raw + i * stride;
    double x, y;
    if (kind == K_CF64) {  // two 8-byte loads: the reference's 8-byte stride (EDC:60-67) is not 16-byte aligned
        x = __longlong_as_double((long long)ld_u64(p, be));
        y = __longlong_as_double((long long)ld_u64(p + 8, be));
    } else {
        const cx<double> z = decode_sample<double>(p, kind, be);
        x = z.x;
        y = z.y;
    }
    if (freq_off != 0.0) {  // x exp(-2 pi i frac(freq_off n))
        // the phase is the ROUNDED product minus its floor (the serial definition); the empty asm keeps
        // the backend (-ffp-contract=fast) from fusing the multiply into the subtraction
        double t = freq_off * (double)i;
        asm volatile("" : "+v"(t));
        t -= floor(t);
        const double a = 2.0 * kPi * t;
        double sn, cs;
        sincos(a, &sn, &cs);  // one argument reduction for both
        const double xr = x * cs + y * sn, xi = y * cs - x * sn;
        x = xr;
        y = xi;
    }
    return {x, y};
}

__global__ __launch_bounds__(256) void extract_mix_kernel(const uint8_t *__restrict__ raw, int kind, int be,
                                                          uint32_t stride, uint64_t count, double freq_off,
                                                          double *__restrict__ re, double *__restrict__ im) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const cx<double> z = read_mixed(raw, kind, be, stride, i, freq_off);
        re[i] = z.x;
        im[i] = z.y;
    }
}

// "fast" mode in one pass (every sample is used by exactly one output): reader, mixer and the
// boxcar of `down` taps, summed in the tap order of the definition (k ascending = index descending)
__global__ __launch_bounds__(256) void boxcar_decim_kernel(const uint8_t *__restrict__ raw, int kind, int be,
                                                           uint32_t stride, double freq_off, uint32_t down,
                                                           double *__restrict__ ore, double *__restrict__ oim,
                                                           uint64_t n_out) {
    const double h = 1.0 / (double)down;
    for (uint64_t m = (uint64_t)blockIdx.x * 256 + threadIdx.x; m < n_out; m += (uint64_t)gridDim.x * 256) {
        double ar = 0.0, ai = 0.0;
        for (uint32_t k = 0; k < down; ++k) {
            const cx<double> z = read_mixed(raw, kind, be, stride, m * down + (down - 1 - k), freq_off);
            ar += h * z.x;
            ai += h * z.y;
        }
        ore[m] = ar;
        oim[m] = ai;
    }
}

// y[m] = sum_k h[k] xm[m down + c - k], zero outside [0, n).  G adjacent lanes share one output: lane g takes
// the taps g, g + G, g + 2G, ... so the group reads G consecutive samples per step (one thread per output
// would read with a stride of `down` doubles per lane), then log2(G) butterfly steps add the partial sums.
// A tree, not the tap-order loop of the definition: equal to it to rounding (|dy| <= 1e-12 max|x| in the
// tests, measured ~1e-16).
template <int G>
__global__ __launch_bounds__(256) void fir_decim_kernel(const double *__restrict__ mr, const double *__restrict__ mi,
                                                        uint64_t n, const double *__restrict__ h, uint32_t K,
                                                        uint32_t c, uint32_t down, double *__restrict__ ore,
                                                        double *__restrict__ oim, uint64_t n_out) {
    constexpr int PER_WG = 256 / G;  // outputs per workgroup and step
    const int g = threadIdx.x % G, sub = threadIdx.x / G;
    for (uint64_t m0 = (uint64_t)blockIdx.x * PER_WG; m0 < n_out; m0 += (uint64_t)gridDim.x * PER_WG) {
        const uint64_t m = m0 + sub;
        double ar = 0.0, ai = 0.0;
        if (m < n_out) {
            const int64_t top = (int64_t)(m * down) + (int64_t)c;  // index read by tap 0
            for (uint32_t k = g; k < K; k += G) {
                const int64_t idx = top - (int64_t)k;
                if (idx >= 0 && (uint64_t)idx < n) {
                    const double hk = h[k];
                    ar += hk * mr[idx];
                    ai += hk * mi[idx];
                }
            }
        }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            ar += __shfl_xor(ar, off, 64);
            ai += __shfl_xor(ai, off, 64);
        }
        if (g == 0 && m < n_out) {
            ore[m] = ar;
            oim[m] = ai;
        }
    }
}

// Conventional mode in one pass: a workgroup decodes and mixes the span its outputs need straight into LDS
// (planar, 16 KiB + 16 KiB for 2048 samples) and runs the grouped FIR from there -- no scratch round trip of the
// mixed burst.  Spans of neighbouring workgroups overlap by K - down samples (3 % at down = 8).
constexpr int FIR_SPAN = 2048;  // samples of the mixed burst held per workgroup

template <int G>
__global__ __launch_bounds__(256) void mix_fir_kernel(const uint8_t *__restrict__ raw, int kind, int be, uint32_t stride,
                                                      uint64_t n, double freq_off, const double *__restrict__ h,
                                                      uint32_t K, uint32_t c, uint32_t down, uint32_t outs_wg,
                                                      double *__restrict__ ore, double *__restrict__ oim,
                                                      uint64_t n_out) {
    __shared__ double lre[FIR_SPAN], lim[FIR_SPAN];
    constexpr int PER_STEP = 256 / G;
    const int g = threadIdx.x % G, sub = threadIdx.x / G;
    for (uint64_t m0 = (uint64_t)blockIdx.x * outs_wg; m0 < n_out; m0 += (uint64_t)gridDim.x * outs_wg) {
        const uint32_t outs = n_out - m0 < outs_wg ? (uint32_t)(n_out - m0) : outs_wg;
        const uint32_t span = (outs - 1) * down + K;
        const int64_t first = (int64_t)(m0 * down) + (int64_t)c - (int64_t)(K - 1);  // burst index of lre[0]
        __syncthreads();  // the previous tile has been read
        for (uint32_t i = threadIdx.x; i < span; i += 256) {
            const int64_t idx = first + i;
            cx<double> z{0.0, 0.0};
            if (idx >= 0 && (uint64_t)idx < n) z = read_mixed(raw, kind, be, stride, (uint64_t)idx, freq_off);
            lre[i] = z.x;
            lim[i] = z.y;
        }
        __syncthreads();
        for (uint32_t o0 = 0; o0 < outs; o0 += PER_STEP) {
            const uint32_t o = o0 + sub;
            double ar = 0.0, ai = 0.0;
            if (o < outs) {
                const uint32_t top = o * down + (K - 1);  // span index read by tap 0
                for (uint32_t k = g; k < K; k += G) {
                    const double hk = h[k];
                    ar += hk * lre[top - k];
                    ai += hk * lim[top - k];
                }
            }
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) {
                ar += __shfl_xor(ar, off, 64);
                ai += __shfl_xor(ai, off, 64);
            }
            if (g == 0 && o < outs) {
                ore[m0 + o] = ar;
                oim[m0 + o] = ai;
            }
        }
    }
}

// ---- traces -------------------------------------------------------------------------------------
struct Aff { double a, b; };  // v -> a v + b
__device__ __forceinline__ Aff then(Aff first, Aff second) { return {first.a * second.a, second.a * first.b + second.b}; }

// A wave owns 512 consecutive elements of its workgroup's tile.  Global memory is touched in STRIPED order
// (instruction k, lane l -> element 64 k + l: consecutive lanes, consecutive addresses), the scan wants BLOCKED
// order (lane l -> elements 8 l .. 8 l + 7); the two are exchanged through a wave-private, padded LDS strip.
// One thread reading its eight consecutive doubles straight from memory made every load instruction touch 64
// different cache lines (the first version: 2.3x slower).
constexpr int TR_WAVE = 64 * TR_ITEMS;  // elements per wave
__device__ __forceinline__ int tr_pad(int i) { return i + (i >> 3); }
__device__ __forceinline__ void tr_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void striped_to_blocked(double (&v)[TR_ITEMS], double *strip, int lane) {
    tr_wave_sync();  // earlier readers of the strip are done
#pragma unroll
    for (int k = 0; k < TR_ITEMS; ++k) strip[tr_pad(k * 64 + lane)] = v[k];
    tr_wave_sync();
#pragma unroll
    for (int k = 0; k < TR_ITEMS; ++k) v[k] = strip[tr_pad(lane * TR_ITEMS + k)];
}
__device__ __forceinline__ void blocked_to_striped(double (&v)[TR_ITEMS], double *strip, int lane) {
    tr_wave_sync();
#pragma unroll
    for (int k = 0; k < TR_ITEMS; ++k) strip[tr_pad(lane * TR_ITEMS + k)] = v[k];
    tr_wave_sync();
#pragma unroll
    for (int k = 0; k < TR_ITEMS; ++k) v[k] = strip[tr_pad(k * 64 + lane)];
}

// inputs of the trace's sequence for this wave, STRIPED: x[k] <-> element e0 + 64 k + lane (zero past n_out)
template <int KIND_TRACE>
__device__ __forceinline__ void trace_inputs(const double *__restrict__ re, const double *__restrict__ im, uint64_t e0,
                                             int lane, uint64_t n_out, double fs, double (&x)[TR_ITEMS]) {
    if constexpr (KIND_TRACE == 0) {
#pragma unroll
        for (int k = 0; k < TR_ITEMS; ++k) {
            const uint64_t e = e0 + k * 64 + lane;
            x[k] = e < n_out ? hypot(re[e], im[e]) : 0.0;  // ADC:230
        }
    } else {  // ADC:265-276, output e is sample i = e + 1: one atan2 per sample, handed to the neighbouring output
        double ph[TR_ITEMS];
#pragma unroll
        for (int k = 0; k < TR_ITEMS; ++k) {
            const uint64_t e = e0 + k * 64 + lane;
            ph[k] = e <= n_out ? atan2(im[e], re[e]) : 0.0;  // samples 0 .. n_out exist
        }
        // phase of sample e + 1: the next lane's value, lane 63 takes lane 0 of the next instruction, and the
        // wave's very last output the first sample of the next wave
        const uint64_t e_last = e0 + TR_WAVE;
        double ph_next_wave = 0.0;
        if (lane == 63 && e_last <= n_out) ph_next_wave = atan2(im[e_last], re[e_last]);
#pragma unroll
        for (int k = 0; k < TR_ITEMS; ++k) {
            double nxt = __shfl_down(ph[k], 1, 64);
            const double wrap = k + 1 < TR_ITEMS ? __shfl(ph[k + 1 < TR_ITEMS ? k + 1 : k], 0, 64) : ph_next_wave;
            if (lane == 63) nxt = wrap;
            double d = nxt - ph[k];
            if (d > kPi) d -= 2 * kPi;
            else if (d < -kPi) d += 2 * kPi;
            const uint64_t e = e0 + k * 64 + lane;
            x[k] = e < n_out ? (d / (2 * kPi)) * fs : 0.0;
        }
    }
}

// inclusive scan of the 256 thread aggregates of a workgroup; returns this thread's EXCLUSIVE prefix and
// leaves the workgroup total in *total
__device__ __forceinline__ Aff wg_scan(Aff mine, Aff *lds, Aff *total) {
    const int t = threadIdx.x;
    lds[t] = mine;
    __syncthreads();
    for (int off = 1; off < TR_THREADS; off <<= 1) {
        Aff prev{1.0, 0.0};
        const bool has = t >= off;
        if (has) prev = lds[t - off];
        __syncthreads();
        if (has) lds[t] = then(prev, lds[t]);
        __syncthreads();
    }
    *total = lds[TR_THREADS - 1];
    const Aff excl = t == 0 ? Aff{1.0, 0.0} : lds[t - 1];
    __syncthreads();
    return excl;
}

// FINAL == false: per-tile aggregate map -> tile_aff[tile].
// FINAL == true: carry[tile] (value at the end of the previous tile) in, trace values out.
template <int KIND_TRACE, bool FINAL>
__global__ __launch_bounds__(TR_THREADS) void trace_kernel(const double *__restrict__ re, const double *__restrict__ im,
                                                           uint64_t n_out, double alpha, double fs, double add,
                                                           Aff *__restrict__ tile_aff, const double *__restrict__ carry,
                                                           double *out) {
    __shared__ Aff lds[TR_THREADS];
    __shared__ double strips[TR_THREADS / 64][TR_WAVE + TR_WAVE / 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *strip = strips[wave];
    const uint64_t e0 = (uint64_t)blockIdx.x * TR_TILE + (uint64_t)wave * TR_WAVE;  // first element of this wave
    const uint64_t j0 = e0 + (uint64_t)lane * TR_ITEMS;                              // first element of this thread (blocked)
    double x[TR_ITEMS];
    if constexpr (!FINAL) {  // first pass: compute the inputs and park them in `out` for the last pass
        trace_inputs<KIND_TRACE>(re, im, e0, lane, n_out, fs, x);
#pragma unroll
        for (int k = 0; k < TR_ITEMS; ++k) {
            const uint64_t e = e0 + k * 64 + lane;
            if (e < n_out) out[e] = x[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < TR_ITEMS; ++k) {
            const uint64_t e = e0 + k * 64 + lane;
            x[k] = e < n_out ? out[e] : 0.0;
        }
    }
    striped_to_blocked(x, strip, lane);
    Aff agg{1.0, 0.0};
#pragma unroll
    for (int k = 0; k < TR_ITEMS; ++k) {
        const uint64_t j = j0 + k;
        if (j < n_out) agg = then(agg, j == 0 ? Aff{0.0, x[k]} : Aff{1 - alpha, alpha * x[k]});  // ADC:233-237 / 277-281
    }
    Aff total;
    const Aff excl = wg_scan(agg, lds, &total);
    if constexpr (!FINAL) {
        if (threadIdx.x == 0) tile_aff[blockIdx.x] = total;
    } else {
        double v = excl.a * carry[blockIdx.x] + excl.b;  // value just before this thread's first element
#pragma unroll
        for (int k = 0; k < TR_ITEMS; ++k) {
            const uint64_t j = j0 + k;
            if (j < n_out) {
                v = j == 0 ? x[k] : alpha * x[k] + (1 - alpha) * v;
                x[k] = KIND_TRACE == 0 ? 20 * log10(v) : v + add;  // ADC:239 / ADC:282
            }
        }
        blocked_to_striped(x, strip, lane);
#pragma unroll
        for (int k = 0; k < TR_ITEMS; ++k) {
            const uint64_t e = e0 + k * 64 + lane;
            if (e < n_out) out[e] = x[k];
        }
    }
}

// one workgroup: carry[tile] = value at the end of tile - 1 (0 for tile 0, never used: element 0 resets)
__global__ __launch_bounds__(TR_THREADS) void trace_carry_kernel(const Aff *__restrict__ tile_aff, uint32_t n_tiles,
                                                                 double *__restrict__ carry) {
    __shared__ Aff lds[TR_THREADS];
    const uint32_t per = (n_tiles + TR_THREADS - 1) / TR_THREADS;
    const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    Aff agg{1.0, 0.0};
    for (uint32_t i = t0; i < t1; ++i) agg = then(agg, tile_aff[i]);
    Aff total;
    Aff run = wg_scan(agg, lds, &total);
    // every prefix that contains element 0 has a == 0, so its b is the value itself
    for (uint32_t i = t0; i < t1; ++i) {
        carry[i] = run.b;
        run = then(run, tile_aff[i]);
    }
}

}  // namespace

hipError_t launch_extract_mix(const uint8_t *raw, int kind, int be, uint32_t stride, uint64_t count, double freq_off,
                              double *re, double *im, hipStream_t s) {
    if (count == 0) return hipSuccess;
    const uint64_t wgs = (count + 255) / 256;
    hipLaunchKernelGGL(extract_mix_kernel, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(256), 0, s, raw, kind, be,
                       stride, count, freq_off, re, im);
    return hipGetLastError();
}

hipError_t launch_boxcar_decim(const uint8_t *raw, int kind, int be, uint32_t stride, double freq_off, uint32_t down,
                               double *ore, double *oim, uint64_t n_out, hipStream_t s) {
    if (n_out == 0) return hipSuccess;
    const uint64_t wgs = (n_out + 255) / 256;
    hipLaunchKernelGGL(boxcar_decim_kernel, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(256), 0, s, raw, kind, be,
                       stride, freq_off, down, ore, oim, n_out);
    return hipGetLastError();
}

hipError_t launch_fir_decim(const double *mr, const double *mi, uint64_t n, const double *h, uint32_t K, uint32_t c,
                            uint32_t down, double *ore, double *oim, uint64_t n_out, hipStream_t s) {
    if (n_out == 0) return hipSuccess;
    // lanes per output: about one lane per eight taps, a power of two between 1 and 64
    int G = 1;
    while (G < 64 && (uint32_t)(G * 16) <= K) G *= 2;
    const uint64_t per_wg = 256 / G, wgs64 = (n_out + per_wg - 1) / per_wg;
    const dim3 grid((unsigned)(wgs64 < 32768 ? wgs64 : 32768)), block(256);
#define SPEC_FIR_CASE(GG) case GG: hipLaunchKernelGGL(fir_decim_kernel<GG>, grid, block, 0, s, mr, mi, n, h, K, c, down, ore, oim, n_out); break
    switch (G) {
        SPEC_FIR_CASE(1); SPEC_FIR_CASE(2); SPEC_FIR_CASE(4); SPEC_FIR_CASE(8);
        SPEC_FIR_CASE(16); SPEC_FIR_CASE(32); SPEC_FIR_CASE(64);
    }
#undef SPEC_FIR_CASE
    return hipGetLastError();
}

// one-pass conventional down-converter; false when the filter does not fit the LDS span (very large `down`)
bool mix_fir_applicable(uint32_t K, uint32_t down) { return (uint64_t)K + down <= FIR_SPAN; }

hipError_t launch_mix_fir(const uint8_t *raw, int kind, int be, uint32_t stride, uint64_t n, double freq_off,
                          const double *h, uint32_t K, uint32_t c, uint32_t down, double *ore, double *oim, uint64_t n_out,
                          hipStream_t s) {
    if (n_out == 0) return hipSuccess;
    int G = 1;
    while (G < 64 && (uint32_t)(G * 16) <= K) G *= 2;
    uint32_t outs_wg = (FIR_SPAN - K) / down + 1;  // (outs - 1) down + K <= FIR_SPAN
    const uint32_t per_step = 256 / G;
    if (outs_wg > per_step) outs_wg -= outs_wg % per_step;  // whole steps
    const uint64_t wgs64 = (n_out + outs_wg - 1) / outs_wg;
    const dim3 grid((unsigned)(wgs64 < 32768 ? wgs64 : 32768)), block(256);
#define SPEC_MF_CASE(GG) case GG: hipLaunchKernelGGL(mix_fir_kernel<GG>, grid, block, 0, s, raw, kind, be, stride, n, freq_off, h, K, c, down, outs_wg, ore, oim, n_out); break
    switch (G) {
        SPEC_MF_CASE(1); SPEC_MF_CASE(2); SPEC_MF_CASE(4); SPEC_MF_CASE(8);
        SPEC_MF_CASE(16); SPEC_MF_CASE(32); SPEC_MF_CASE(64);
    }
#undef SPEC_MF_CASE
    return hipGetLastError();
}

size_t trace_scratch_bytes(uint64_t n_out) {
    const uint64_t tiles = (n_out + TR_TILE - 1) / TR_TILE;
    return (size_t)tiles * (sizeof(Aff) + sizeof(double));
}

hipError_t launch_trace(int kind_trace, const double *re, const double *im, uint64_t n_out, double alpha, double fs,
                        double add, void *scratch, double *out, hipStream_t s) {
    if (n_out == 0) return hipSuccess;
    const uint64_t tiles = (n_out + TR_TILE - 1) / TR_TILE;
    if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    Aff *tile_aff = static_cast<Aff *>(scratch);
    double *carry = reinterpret_cast<double *>(tile_aff + tiles);
    if (kind_trace == 0) {
        hipLaunchKernelGGL((trace_kernel<0, false>), dim3((unsigned)tiles), dim3(TR_THREADS), 0, s, re, im, n_out, alpha, fs,
                           add, tile_aff, nullptr, out);
        hipLaunchKernelGGL(trace_carry_kernel, dim3(1), dim3(TR_THREADS), 0, s, tile_aff, (uint32_t)tiles, carry);
        hipLaunchKernelGGL((trace_kernel<0, true>), dim3((unsigned)tiles), dim3(TR_THREADS), 0, s, re, im, n_out, alpha, fs,
                           add, tile_aff, carry, out);
    } else {
        hipLaunchKernelGGL((trace_kernel<1, false>), dim3((unsigned)tiles), dim3(TR_THREADS), 0, s, re, im, n_out, alpha, fs,
                           add, tile_aff, nullptr, out);
        hipLaunchKernelGGL(trace_carry_kernel, dim3(1), dim3(TR_THREADS), 0, s, tile_aff, (uint32_t)tiles, carry);
        hipLaunchKernelGGL((trace_kernel<1, true>), dim3((unsigned)tiles), dim3(TR_THREADS), 0, s, re, im, n_out, alpha, fs,
                           add, tile_aff, carry, out);
    }
    return hipGetLastError();
}

}  // namespace specgpu
