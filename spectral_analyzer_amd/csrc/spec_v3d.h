// spec_v3d.h -- the fp64 member of the packed kernel family: the strict-parity pipeline (cf64 recordings and
// the SPEC_OUT_*_F64 outputs, |dB error| <= 1e-9) with the structure of spec_v2.h -- small radix first, LDS
// twiddle tables for the middle pass, last-pass twiddles in registers, padded LDS exchanges, one run of
// consecutive lines per sub-line with the next line's samples in flight behind the current FFT, overlap
// reuse in registers, buffer addressing.  Arithmetic is scalar fp64 on 2-vectors (spec_fft_pk.h, V = v2d).
// 256 ... 8192 points (16 points per thread: 64 VGPRs of line state; 8192 = plan id 113 of spec_v2.h); longer
// fp64 lines stay on the generic and four-step kernels.
#pragma once
#include "spec_v2.h"

namespace specgpu {

namespace {

// Last-pass twiddles W_N^(r t), r = 1 .. 15, from two LDS tables instead of 60 registers per thread (round 4, as
// spec_k_v3h.hip): t = j + J h, W_N^(r t) = W_N^(r j) W_N^(r J h) -- one more complex product per point of the last pass,
// 30 table reads per line; the registers hold the next line's samples in flight for cf64 / cf32 input as well.
#ifndef SPEC_V3D_LDS_TWL
#define SPEC_V3D_LDS_TWL 1
#endif
#ifndef SPEC_V3D_WIN_COMPUTED
#define SPEC_V3D_WIN_COMPUTED 1
#endif
#ifndef SPEC_V3D_PREFETCH_ALL
#define SPEC_V3D_PREFETCH_ALL 1
#endif
template <int L> struct V3dTw {
    using PL = Plan2<L>;
    static constexpr int J = PL::T < 32 ? PL::T : 32, HB = PL::T / J;
    static constexpr int ENTRIES = 15 * J + (HB > 1 ? 15 * HB : 0);
    // 8192 points only: 0.36 -> 0.43 of 8 TB/s (cf64 -> f64; the registers' 60 hold the next line in flight).  Measured
    // slower at 4096 points (0.51 -> 0.45: the tables take the second workgroup's LDS) and at 1024 (0.58 -> 0.56).
    static constexpr bool USE = SPEC_V3D_LDS_TWL != 0 && L == 113;
    static constexpr size_t BYTES = USE ? (size_t)ENTRIES * sizeof(v2d) : 0;
};
template <int L> __device__ __forceinline__ void v3d_fill_twl(v2d *tl, const v2d *__restrict__ tw, int tid) {
    using W = V3dTw<L>;
    for (int e = tid; e < W::ENTRIES; e += Plan2<L>::WG) {
        if (e < 15 * W::J) tl[e] = tw[(e / W::J + 1) * (e % W::J)];
        else { const int f = e - 15 * W::J; tl[e] = tw[(f / W::HB + 1) * W::J * (f % W::HB)]; }
    }
}
template <int L, int PASS = 0>
__device__ __forceinline__ void v3d_fft(v2d (&v)[Plan2<L>::E], int t, v2d *lds, const v2d *tab, const v2d *tl) {
    using PL = Plan2<L>;
    using W = V3dTw<L>;
    if constexpr (PASS + 1 < PL::NPASS) {
        v2d none[16];  // (only the last pass reads its twiddle registers)
        v2_pass<L, PASS>(v, t, tab, none);
        v2_sync<L>();  // WAR: the previous exchange has been read by everyone
        v2_store<L, PASS>(v, t, lds);
        v2_sync<L>();
        v2_load<L, PASS>(v, t, lds);
        v3d_fft<L, PASS + 1>(v, t, lds, tab, tl);
    } else {
        static_assert(PL::radix[PASS] == 16 && PL::E == 16, "last pass: one radix-16 butterfly per thread");
        const v2d *ra = tl + (t & (W::J - 1));
        const v2d *rb = tl + 15 * W::J + (t / W::J);
        // five at a time: left alone the scheduler requests every table entry first (120 registers)
#pragma unroll
        for (int r0 = 1; r0 < 16; r0 += 5) {
#pragma unroll
            for (int r = r0; r < r0 + 5; ++r) {
                if constexpr (W::HB > 1) v[r] = pk_cmul(v[r], pk_cmul(ra[(r - 1) * W::J], rb[(r - 1) * W::HB]));
                else v[r] = pk_cmul(v[r], ra[(r - 1) * W::J]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pk_dft16(v);
    }
}

// MODE 0: spectrogram lines (MC:980-999 around SS:33-85), fp64 arithmetic; doubles or floats out.
// MODE 1: Welch partial sums in fp64 (the dialog's calculatePsdWelch call, ADC:308-312, and cf64 / big-endian /
// fp64-output PSDs): every sub-line adds |X|^2 of its run of segments in registers and leaves ONE double slab;
// welch_finalize_kernel sums the slabs in a fixed order.  No power line ever reaches HBM.
template <int L, int KIND, int SH, bool HAS_WIN, bool BE, int MODE = 0>
__global__ __launch_bounds__(Plan2<L>::WG, 2) void v3d_kernel(const V2Args a) {
    using PL = Plan2<L>;
    using RW = Raw2<KIND>;
    using raw_t = typename RW::type;
    static_assert(PL::E == 16, "fp64 family: 16 points per thread");
    constexpr int BPS = RW::BPS, N = PL::N, T = PL::T, E = PL::E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, t = tid % T, q = tid / T;
    v2d *lds = reinterpret_cast<v2d *>(smem) + (size_t)q * PL::LINE;
    v2d *tab = reinterpret_cast<v2d *>(smem + (size_t)PL::LPW * PL::LINE * 16);
    const v2d *__restrict__ tw = static_cast<const v2d *>(a.tw);

    if constexpr (PL::NPASS > 2) fill_tables<L, 1>(tab, tw, tid);
    // table of the dB epilogue's logarithm (spec_fft.h, db20_tab_n) behind the twiddle tables
    const double *dbt = reinterpret_cast<const double *>(smem + p2_lds_bytes<L, 16>());
    if constexpr (MODE == 0) {
        double *dbt_w = reinterpret_cast<double *>(smem + p2_lds_bytes<L, 16>());
        for (int e = tid; e < DB20_TAB_DOUBLES; e += PL::WG) dbt_w[e] = DB20_TAB[e];
    }
    (void)dbt;
    constexpr bool LDS_TWL = V3dTw<L>::USE;
    v2d *tl = reinterpret_cast<v2d *>(smem + p2_lds_bytes<L, 16>() + (MODE == 0 ? DB20_TAB_DOUBLES * sizeof(double) : 0));
    v2d twl[16];
    if constexpr (LDS_TWL) {
        v3d_fill_twl<L>(tl, tw, tid);
    } else {
#pragma unroll
        for (int r = 1; r < 16; ++r) twl[r] = tw[(r * t) & (N - 1)];
    }
    const double *win = static_cast<const double *>(a.win);
    // Round 4: the Hann window computed, not read (as spec_k_v3h.hip): w[n] = 1/2 - 1/2 Re(W_N^t W_16^m) for n = t + m T, from the
    // thread's W_N^t and sixteen W_16^m in LDS -- three fp64 operations per sample instead of a load from L2 whose latency every
    // line waited out (+3 ... +18 %, profiles/r04_fp64_window.txt).  Any other table (the ones of a rectangular Welch) is read.
    v2d *w16 = reinterpret_cast<v2d *>(smem + p2_lds_bytes<L, 16>() + (MODE == 0 ? DB20_TAB_DOUBLES * sizeof(double) : 0) + V3dTw<L>::BYTES);
    const bool hann = HAS_WIN && SPEC_V3D_WIN_COMPUTED != 0 && a.win_hann == 1;
    v2d wt = v2d{1.0, 0.0};
    if constexpr (HAS_WIN) {
        if (hann) {
            if (tid < 16) w16[tid] = tw[tid * T];
            wt = tw[t];
        }
    }
    if constexpr (PL::NPASS > 2 || MODE == 0 || LDS_TWL || HAS_WIN) __syncthreads();

    const uint32_t unit = blockIdx.x / a.wgs_per_unit, wg = blockIdx.x % a.wgs_per_unit;
    const uint32_t line0 = wg * (uint32_t)PL::LPW * a.run;
    uint32_t lines_wg = a.n_lines - line0;
    if (lines_wg > (uint32_t)PL::LPW * a.run) lines_wg = PL::LPW * a.run;
    const uint32_t line_bytes = a.hop * BPS;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.iq) + (uint64_t)unit * a.unit_stride + (uint64_t)line0 * line_bytes, 0,
        (lines_wg - 1) * line_bytes + (uint32_t)N * BPS, 0x00020000);
    const int voff = (int)(q * a.run * line_bytes) + t * BPS;
    constexpr int AUX = 2, ST_AUX = 2;
    constexpr int NEW = SH > 0 ? SH : E;
    // The next line's samples stay in flight behind the current FFT only for the 2- and 4-byte formats
    // (16 / 16 registers).  For cf32 / cf64 the 32 / 64 registers of a second line push the kernel over its
    // 256 and the spills cost more than the prefetch hides (measured: cf64 4096-pt 37 -> 46 M lines/s,
    // cf32 -> f64 51 -> 53 M lines/s without it); those load each line at its start.
    // (With the last pass's twiddles in LDS every format has the room.)
    // (Welch keeps the old rule: 16 more registers of sums, and its overlap is re-read from L2 anyway.)
    constexpr bool PREFETCH = (KIND != K_CF64 && KIND != K_CF32) || (LDS_TWL && SPEC_V3D_PREFETCH_ALL != 0 && MODE == 0);
    raw_t raw[E];
    if constexpr (PREFETCH) {
#pragma unroll
        for (int m = 0; m < E; ++m) raw[m] = RW::template load<AUX>(src, voff, m * T * BPS);
    }

    const bool out64 = a.out_fmt == OUT_DB20_F64 || a.out_fmt == OUT_POW_F64;
    const bool out_db = a.out_fmt == OUT_DB20_F64 || a.out_fmt == OUT_DB20_F32;
    const uint32_t esz = out64 ? 8u : 4u;
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<uint8_t *>(a.out) + (uint64_t)line0 * N * esz, 0, lines_wg * (uint32_t)N * esz, 0x00020000);
    const int ovoff = (int)(q * a.run * (uint32_t)N * esz) + t * (int)esz;
    const uint32_t my_first = q * a.run;
    const uint32_t my_lines = my_first >= lines_wg ? 0 : (lines_wg - my_first < a.run ? lines_wg - my_first : a.run);
    const uint32_t iters = PL::WAVE_LOCAL ? a.run : my_lines;
    constexpr double scale = (double)RW::SCALE;
    double acc[MODE == 1 ? E : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int m = 0; m < E; ++m) acc[m] = 0.0;
    }

    for (uint32_t line = 0; line < iters; ++line) {
        v2d v[E];
        if constexpr (!PREFETCH) {
            const int off = (int)(line * line_bytes);
#pragma unroll
            for (int m = 0; m < E; ++m) raw[m] = RW::template load<0>(src, voff, off + m * T * BPS);
        }
#pragma unroll
        for (int m = 0; m < E; ++m) v[m] = RW::template dec<v2d>(BE ? RW::swap(raw[m]) : raw[m]);  // SMH:87-91
        if constexpr (HAS_WIN) {
            if (hann) {
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const v2d cs = w16[m];
                    const double c = __builtin_fma(wt.x, cs.x, -(wt.y * cs.y));
                    const double w = __builtin_fma(-0.5, c, 0.5);
                    v[m] *= v2d{w, w};
                }
            } else {
                const double *wp = win;
                asm volatile("" : "+s"(wp));  // window values are re-read every line (no registers to keep them)
#pragma unroll
                for (int m = 0; m < E; ++m) { const double w = wp[t + m * T]; v[m] *= v2d{w, w}; }
            }
        }
        if constexpr (PREFETCH) {
            if constexpr (SH > 0 && SH < E) {
#pragma unroll
                for (int m = 0; m < E - SH; ++m) raw[m] = raw[m + SH];
            }
            const int next_off = (int)((line + 1) * line_bytes);
#pragma unroll
            for (int m = E - NEW; m < E; ++m) raw[m] = RW::template load<AUX>(src, voff, next_off + m * T * BPS);
        }

        if constexpr (LDS_TWL) v3d_fft<L>(v, t, lds, tab, tl);
        else v2_fft<L>(v, t, lds, tab, twl);

        if constexpr (MODE == 1) {
            if (line < my_lines) {
#pragma unroll
                for (int m = 0; m < E; ++m) acc[m] = __builtin_fma(v[m].x, v[m].x, __builtin_fma(v[m].y, v[m].y, acc[m]));  // two chained FMAs per point
            }
            continue;
        }
        // out[(k + N/2) mod N] = 20 log10(|X_k| + 1e-10)  (SS:78-81), or |X_k|^2
        // (the table logarithm of spec_fft.h, four bins at a time: the series form without a branch per bin)
        const int out_off = (int)(line * (uint32_t)N * esz);
        constexpr int G = 4;
#pragma unroll
        for (int g = 0; g < E; g += G) {
            double val[G];
            if (out_db) {
                cx<double> z[G];
#pragma unroll
                for (int j = 0; j < G; ++j) z[j] = cx<double>{v[g + j].x * scale, v[g + j].y * scale};
                db20_tab_n<G>(z, dbt, val);
            } else {
#pragma unroll
                for (int j = 0; j < G; ++j) val[j] = pk_norm(v[g + j]) * (scale * scale);
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int m = g + j;
                const int so = out_off + ((m + E / 2) & (E - 1)) * T * (int)esz;
                if (out64) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(val[j]);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)bits, (uint32_t)(bits >> 32)}, dst, ovoff, so, ST_AUX);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)val[j]), dst, ovoff, so, ST_AUX);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // one group's logarithms at a time (registers)
        }
    }
    if constexpr (MODE == 1) {  // one fp64 slab per sub-line (zeros for idle ones), unshifted bins
        double *slab = static_cast<double *>(a.out) + ((uint64_t)blockIdx.x * PL::LPW + q) * N;
#pragma unroll
        for (int m = 0; m < E; ++m) slab[t + m * T] = acc[m] * (scale * scale);
    }
}

template <int L, int KIND, int SH, bool HAS_WIN, bool BE = false, int MODE = 0>
hipError_t v3d_launch1(const V2Args &a, hipStream_t s) {
    using PL = Plan2<L>;
    constexpr size_t lds = p2_lds_bytes<L, 16>() + (MODE == 0 ? DB20_TAB_DOUBLES * sizeof(double) : 0) + V3dTw<L>::BYTES + (HAS_WIN ? 16 * sizeof(v2d) : 0);
    static_assert(lds <= 160 * 1024, "one workgroup's LDS");
    auto kern = v3d_kernel<L, KIND, SH, HAS_WIN, BE, MODE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.wgs_per_unit * (a.n_units ? a.n_units : 1)), dim3(PL::WG), lds, s, a);
    return hipGetLastError();
}

// Welch partial sums: always through the window table (rectangular = ones: one variant per format), the overlap
// re-read from L2 (any hop); big-endian files by the swapping variant
template <int L, int KIND> hipError_t v3d_launch_welch(const V2Args &a, hipStream_t s) {
    if constexpr (KIND == K_CF32 || KIND == K_CI16 || KIND == K_CF64) {
        if (a.be) return v3d_launch1<L, KIND, 0, true, true, 1>(a, s);
    }
    return v3d_launch1<L, KIND, 0, true, false, 1>(a, s);
}
template <int L> hipError_t v3d_launch_welch_kind(const V2Args &a, int kind, hipStream_t s) {
    switch (kind) {
    case K_CF64: return v3d_launch_welch<L, K_CF64>(a, s);
    case K_CF32: return v3d_launch_welch<L, K_CF32>(a, s);
    case K_CI16: return v3d_launch_welch<L, K_CI16>(a, s);
    case K_CU8: return v3d_launch_welch<L, K_CU8>(a, s);
    case K_CI8: return v3d_launch_welch<L, K_CI8>(a, s);
    default: return hipErrorInvalidValue;
    }
}

// register-reuse variant for 50 % overlap; every other hop and big-endian files re-read the overlap from L2
template <int L, int KIND> hipError_t v3d_launch_sh(const V2Args &a, hipStream_t s) {
    constexpr int N = Plan2<L>::N, E = Plan2<L>::E;
    const bool win = a.win != nullptr;
    if constexpr (KIND == K_CF32 || KIND == K_CI16 || KIND == K_CF64) {
        if (a.be) return win ? v3d_launch1<L, KIND, 0, true, true>(a, s) : v3d_launch1<L, KIND, 0, false, true>(a, s);
    }
    if (a.hop == N / 2) return win ? v3d_launch1<L, KIND, E / 2, true>(a, s) : v3d_launch1<L, KIND, E / 2, false>(a, s);
    return win ? v3d_launch1<L, KIND, 0, true>(a, s) : v3d_launch1<L, KIND, 0, false>(a, s);
}

template <int L> hipError_t v3d_launch_kind(const V2Args &a, int kind, hipStream_t s) {
    switch (kind) {
    case K_CF64: return v3d_launch_sh<L, K_CF64>(a, s);
    case K_CF32: return v3d_launch_sh<L, K_CF32>(a, s);
    case K_CI16: return v3d_launch_sh<L, K_CI16>(a, s);
    case K_CU8: return v3d_launch_sh<L, K_CU8>(a, s);
    case K_CI8: return v3d_launch_sh<L, K_CI8>(a, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

}  // namespace specgpu
