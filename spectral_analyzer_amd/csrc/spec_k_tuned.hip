// spec_k_tuned.hip -- the 4096-point spectrogram kernel of the headline
// configuration (BASELINE configs[1]/[2]: cf32 / ci16, 50 % overlap).
//
// One workgroup = 256 threads = one line at a time, 16 points per thread,
// three radix-16 Stockham passes (spec_fft.h), walking a run of consecutive
// lines.  What it adds over the generic kernel:
//   * overlap reuse in registers: thread t owns samples t + 256*m of the line;
//     a hop of 256*SH samples shifts them by SH registers, so only the SH*256
//     new samples are fetched per line -- every input byte crosses HBM once;
//   * the next line's samples are requested before the current line's FFT and
//     stay in flight behind it (no barrier in this kernel drains vmcnt);
//   * twiddles of pass 3 live in registers for the whole run, those of pass 2
//     (16 distinct per radix digit) come from a 2 KiB LDS table;
//   * exchange 1 uses the XOR swizzle (stride-16 writes), exchange 2 needs
//     none, so only 16 address XORs per line are spent on bank conflicts;
//   * 40 KiB of LDS per workgroup pins residency at exactly 4 workgroups per
//     CU, which lets the host hand every resident workgroup an equal run.
#include "spec_fft.h"
#include "spec_fft_pk.h"
#include "spec_internal.h"

namespace specgpu {

namespace {

constexpr int N = 4096, T = 256, E = 16;

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Raw sample formats of the tuned path.  INV_SCALE is the factor the decode
// table divides by (SS:44-45); the FFT is linear, so for ci16 the division is
// not applied to the samples but folded into the epilogue (one constant there
// instead of 16 multiplies per thread and line).
template <int KIND> struct Raw;
template <> struct Raw<K_CF32> {
    using type = u32x2;
    static constexpr int BPS = 8;
    static constexpr float SCALE = 1.0f;
    template <int AUX> static __device__ __forceinline__ u32x2 load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    }
    static __device__ __forceinline__ v2f dec(u32x2 u) { return v2f{__uint_as_float(u.x), __uint_as_float(u.y)}; }
};
template <> struct Raw<K_CI16> {
    using type = uint32_t;
    static constexpr int BPS = 4;
    static constexpr float SCALE = 1.0f / 32768.0f;  // SS:44-45
    template <int AUX> static __device__ __forceinline__ uint32_t load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX);
    }
    static __device__ __forceinline__ v2f dec(uint32_t u) {
        return v2f{(float)(int16_t)(u & 0xFFFFu), (float)((int32_t)u >> 16)};
    }
};

// Epilogue for the 16 bins of a thread.  `v` is the spectrum of the UNSCALED
// samples; the true spectrum is SCALE * v.  DB: 20 log10(|X| + 1e-10) (SS:80-81);
// else |X|^2.  One range test per thread: while every |X|^2 of the thread is
// above 1e-4 (|X| > 1e-2), |X| + 1e-10 rounds to |X| in fp32 and the value is
// 10 log10(p) with no square root.
template <bool DB>
__device__ __forceinline__ void epilogue_x16(const v2f (&v)[E], float scale, float (&d)[E]) {
    float p[E];
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const v2f s = v[m] * v[m];
        p[m] = s.x + s.y;
    }
    const float s2 = scale * scale;
    if constexpr (!DB) {
#pragma unroll
        for (int m = 0; m < E; ++m) d[m] = p[m] * s2;
    } else {
        float lo = fminf(fminf(p[0], p[1]), p[2]), hi = fmaxf(fmaxf(p[0], p[1]), p[2]);
#pragma unroll
        for (int m = 3; m + 1 < E; m += 2) { lo = fminf(fminf(lo, p[m]), p[m + 1]); hi = fmaxf(fmaxf(hi, p[m]), p[m + 1]); }
        lo = fminf(lo, p[E - 1]);
        hi = fmaxf(hi, p[E - 1]);
        constexpr float k10 = 3.01029995663981195f;  // 10 log10(2)
        const float off = k10 * __log2f(s2);         // 20 log10(scale): 0 for cf32
        if (lo * s2 > 1e-4f && hi < 1e37f) {
#pragma unroll
            for (int m = 0; m < E; ++m) d[m] = __builtin_fmaf(k10, __log2f(p[m]), off);
        } else {  // rare: a bin near the -200 dB floor, or near overflow
#pragma unroll
            for (int m = 0; m < E; ++m) d[m] = db20(cx<float>{v[m].x * scale, v[m].y * scale});
        }
    }
}

template <int KIND, int SH, bool HAS_WIN, bool NT, int OCC>
__global__ __launch_bounds__(256, OCC) void spectro4096_kernel(const WfArgs a) {
    using RW = Raw<KIND>;
    using raw_t = typename RW::type;
    constexpr int BPS = RW::BPS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v2f *lds = reinterpret_cast<v2f *>(smem);
    v4f *tw2_lds = reinterpret_cast<v4f *>(lds + N);  // [r][k] : W_256^(r k) as (c, d, -d, d), 16 x 16
    const int t = threadIdx.x, k2 = t & 15;
    const v2f *__restrict__ tw = static_cast<const v2f *>(a.tw);

    // twiddle set-up (once per run)
    {
        const v2f w0 = tw[(t >> 4) * (t & 15) * 16];
        tw2_lds[t] = v4f{w0.x, w0.y, -w0.y, w0.y};
    }
    v2f tw3[E];
#pragma unroll
    for (int r = 1; r < E; ++r) tw3[r] = tw[r * t];
    float w[E];
    if constexpr (HAS_WIN) {
        const float *__restrict__ win = static_cast<const float *>(a.win);
#pragma unroll
        for (int m = 0; m < E; ++m) w[m] = win[t + m * T];
    }

    const uint64_t first = (uint64_t)blockIdx.x * a.lines_per_wg;
    uint64_t last = first + a.lines_per_wg;
    if (last > a.n_lines) last = a.n_lines;
    constexpr uint32_t LINE_BYTES = (uint32_t)SH * T * BPS;
    constexpr int AUX = NT ? 2 : 0;  // nt
    // Buffer descriptors over this workgroup's run (wave-uniform, SGPRs): every
    // access is base + per-thread voffset (constant) + scalar soffset, so no VALU
    // address arithmetic is spent, and reads past the run return 0 without
    // touching memory (the prefetch issued behind the run's last line).
    const uint32_t run = (uint32_t)(last - first);
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.iq) + first * (uint64_t)LINE_BYTES, 0,
        (run - 1) * LINE_BYTES + (uint32_t)N * BPS, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<float *>(a.out) + first * (uint64_t)N, 0, run * (uint32_t)N * 4u, 0x00020000);
    const int toff = t * BPS, tout = t * 4;

    raw_t raw[E];
#pragma unroll
    for (int m = 0; m < E; ++m) raw[m] = RW::template load<AUX>(src, toff, m * T * BPS);

    // LDS addresses (in elements)
    int wr1 = 16 * t + k2;                              // exchange 1 write: (16 t + k2) ^ r
    const int rd1 = (t & ~15) | ((t ^ (t >> 4)) & 15);  // exchange 1 read : swz(t) + 256 m
    const int wr2 = (t >> 4) * 256 + k2;                // exchange 2 write: + 16 r
    const int rd2 = t;                                  // exchange 2 read : + 256 m

    for (uint32_t line = 0; line < run; ++line) {
        v2f v[E];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            v[m] = RW::dec(raw[m]);
            if constexpr (HAS_WIN) v[m] *= v2f{w[m], w[m]};
        }
        // slide the window of samples and request the next line's new ones
        if constexpr (SH < E) {
#pragma unroll
            for (int m = 0; m < E - SH; ++m) raw[m] = raw[m + SH];
        }
        const int next_off = (int)((line + 1) * LINE_BYTES);
#pragma unroll
        for (int m = E - SH; m < E; ++m) raw[m] = RW::template load<AUX>(src, toff, next_off + m * T * BPS);

        // pass 1 (P = 1): no twiddles
        pk_dft16(v);
        __syncthreads();  // WAR: everyone has finished reading exchange 2 of the previous line
        asm volatile("" : "+v"(wr1));  // recompute the 16 XORed addresses per line: cheaper than 16 pinned VGPRs
#pragma unroll
        for (int r = 0; r < E; ++r) lds[wr1 ^ r] = v[r];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < E; ++m) v[m] = lds[rd1 + 256 * m];
        // pass 2 (P = 16): W_256^(r k), k = t & 15
#pragma unroll
        for (int r = 1; r < E; ++r) {
            const v4f q = tw2_lds[r * 16 + k2];
            v[r] = pk_cmul_pre(v[r], v2f{q.x, q.y}, v2f{q.z, q.w});
        }
        pk_dft16(v);
        __syncthreads();  // WAR on exchange 1
#pragma unroll
        for (int r = 0; r < E; ++r) lds[wr2 + 16 * r] = v[r];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < E; ++m) v[m] = lds[rd2 + 256 * m];
        // pass 3 (P = 256): W_4096^(r t)
#pragma unroll
        for (int r = 1; r < E; ++r) {
            // keep (c, d) only: make the value opaque so that hipcc does not hoist the
            // derived (-d, d) pairs out of the line loop (30 more VGPRs)
            asm volatile("" : "+v"(tw3[r]));
            v[r] = pk_cmul(v[r], tw3[r]);
        }
        pk_dft16(v);

        // epilogue: |X| -> dB, fftshift folded into the index (SS:76-82)
        float d[E];
        if (a.out_fmt == OUT_DB20_F32) epilogue_x16<true>(v, RW::SCALE, d);
        else epilogue_x16<false>(v, RW::SCALE, d);
        const int out_off = (int)(line * (uint32_t)N * 4u);
#pragma unroll
        for (int m = 0; m < E; ++m)  // (t + m T + N/2) mod N
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d[m]), dst, tout,
                                                  out_off + ((m + E / 2) & (E - 1)) * T * 4, AUX);
    }
}

// LDS request: 32 KiB line + 4 KiB twiddles, padded so that exactly OCC
// workgroups fit the 160 KiB of a CU (the host hands out equal runs).
constexpr size_t tuned_lds(int occ) { return occ >= 4 ? 40 * 1024 : occ == 3 ? 48 * 1024 : 80 * 1024; }

template <int KIND, int SH, bool HAS_WIN, bool NT, int OCC> hipError_t launch_1(const WfArgs &a, hipStream_t s) {
    const uint64_t n_wg = (a.n_lines + a.lines_per_wg - 1) / a.lines_per_wg;
    hipLaunchKernelGGL((spectro4096_kernel<KIND, SH, HAS_WIN, NT, OCC>), dim3((unsigned)n_wg), dim3(256),
                       tuned_lds(OCC), s, a);
    return hipGetLastError();
}

// variant: bit 0 = non-temporal loads/stores, bits 1-2 = occupancy choice
// (0 -> default for the datatype, 1 -> 4, 2 -> 3, 3 -> 2 workgroups per CU).
// Defaults are the largest residency the kernel reaches without spilling:
// cf32 needs 156 VGPRs (3 per CU), ci16 120 (4 per CU).
template <int KIND> constexpr int tuned_occ(int variant) {
    switch ((variant >> 1) & 3) {
    case 1: return 4;
    case 2: return 3;
    case 3: return 2;
    default: return KIND == K_CI16 ? 4 : 3;
    }
}

template <int KIND, int SH, bool HAS_WIN> hipError_t launch_v(const WfArgs &a, int variant, hipStream_t s) {
    if constexpr (SH == 8 && !HAS_WIN) {  // the headline configuration carries the experiment matrix
        const bool nt = variant & 1;
        switch (tuned_occ<KIND>(variant)) {
        case 4: return nt ? launch_1<KIND, SH, HAS_WIN, true, 4>(a, s) : launch_1<KIND, SH, HAS_WIN, false, 4>(a, s);
        case 2: return nt ? launch_1<KIND, SH, HAS_WIN, true, 2>(a, s) : launch_1<KIND, SH, HAS_WIN, false, 2>(a, s);
        default: return nt ? launch_1<KIND, SH, HAS_WIN, true, 3>(a, s) : launch_1<KIND, SH, HAS_WIN, false, 3>(a, s);
        }
    } else {
        return launch_1<KIND, SH, HAS_WIN, false, 3>(a, s);
    }
}

template <int KIND, int SH> hipError_t launch_w(const WfArgs &a, int variant, hipStream_t s) {
    return a.win ? launch_v<KIND, SH, true>(a, variant, s) : launch_v<KIND, SH, false>(a, variant, s);
}

template <int KIND> hipError_t launch_k(const WfArgs &a, int variant, hipStream_t s) {
    switch (a.hop) {
    case 1024: return launch_w<KIND, 4>(a, variant, s);
    case 2048: return launch_w<KIND, 8>(a, variant, s);
    case 4096: return launch_w<KIND, 16>(a, variant, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool tuned4096_applicable(const WfArgs &a, int log2n) {
    return log2n == 12 && !a.be && (a.kind == K_CF32 || a.kind == K_CI16) &&
           (a.hop == 1024 || a.hop == 2048 || a.hop == 4096) &&
           (a.out_fmt == OUT_DB20_F32 || a.out_fmt == OUT_POW_F32);
}

hipError_t launch_spectro4096(const WfArgs &a, int variant, hipStream_t s) {
    return a.kind == K_CF32 ? launch_k<K_CF32>(a, variant, s) : launch_k<K_CI16>(a, variant, s);
}

}  // namespace specgpu
