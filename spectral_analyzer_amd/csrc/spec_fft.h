// spec_fft.h -- device-side building blocks of the LDS Stockham FFT (gfx950).
//
// One time slice ("line") of nfft IQ samples is transformed by T = nfft / E
// threads, each holding E complex points in registers:  v[m] <-> index t + m*T.
// A pass of radix R combines registers {s + r*S} (S = E / R butterflies per
// thread), i.e. elements spaced nfft / R apart -- the Stockham autosort form,
// where every pass READS at stride T (coalesced from HBM in pass 0, conflict
// free from LDS afterwards) and WRITES butterfly i = t + s*T to
//     j + r*P,  j = (i - k)*R + k,  k = i mod P,   P = product of earlier radices
// so that the last pass leaves natural order in the registers: v[m] = X[t + m*T].
// The first pass comes straight from global memory and the last goes straight
// to the epilogue (|X|^2, log10, fftshift folded into the store index), so an
// nfft = 16^3 line makes two LDS round trips and touches HBM once each way.
//
// Reference behaviour implemented: SpectralService.java:40-82 (decode table,
// unnormalised forward DFT, 20 log10(|X| + 1e-10), fftshift).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace specgpu {

template <typename R> struct cx { R x, y; };

template <typename R> __device__ __forceinline__ cx<R> cadd(cx<R> a, cx<R> b) { return {a.x + b.x, a.y + b.y}; }
template <typename R> __device__ __forceinline__ cx<R> csub(cx<R> a, cx<R> b) { return {a.x - b.x, a.y - b.y}; }
template <typename R> __device__ __forceinline__ cx<R> cmul(cx<R> a, cx<R> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// a * (-i)
template <typename R> __device__ __forceinline__ cx<R> mul_mi(cx<R> a) { return {a.y, -a.x}; }

// ---------------------------------------------------------------------------
// small DFTs on registers, forward sign (exp(-2 pi i nk/R)), natural order out
// ---------------------------------------------------------------------------
template <typename R> __device__ __forceinline__ void dft2(cx<R> &a, cx<R> &b) {
    cx<R> t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

template <typename R> __device__ __forceinline__ void dft4(cx<R> &x0, cx<R> &x1, cx<R> &x2, cx<R> &x3) {
    cx<R> t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = mul_mi(csub(x1, x3));
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = cadd(t1, t3);
    x3 = csub(t1, t3);
}

template <typename R> __device__ __forceinline__ void dft8(cx<R> *u) {
    constexpr R h = (R)0.70710678118654752440084436210485L;
    // n = 2*n1 + n2 (n2 in {0,1}), k = k1 + 4*k2
    dft4(u[0], u[2], u[4], u[6]);  // n2 = 0 -> A0[k1]
    dft4(u[1], u[3], u[5], u[7]);  // n2 = 1 -> A1[k1]
    // A1[k1] *= W8^k1
    u[3] = cx<R>{(u[3].x + u[3].y) * h, (u[3].y - u[3].x) * h};   // W8^1 = h(1 - i)
    u[5] = mul_mi(u[5]);                                          // W8^2 = -i
    u[7] = cx<R>{(u[7].y - u[7].x) * h, -(u[7].x + u[7].y) * h};  // W8^3 = -h(1 + i)
    // X[k1] = A0 + A1, X[k1 + 4] = A0 - A1
    cx<R> y[8];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        y[k1] = cadd(u[2 * k1], u[2 * k1 + 1]);
        y[k1 + 4] = csub(u[2 * k1], u[2 * k1 + 1]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = y[k];
}

template <typename R> __device__ __forceinline__ void dft16(cx<R> *u) {
    constexpr R h = (R)0.70710678118654752440084436210485L;
    constexpr R c1 = (R)0.92387953251128675612818318939679L;  // cos(pi/8)
    constexpr R s1 = (R)0.38268343236508977172845998403040L;  // sin(pi/8)
    // n = 4*n1 + n2, k = k1 + 4*k2 ; A[n2][k1] = DFT4 over n1
    dft4(u[0], u[4], u[8], u[12]);
    dft4(u[1], u[5], u[9], u[13]);
    dft4(u[2], u[6], u[10], u[14]);
    dft4(u[3], u[7], u[11], u[15]);
    // after dft4 the slot 4*k1 + n2 holds A[n2][k1]; multiply by W16^(n2*k1)
    const cx<R> w1{c1, -s1}, w2{h, -h}, w3{s1, -c1}, w6{-h, -h}, w9{-c1, s1};
    u[5] = cmul(u[5], w1);    // n2=1,k1=1
    u[6] = cmul(u[6], w2);    // n2=2,k1=1
    u[7] = cmul(u[7], w3);    // n2=3,k1=1
    u[9] = cmul(u[9], w2);    // n2=1,k1=2
    u[10] = mul_mi(u[10]);    // n2=2,k1=2 : W16^4 = -i
    u[11] = cmul(u[11], w6);  // n2=3,k1=2
    u[13] = cmul(u[13], w3);  // n2=1,k1=3
    u[14] = cmul(u[14], w6);  // n2=2,k1=3
    u[15] = cmul(u[15], w9);  // n2=3,k1=3
    // X[k1 + 4*k2] = DFT4 over n2 of A'[n2][k1]
    dft4(u[0], u[1], u[2], u[3]);
    dft4(u[4], u[5], u[6], u[7]);
    dft4(u[8], u[9], u[10], u[11]);
    dft4(u[12], u[13], u[14], u[15]);
    // slot 4*k1 + k2 holds X[k1 + 4*k2] -> transpose the 4x4 to natural order
    cx<R> y[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) y[k1 + 4 * k2] = u[4 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = y[k];
}

template <typename R, int RADIX> __device__ __forceinline__ void dft(cx<R> *u) {
    if constexpr (RADIX == 2) dft2(u[0], u[1]);
    else if constexpr (RADIX == 4) dft4(u[0], u[1], u[2], u[3]);
    else if constexpr (RADIX == 8) dft8(u);
    else dft16(u);
}

// ---------------------------------------------------------------------------
// compile-time plan per log2(nfft)
// ---------------------------------------------------------------------------
template <int LOG2N> struct Plan;
#define SPEC_PLAN(L, E_, NP, ...)                                \
    template <> struct Plan<L> {                                 \
        static constexpr int N = 1 << L, E = E_, T = N / E_;     \
        static constexpr int WG = T >= 256 ? T : 256;            \
        static constexpr int LPW = WG / T; /* lines per WG */    \
        static constexpr int NPASS = NP;                         \
        static constexpr int radix[4] = {__VA_ARGS__};           \
    };
SPEC_PLAN(1, 2, 1, 2, 1, 1, 1)
SPEC_PLAN(2, 4, 1, 4, 1, 1, 1)
SPEC_PLAN(3, 8, 1, 8, 1, 1, 1)
SPEC_PLAN(4, 16, 1, 16, 1, 1, 1)
SPEC_PLAN(5, 16, 2, 16, 2, 1, 1)
SPEC_PLAN(6, 8, 2, 8, 8, 1, 1)
SPEC_PLAN(7, 16, 2, 16, 8, 1, 1)
SPEC_PLAN(8, 16, 2, 16, 16, 1, 1)
SPEC_PLAN(9, 8, 3, 8, 8, 8, 1)
SPEC_PLAN(10, 16, 3, 16, 16, 4, 1)
SPEC_PLAN(11, 16, 3, 16, 16, 8, 1)
SPEC_PLAN(12, 16, 3, 16, 16, 16, 1)
SPEC_PLAN(13, 16, 4, 16, 16, 16, 2)
SPEC_PLAN(14, 16, 4, 16, 16, 16, 4)
#undef SPEC_PLAN

template <int LOG2N, int PASS> constexpr int plan_P() {  // product of radices before PASS
    int p = 1;
    for (int q = 0; q < PASS; ++q) p *= Plan<LOG2N>::radix[q];
    return p;
}

// LDS element swizzle: XOR the low four index bits with the next four.  For the
// radix-16 plans this makes both the strided butterfly writes (16 lanes, 16
// elements apart) and the stride-T reads of the next pass bank-conflict free
// for 8-byte accesses (see DESIGN.md "LDS layout").
__device__ __forceinline__ int lds_swz(int a) { return a ^ ((a >> 4) & 15); }

// One Stockham pass on the registers of one thread.
//   tw : table W_N^m = exp(-2 pi i m / N), m in [0, N)
template <typename R, int LOG2N, int PASS>
__device__ __forceinline__ void fft_pass_regs(cx<R> (&v)[Plan<LOG2N>::E], int t,
                                              const cx<R> *__restrict__ tw) {
    using PL = Plan<LOG2N>;
    constexpr int RADIX = PL::radix[PASS], S = PL::E / RADIX, P = plan_P<LOG2N, PASS>();
#pragma unroll
    for (int s = 0; s < S; ++s) {
        cx<R> u[RADIX];
#pragma unroll
        for (int r = 0; r < RADIX; ++r) u[r] = v[s + r * S];
        if constexpr (P > 1) {
            const int k = (t + s * PL::T) & (P - 1);
            constexpr int STEP = PL::N / (P * RADIX);
#pragma unroll
            for (int r = 1; r < RADIX; ++r) u[r] = cmul(u[r], tw[r * k * STEP]);
        }
        dft<R, RADIX>(u);
#pragma unroll
        for (int r = 0; r < RADIX; ++r) v[s + r * S] = u[r];
    }
}

// registers -> LDS in the autosort order of PASS (call after fft_pass_regs)
template <typename R, int LOG2N, int PASS>
__device__ __forceinline__ void fft_pass_store(const cx<R> (&v)[Plan<LOG2N>::E], int t, cx<R> *lds) {
    using PL = Plan<LOG2N>;
    constexpr int RADIX = PL::radix[PASS], S = PL::E / RADIX, P = plan_P<LOG2N, PASS>();
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = t + s * PL::T, k = i & (P - 1), j = (i - k) * RADIX + k;
#pragma unroll
        for (int r = 0; r < RADIX; ++r) lds[lds_swz(j + r * P)] = v[s + r * S];
    }
}

// LDS -> registers at stride T
template <typename R, int LOG2N>
__device__ __forceinline__ void fft_pass_load(cx<R> (&v)[Plan<LOG2N>::E], int t, const cx<R> *lds) {
    using PL = Plan<LOG2N>;
#pragma unroll
    for (int m = 0; m < PL::E; ++m) v[m] = lds[lds_swz(t + m * PL::T)];
}

// Whole transform of the line held in v (input v[m] = x[t + m*T], output
// v[m] = X[t + m*T]).  All threads of the workgroup must call it together.
template <typename R, int LOG2N, int PASS = 0>
__device__ __forceinline__ void fft_line(cx<R> (&v)[Plan<LOG2N>::E], int t, cx<R> *lds,
                                         const cx<R> *__restrict__ tw) {
    using PL = Plan<LOG2N>;
    fft_pass_regs<R, LOG2N, PASS>(v, t, tw);
    if constexpr (PASS + 1 < PL::NPASS) {
        fft_pass_store<R, LOG2N, PASS>(v, t, lds);
        __syncthreads();
        fft_pass_load<R, LOG2N>(v, t, lds);
        __syncthreads();
        fft_line<R, LOG2N, PASS + 1>(v, t, lds, tw);
    }
}

// Same transform with a change of thread roles at the first exchange: pass 0 is
// done by butterfly index t0 on line buffer lds0, everything after the first
// LDS round trip by (t1, lds1).  Used by the large-N kernels, whose global loads
// want one lane order (lines fastest) and whose stores want the other.
template <typename R, int LOG2N>
__device__ __forceinline__ void fft_line_remap(cx<R> (&v)[Plan<LOG2N>::E], int t0, cx<R> *lds0, int t1,
                                               cx<R> *lds1, const cx<R> *__restrict__ tw) {
    using PL = Plan<LOG2N>;
    static_assert(PL::NPASS >= 2, "needs an exchange to remap at");
    fft_pass_regs<R, LOG2N, 0>(v, t0, tw);
    fft_pass_store<R, LOG2N, 0>(v, t0, lds0);
    __syncthreads();
    fft_pass_load<R, LOG2N>(v, t1, lds1);
    __syncthreads();
    fft_line<R, LOG2N, 1>(v, t1, lds1, tw);
}

// ---------------------------------------------------------------------------
// sample decode (SpectralService.java:40-65, ExtractDownConvertService.java:79-81)
// ---------------------------------------------------------------------------
enum : int { K_ZERO = 0, K_CU8 = 1, K_CI8 = 2, K_CI16 = 3, K_CF32 = 4, K_CF64 = 5 };

template <typename R>
__device__ __forceinline__ cx<R> decode_sample(const uint8_t *__restrict__ p, int kind, bool be) {
    switch (kind) {
    case K_CF32: {  // SS:46-49
        uint2 u = *reinterpret_cast<const uint2 *>(p);
        if (be) { u.x = __builtin_bswap32(u.x); u.y = __builtin_bswap32(u.y); }
        return {(R)__uint_as_float(u.x), (R)__uint_as_float(u.y)};
    }
    case K_CI16: {  // SS:42-45
        uint32_t u = *reinterpret_cast<const uint32_t *>(p);
        uint16_t a = (uint16_t)(u & 0xFFFFu), b = (uint16_t)(u >> 16);
        if (be) { a = __builtin_bswap16(a); b = __builtin_bswap16(b); }
        return {(R)(int16_t)a * (R)(1.0 / 32768.0), (R)(int16_t)b * (R)(1.0 / 32768.0)};
    }
    case K_CU8: {  // SS:50-54
        uint16_t u = *reinterpret_cast<const uint16_t *>(p);
        return {((R)(u & 0xFF) - (R)127.5) * (R)(1.0 / 128), ((R)(u >> 8) - (R)127.5) * (R)(1.0 / 128)};
    }
    case K_CI8: {  // SS:55-59
        uint16_t u = *reinterpret_cast<const uint16_t *>(p);
        return {(R)(int8_t)(u & 0xFF) * (R)(1.0 / 128), (R)(int8_t)(u >> 8) * (R)(1.0 / 128)};
    }
    case K_CF64: {  // EDC:79-81
        ulonglong2 u = *reinterpret_cast<const ulonglong2 *>(p);
        if (be) { u.x = __builtin_bswap64(u.x); u.y = __builtin_bswap64(u.y); }
        return {(R)__longlong_as_double((long long)u.x), (R)__longlong_as_double((long long)u.y)};
    }
    default:  // SS:60-63
        return {(R)0, (R)0};
    }
}

// ---------------------------------------------------------------------------
// epilogue: 20 log10(|X| + 1e-10)  (SS:80-81)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float db20(cx<float> z) {
    const float p = z.x * z.x + z.y * z.y;
    // |X| + 1e-10 == |X| in fp32 once |X| > 1e-10 * 2^24; above that threshold
    // 10 log10(p) is the same value without the square root.
    constexpr float k10 = 3.01029995663981195f;   // 10 log10(2)
    if (p > 1e-4f && p < 1e37f) return k10 * __log2f(p);
    const float a = sqrtf(p > 1e37f ? 1.0f : p) ;
    if (p >= 1e37f) {  // |X|^2 would overflow: scale first
        const float s = 1.0f / 1.8446744e19f;   // 2^-64
        const float xs = z.x * s, ys = z.y * s;
        return k10 * (__log2f(xs * xs + ys * ys) + 128.0f);
    }
    return 2.0f * k10 * __log2f(a + 1e-10f);
}

// fp64 form of the same expression.  The library hypot() + log10() cost ~150 fp64 operations
// per bin -- more than the 65536-point FFT itself spends per bin -- so the common range gets
// sqrt(x^2 + y^2) and a log2 built from frexp + the atanh series (|error| < 3e-13 dB, two orders
// under the 1e-9 dB parity tolerance); tiny / huge magnitudes keep the library path.
__device__ __forceinline__ double db20(cx<double> z) {
    const double p = __builtin_fma(z.x, z.x, z.y * z.y);
    if (!(p > 1e-280 && p < 1e280)) return 20.0 * log10(hypot(z.x, z.y) + 1e-10);
    const double a = sqrt(p) + 1e-10;                       // |X| + 1e-10   (SS:80-81)
    int e;
    double m = frexp(a, &e);                                // a = m 2^e, m in [0.5, 1)
    if (m < 0.70710678118654752440) { m += m; e -= 1; }     // m in [1/sqrt2, sqrt2)
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;     // ln m = 2 atanh(s), |s| < 0.1716
    double q = 2.0 / 15.0;
    q = __builtin_fma(q, s2, 2.0 / 13.0);
    q = __builtin_fma(q, s2, 2.0 / 11.0);
    q = __builtin_fma(q, s2, 2.0 / 9.0);
    q = __builtin_fma(q, s2, 2.0 / 7.0);
    q = __builtin_fma(q, s2, 2.0 / 5.0);
    q = __builtin_fma(q, s2, 2.0 / 3.0);
    q = __builtin_fma(q, s2, 2.0);
    // 20 log10(a) = 20 log10(2) (e + ln(m) / ln 2)
    return 6.0205999132796239043 * ((double)e + (q * s) * 1.4426950408889634074);
}

}  // namespace specgpu
