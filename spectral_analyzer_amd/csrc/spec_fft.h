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

// Table form of the same expression for the kernels whose epilogue sets the pace (spec_k_team.hip): no square
// root, no division.  For |X| > 1e-2:  20 log10(|X| + 1e-10) = 10 log10(p) + (20 / ln 10) 1e-10 / |X|  (the
// next term of the series is < 5e-16 dB), p = |X|^2 = m 2^e with m in [1, 2) cut into 32 intervals: ln m =
// ln(m inv_i) - ln(inv_i), inv_i = fp64(1 / centre of interval i), -ln(inv_i) tabulated for that ROUNDED inv_i
// (an identity, no approximation), |m inv_i - 1| < 1/64 so that ln(1 + r) needs the terms up to r^8; 1 / |X| from
// the hardware reciprocal-square-root estimate (1e-7 of a term that is < 1e-7 dB).  Checked on 2e5 random magnitudes in
// 1e-2 ... 1e18 against 60-digit arithmetic: |error| <= 1.2e-13 dB.  Weaker magnitudes take the expression as
// written (square root, + 1e-10, the same table logarithm), huge ones are rescaled first.
// DB20_TAB: {inv_i, -ln(inv_i)} pairs; the caller copies them to LDS (`tab`) once per workgroup.
__device__ const double DB20_TAB[64] = {
    0x1.f81f81f81f820p-1, 0x1.fc0a8b0fc03c4p-7, 0x1.e9131abf0b767p-1, 0x1.77458f632dcffp-5,
    0x1.dae6076b981dbp-1, 0x1.341d7961bd1d0p-4, 0x1.cd85689039b0bp-1, 0x1.a926d3a4ad562p-4,
    0x1.c0e070381c0e0p-1, 0x1.0d77e7cd08e5bp-3, 0x1.b4e81b4e81b4fp-1, 0x1.44d2b6ccb7d1cp-3,
    0x1.a98ef606a63bep-1, 0x1.7ab890210d907p-3, 0x1.9ec8e951033d9p-1, 0x1.af3c94e80bff3p-3,
    0x1.948b0fcd6e9e0p-1, 0x1.e27076e2af2e8p-3, 0x1.8acb90f6bf3aap-1, 0x1.0a324e27390e2p-2,
    0x1.8181818181818p-1, 0x1.22941fbcf7966p-2, 0x1.78a4c8178a4c8p-1, 0x1.3a64c556945eap-2,
    0x1.702e05c0b8170p-1, 0x1.51aad872df82ep-2, 0x1.6816816816817p-1, 0x1.686c81e9b14adp-2,
    0x1.6058160581606p-1, 0x1.7eaf83b82afc2p-2, 0x1.58ed2308158edp-1, 0x1.947941c2116fbp-2,
    0x1.51d07eae2f815p-1, 0x1.a9cec9a9a084ap-2, 0x1.4afd6a052bf5bp-1, 0x1.beb4d9da71b7ap-2,
    0x1.446f86562d9fbp-1, 0x1.d32fe7e00ebd5p-2, 0x1.3e22cbce4a902p-1, 0x1.e744261d68789p-2,
    0x1.3813813813814p-1, 0x1.faf588f78f31dp-2, 0x1.323e34a2b10bfp-1, 0x1.0723e5c1cdf41p-1,
    0x1.2c9fb4d812ca0p-1, 0x1.109f39e2d4c96p-1, 0x1.27350b8812735p-1, 0x1.19ee6b467c96fp-1,
    0x1.21fb78121fb78p-1, 0x1.23130d7bebf43p-1, 0x1.1cf06ada2811dp-1, 0x1.2c0e9ed448e8cp-1,
    0x1.1811811811812p-1, 0x1.34e289d9ce1d2p-1, 0x1.135c81135c811p-1, 0x1.3d9026a7156fbp-1,
    0x1.0ecf56be69c90p-1, 0x1.4618bc21c5ec2p-1, 0x1.0a6810a6810a7p-1, 0x1.4e7d811b75bb0p-1,
    0x1.0624dd2f1a9fcp-1, 0x1.56bf9d5b3f399p-1, 0x1.0204081020408p-1, 0x1.5ee02a9241676p-1,
};
// ln of a positive normal number by the table (see above)
__device__ __forceinline__ double ln_tab(double v, const double *tab) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const int e = (int)(bits >> 52) - 1023;
    const double m = __longlong_as_double((long long)((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    const int i = (int)(bits >> 47) & 31;
    const double inv = tab[2 * i], li = tab[2 * i + 1];
    const double r = __builtin_fma(m, inv, -1.0);
    double q = -1.0 / 8;
    q = __builtin_fma(q, r, 1.0 / 7);
    q = __builtin_fma(q, r, -1.0 / 6);
    q = __builtin_fma(q, r, 1.0 / 5);
    q = __builtin_fma(q, r, -1.0 / 4);
    q = __builtin_fma(q, r, 1.0 / 3);
    q = __builtin_fma(q, r, -0.5);
    return __builtin_fma((double)e, 0x1.62e42fefa39efp-1, li + __builtin_fma(r * r, q, r));
}
// 20 log10(|X| + 1e-10).  Everything inline and short: the library fall-backs of db20() are ~250 instructions per
// bin, eight times per line they made the row side's loop body larger than the instruction cache it shares.
__device__ __forceinline__ double db20_tab(cx<double> z, const double *tab) {
    constexpr double K10 = 0x1.15f2ced384f29p+2;  // 10 / ln 10
    const double p = __builtin_fma(z.x, z.x, z.y * z.y);
    // one logarithm, three ways to its argument: result = mul * ln(arg) + add
    double arg = p, mul = K10, add;
    if (p > 1e-4 && p < 1e300) {  // |X| > 1e-2: the series form.  1 / |X| from the hardware estimate (v_rsq_f64, one
        // quarter-rate instruction; two conversions around v_rsq_f32 cost three): ~1e-8 of a term below 1e-7 dB
        add = 0x1.dd8307784b277p-31 * __builtin_amdgcn_rsq(p);
    } else if (p <= 1e-4) {  // weak bins, zero and underflow: the expression as written, |X| + 1e-10 in [1e-10, 1e-2]
        arg = sqrt(p) + 1e-10;
        mul = 2.0 * K10;
        add = 0.0;
        if (arg == 1e-10) return -200.0;  // silence is exactly 20 log10(1e-10)
    } else {  // |X|^2 beyond 1e300, infinite or NaN: rescale by 2^-600 (|X| + 1e-10 == |X| here)
        const double xs = z.x * 0x1p-600, ys = z.y * 0x1p-600;
        arg = __builtin_fma(xs, xs, ys * ys);
        add = 600.0 * 0x1.8151824c7587fp+2;  // 20 log10(2^600)
        if (!(arg < 1e300)) return arg;  // +inf stays +inf, NaN stays NaN (Math.log10 does the same)
    }
    return __builtin_fma(ln_tab(arg, tab), mul, add);
}

}  // namespace specgpu
