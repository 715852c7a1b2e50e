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
// root, no division, no integer-to-double conversion.  For |X|^2 = p in [2^-13, 2^996):
//   20 log10(|X| + 1e-10) = 10 log10(p) + (20 / ln 10) 1e-10 / |X|     (the next term of the series is < 4e-16 dB),
// p = m 2^e with m in [1, 2) cut into 128 intervals: ln m = ln(m inv_i) - ln(inv_i), inv_i = fp64(1 / centre of
// interval i), -ln(inv_i) tabulated for that ROUNDED inv_i (an identity, no approximation), |m inv_i - 1| <= 2^-8
// so that ln(1 + r) needs the terms up to r^5 (r^6 / 6 < 6e-16); 1 / |X| from the hardware reciprocal-square-root
// estimate (1e-7 of a term that is < 1e-7 dB).  tools/gen_db20_table.py generates the table and evaluates the
// expression operation by operation against 60-digit arithmetic: |error| <= 6e-14 dB over 1.1e-2 ... 1e18.
// Weaker magnitudes take the expression as written (square root, + 1e-10, the same table logarithm), huge ones
// are rescaled first.
// DB20_TAB: {inv_i, -ln(inv_i)} pairs; the caller copies them to LDS (`tab`, 16-byte aligned) once per workgroup.
constexpr int DB20_TAB_DOUBLES = 256;
__device__ const double DB20_TAB[DB20_TAB_DOUBLES] = {
    0x1.fe01fe01fe020p-1, 0x1.ff00aa2b10ba0p-9, 0x1.fa11caa01fa12p-1, 0x1.7dc475f810a69p-7,
    0x1.f6310aca0dbb5p-1, 0x1.3cea44346a584p-6, 0x1.f25f644230ab5p-1, 0x1.b9fc027af919ap-6,
    0x1.ee9c7f8458e02p-1, 0x1.1b0d98923d97fp-5, 0x1.eae807aba01ebp-1, 0x1.58a5bafc8e4d3p-5,
    0x1.e741aa59750e4p-1, 0x1.95c830ec8e3f2p-5, 0x1.e3a9179dc1a73p-1, 0x1.d276b8adb0b56p-5,
    0x1.e01e01e01e01ep-1, 0x1.075983598e471p-4, 0x1.dca01dca01dcap-1, 0x1.253f62f0a1417p-4,
    0x1.d92f2231e7f8ap-1, 0x1.42edcbea646eep-4, 0x1.d5cac807572b2p-1, 0x1.60658a93750c4p-4,
    0x1.d272ca3fc5b1ap-1, 0x1.7da766d7b12d0p-4, 0x1.cf26e5c44bfc6p-1, 0x1.9ab42462033aep-4,
    0x1.cbe6d9601cbe7p-1, 0x1.b78c82bb0eda0p-4, 0x1.c8b265afb8a42p-1, 0x1.d4313d66cb35dp-4,
    0x1.c5894d10d4986p-1, 0x1.f0a30c01162a4p-4, 0x1.c26b5392ea01cp-1, 0x1.0671512ca596fp-3,
    0x1.bf583ee868d8bp-1, 0x1.14785846742acp-3, 0x1.bc4fd65883e7bp-1, 0x1.2266f190a5acdp-3,
    0x1.b951e2b18ff23p-1, 0x1.303d718e47fd5p-3, 0x1.b65e2e3beee05p-1, 0x1.3dfc2b0ecc62ap-3,
    0x1.b37484ad806cep-1, 0x1.4ba36f39a55e5p-3, 0x1.b094b31d922a4p-1, 0x1.59338d9982085p-3,
    0x1.adbe87f94905ep-1, 0x1.66acd4272ad51p-3, 0x1.aaf1d2f87ebfdp-1, 0x1.740f8f54037a3p-3,
    0x1.a82e65130e159p-1, 0x1.815c0a14357e9p-3, 0x1.a574107688a4ap-1, 0x1.8e928de886d41p-3,
    0x1.a2c2a87c51ca0p-1, 0x1.9bb362e7dfb85p-3, 0x1.a01a01a01a01ap-1, 0x1.a8becfc882f19p-3,
    0x1.9d79f176b682dp-1, 0x1.b5b519e8fb5a6p-3, 0x1.9ae24ea5510dap-1, 0x1.c2968558c18c2p-3,
    0x1.9852f0d8ec0ffp-1, 0x1.cf6354e09c5ddp-3, 0x1.95cbb0be377aep-1, 0x1.dc1bca0abec7bp-3,
    0x1.934c67f9b2ce6p-1, 0x1.e8c0252aa5a60p-3, 0x1.90d4f120190d5p-1, 0x1.f550a564b7b37p-3,
    0x1.8e6527af1373fp-1, 0x1.00e6c45ad501dp-2, 0x1.8bfce8062ff3ap-1, 0x1.071b85fcd590dp-2,
    0x1.899c0f601899cp-1, 0x1.0d46b579ab74bp-2, 0x1.87427bcc092b9p-1, 0x1.136870293a8b0p-2,
    0x1.84f00c2780614p-1, 0x1.1980d2dd4236fp-2, 0x1.82a4a0182a4a0p-1, 0x1.1f8ff9e48a2f3p-2,
    0x1.8060180601806p-1, 0x1.2596010df763ap-2, 0x1.7e225515a4f1dp-1, 0x1.2b9303ab89d25p-2,
    0x1.7beb3922e017cp-1, 0x1.31871c9544185p-2, 0x1.79baa6bb6398bp-1, 0x1.3772662bfd85cp-2,
    0x1.77908119ac60dp-1, 0x1.3d54fa5c1f710p-2, 0x1.756cac201756dp-1, 0x1.432ef2a04e813p-2,
    0x1.734f0c541fe8dp-1, 0x1.49006804009d0p-2, 0x1.713786d9c7c09p-1, 0x1.4ec9732600269p-2,
    0x1.6f26016f26017p-1, 0x1.548a2c3add263p-2, 0x1.6d1a62681c861p-1, 0x1.5a42ab0f4cfe2p-2,
    0x1.6b1490aa31a3dp-1, 0x1.5ff3070a793d4p-2, 0x1.691473a88d0c0p-1, 0x1.659b57303e1f2p-2,
    0x1.6719f3601671ap-1, 0x1.6b3bb2235943dp-2, 0x1.6524f853b4aa3p-1, 0x1.70d42e2789236p-2,
    0x1.63356b88ac0dep-1, 0x1.7664e1239dbcfp-2, 0x1.614b36831ae94p-1, 0x1.7bede0a37afbfp-2,
    0x1.5f66434292dfcp-1, 0x1.816f41da0d495p-2, 0x1.5d867c3ece2a5p-1, 0x1.86e919a330ba1p-2,
    0x1.5babcc647fa91p-1, 0x1.8c5b7c858b48bp-2, 0x1.59d61f123ccaap-1, 0x1.91c67eb45a83ep-2,
    0x1.5805601580560p-1, 0x1.972a341135159p-2, 0x1.56397ba7c52e2p-1, 0x1.9c86b02dc0862p-2,
    0x1.54725e6bb82fep-1, 0x1.a1dc064d5b995p-2, 0x1.52aff56a8054bp-1, 0x1.a72a4966bd9e9p-2,
    0x1.50f22e111c4c5p-1, 0x1.ac718c258b0e5p-2, 0x1.4f38f62dd4c9bp-1, 0x1.b1b1e0ebdfc5ap-2,
    0x1.4d843bedc2c4cp-1, 0x1.b6eb59d3cf35cp-2, 0x1.4bd3edda68fe1p-1, 0x1.bc1e08b0dad0ap-2,
    0x1.4a27fad76014ap-1, 0x1.c149ff115f027p-2, 0x1.4880522014880p-1, 0x1.c66f4e3ff6ff9p-2,
    0x1.46dce34596066p-1, 0x1.cb8e0744d7acap-2, 0x1.453d9e2c776cap-1, 0x1.d0a63ae721e64p-2,
    0x1.43a2730abee4dp-1, 0x1.d5b7f9ae2c684p-2, 0x1.420b5265e5951p-1, 0x1.dac353e2c5955p-2,
    0x1.40782d10e6566p-1, 0x1.dfc859906d5b5p-2, 0x1.3ee8f42a5af07p-1, 0x1.e4c71a8687704p-2,
    0x1.3d5d991aa75c6p-1, 0x1.e9bfa659861f5p-2, 0x1.3bd60d9232955p-1, 0x1.eeb20c640ddf3p-2,
    0x1.3a524387ac822p-1, 0x1.f39e5bc811e5dp-2, 0x1.38d22d366088ep-1, 0x1.f884a36fe9ec1p-2,
    0x1.3755bd1c945eep-1, 0x1.fd64f20f61571p-2, 0x1.35dce5f9f2af8p-1, 0x1.011fab125ff8ap-1,
    0x1.34679ace01346p-1, 0x1.0389eefce633cp-1, 0x1.32f5ced6a1dfap-1, 0x1.05f14bd26459cp-1,
    0x1.3187758e9ebb6p-1, 0x1.0855c884b450ep-1, 0x1.301c82ac40260p-1, 0x1.0ab76bece14d2p-1,
    0x1.2eb4ea1fed14bp-1, 0x1.0d163ccb9d6b8p-1, 0x1.2d50a012d50a0p-1, 0x1.0f7241c9b497dp-1,
    0x1.2bef98e5a3711p-1, 0x1.11cb81787ccf8p-1, 0x1.2a91c92f3c105p-1, 0x1.1422025243d45p-1,
    0x1.293725bb804a5p-1, 0x1.1675cababa60ep-1, 0x1.27dfa38a1ce4dp-1, 0x1.18c6e0ff5cf07p-1,
    0x1.268b37cd60127p-1, 0x1.1b154b57da29ep-1, 0x1.2539d7e9177b2p-1, 0x1.1d610fe677003p-1,
    0x1.23eb79717605bp-1, 0x1.1faa34b87094cp-1, 0x1.22a0122a0122ap-1, 0x1.21f0bfc65beecp-1,
    0x1.21579804855e6p-1, 0x1.2434b6f483934p-1, 0x1.2012012012012p-1, 0x1.26762013430e0p-1,
    0x1.1ecf43c7fb84cp-1, 0x1.28b500df60783p-1, 0x1.1d8f5672e4abdp-1, 0x1.2af15f02640acp-1,
    0x1.1c522fc1ce059p-1, 0x1.2d2b4012edc9dp-1, 0x1.1b17c67f2bae3p-1, 0x1.2f62a99509546p-1,
    0x1.19e0119e0119ep-1, 0x1.3197a0fa7fe6ap-1, 0x1.18ab083902bdbp-1, 0x1.33ca2ba328994p-1,
    0x1.1778a191bd684p-1, 0x1.35fa4edd36ea0p-1, 0x1.1648d50fc3201p-1, 0x1.38280fe58797fp-1,
    0x1.151b9a3fdd5c9p-1, 0x1.3a5373e7ebdf9p-1, 0x1.13f0e8d344724p-1, 0x1.3c7c7fff73206p-1,
    0x1.12c8b89edc0acp-1, 0x1.3ea33936b2f5bp-1, 0x1.11a3019a74826p-1, 0x1.40c7a4880dceap-1,
    0x1.107fbbe011080p-1, 0x1.42e9c6ddf80bfp-1, 0x1.0f5edfab325a2p-1, 0x1.4509a5133bb0ap-1,
    0x1.0e40655826011p-1, 0x1.472743f33aaadp-1, 0x1.0d24456359e3ap-1, 0x1.4942a83a2fc07p-1,
    0x1.0c0a7868b4171p-1, 0x1.4b5bd6956e273p-1, 0x1.0af2f722eecb5p-1, 0x1.4d72d3a39fd01p-1,
    0x1.09ddba6af8360p-1, 0x1.4f87a3f5026e9p-1, 0x1.08cabb37565e2p-1, 0x1.519a4c0ba3446p-1,
    0x1.07b9f29b8eae2p-1, 0x1.53aad05b99b7cp-1, 0x1.06ab59c7912fbp-1, 0x1.55b9354b40bcep-1,
    0x1.059eea0727586p-1, 0x1.57c57f336f191p-1, 0x1.04949cc1664c5p-1, 0x1.59cfb25fae87fp-1,
    0x1.038c6b78247fcp-1, 0x1.5bd7d30e71c73p-1, 0x1.02864fc7729e9p-1, 0x1.5ddde57149923p-1,
    0x1.0182436517a37p-1, 0x1.5fe1edad18919p-1, 0x1.0080402010080p-1, 0x1.61e3efda46467p-1,
};
// ln of a positive normal number by the table (see above)
__device__ __forceinline__ double ln_tab(double v, const double *tab) {
    const uint32_t hi = (uint32_t)__double2hiint(v);
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), __double2loint(v));
    // the exponent as a double without the quarter-rate v_cvt_f64_i32: (2^52 + biased) - (2^52 + 1023)
    const double e = __hiloint2double(0x43300000, (int)(hi >> 20)) - 4503599627371519.0;
    const double2 row = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tab) + ((hi >> 9) & 0x7F0u));
    const double r = __builtin_fma(m, row.x, -1.0);
    double q = 1.0 / 5;
    q = __builtin_fma(q, r, -1.0 / 4);
    q = __builtin_fma(q, r, 1.0 / 3);
    q = __builtin_fma(q, r, -0.5);
    return __builtin_fma(e, 0x1.62e42fefa39efp-1, row.y + __builtin_fma(r * r, q, r));
}
constexpr double DB20_K10 = 0x1.15f2ced384f29p+2;      // 10 / ln 10
constexpr double DB20_KEPS = 0x1.dd8307784b277p-31;    // (20 / ln 10) 1e-10
// the series form: p = |X|^2 in [2^-13, 2^996)
__device__ __forceinline__ double db20_series(double p, const double *tab) {
    // 1 / |X| from the hardware estimate (v_rsq_f64, one quarter-rate instruction; two conversions around
    // v_rsq_f32 cost three): ~1e-8 of a term below 1e-7 dB
    return __builtin_fma(ln_tab(p, tab), DB20_K10, DB20_KEPS * __builtin_amdgcn_rsq(p));
}
// high word of p - high word of 2^-13, as unsigned: < DB20_SERIES_SPAN exactly when p is in [2^-13, 2^996)
// (zero, denormals, huge values, infinities and NaNs of either sign fall outside)
constexpr uint32_t DB20_SERIES_LO = 0x3F200000u, DB20_SERIES_SPAN = 0x7E300000u - 0x3F200000u;
__device__ __forceinline__ uint32_t db20_series_key(double p) { return (uint32_t)__double2hiint(p) - DB20_SERIES_LO; }
// 20 log10(|X| + 1e-10), any input.  Everything inline and short: the library fall-backs of db20() are ~250
// instructions per bin, eight times per line they made the row side's loop body larger than the instruction
// cache it shares.
__device__ __forceinline__ double db20_tab(cx<double> z, const double *tab) {
    const double p = __builtin_fma(z.x, z.x, z.y * z.y);
    if (db20_series_key(p) < DB20_SERIES_SPAN) return db20_series(p, tab);
    if (p < 1.0) {  // weak bins, zero and underflow: the expression as written, |X| + 1e-10 in [1e-10, 1.2e-2]
        const double arg = sqrt(p) + 1e-10;
        if (arg == 1e-10) return -200.0;  // silence is exactly 20 log10(1e-10)
        return ln_tab(arg, tab) * (2.0 * DB20_K10);
    }
    // |X|^2 beyond 2^996, infinite or NaN: rescale by 2^-600 (|X| + 1e-10 == |X| here)
    const double xs = z.x * 0x1p-600, ys = z.y * 0x1p-600;
    const double arg = __builtin_fma(xs, xs, ys * ys);
    if (!(arg < 1e300)) return arg;  // +inf stays +inf, NaN stays NaN (Math.log10 does the same)
    return __builtin_fma(ln_tab(arg, tab), DB20_K10, 600.0 * 0x1.8151824c7587fp+2);  // + 20 log10(2^600)
}
// The same for the NB bins of one thread: the series form for all of them without a branch, one test per WAVE
// for "some bin of some lane is outside its range", and only then the general form (for every bin: a bin inside the
// range gets the same value either way).
template <int NB> __device__ __forceinline__ void db20_tab_n(const cx<double> (&z)[NB], const double *tab, double (&d)[NB]) {
    uint32_t key = 0;
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const double p = __builtin_fma(z[m].x, z[m].x, z[m].y * z[m].y);
        const uint32_t k = db20_series_key(p);
        key = k > key ? k : key;
        d[m] = db20_series(p, tab);
    }
    if (__builtin_amdgcn_ballot_w64(key >= DB20_SERIES_SPAN) != 0) {
#pragma unroll
        for (int m = 0; m < NB; ++m) d[m] = db20_tab(z[m], tab);
    }
}

}  // namespace specgpu
