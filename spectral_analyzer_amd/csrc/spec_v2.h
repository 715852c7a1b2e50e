// spec_v2.h -- the packed-fp32 kernel family for every LDS-resident size
// (nfft = 64 ... 16384) and the little-endian / byte datatypes (cf32, ci16, cu8,
// ci8): spectrogram lines and Welch partial sums:
//   * plans put the SMALL radix first (256 = 16x16 ... 4096 = 16x16x16, 8192 =
//     32x16x16, 16384 = 32x32x16) so that every pass but the last has few distinct
//     twiddles (LDS tables of (c, d)) and the last pass is always
//     radix 16 with its 15 twiddles W_N^(r t) in registers;
//   * a line is owned by T = nfft/E threads, E = 16 points per thread up to 4096
//     points and 32 above (two LDS exchanges per line instead of three).  Up to
//     1024 points T is at most one wave, so the LDS exchanges are wave-local: no
//     s_barrier at all;
//   * every sub-line walks its own run of consecutive lines, so a hop of SH*T
//     samples is a shift of SH registers and each input byte is fetched once;
//   * buffer addressing with scalar offsets, packed complex math, prefetch of the
//     next line behind the current FFT, epilogue with one range test per thread.
#pragma once
#include "spec_fft.h"
#include "spec_fft_pk.h"
#include "spec_internal.h"

namespace specgpu {

namespace {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int L> struct Plan2;
#define SPEC_PLAN2(L, EE, NP, ...)                                                    \
    template <> struct Plan2<L> {                                                     \
        static constexpr int E = EE;                 /* points per thread */          \
        static constexpr int N = 1 << L, T = N / E, NPASS = NP;                       \
        static constexpr int radix[4] = {__VA_ARGS__};                                \
        static constexpr int WG = T <= 64 ? 256 : T; /* threads per workgroup */      \
        static constexpr int LPW = WG / T;           /* sub-lines per workgroup */    \
        static constexpr bool WAVE_LOCAL = T <= 64;  /* exchanges stay inside a wave */ \
        static constexpr int PADSH = E == 32 ? 5 : 4; /* one pad element per 2^PADSH (narrow-stride exchanges) */ \
        static constexpr int LINE = N + (N >> PADSH) + (T < 32 ? 16 : 0); /* LDS elements per sub-line, pads included */ \
    };
SPEC_PLAN2(6, 16, 2, 4, 16, 1, 1)   // 64 and 128 points: FFT core only; their kernel is spec_k_v2n.hip (wave-cooperative I/O)
SPEC_PLAN2(7, 16, 2, 8, 16, 1, 1)
SPEC_PLAN2(8, 16, 2, 16, 16, 1, 1)
SPEC_PLAN2(9, 16, 3, 2, 16, 16, 1)
SPEC_PLAN2(10, 16, 3, 4, 16, 16, 1)
SPEC_PLAN2(11, 16, 3, 8, 16, 16, 1)
SPEC_PLAN2(12, 16, 3, 16, 16, 16, 1)
SPEC_PLAN2(13, 32, 3, 32, 16, 16, 1)
SPEC_PLAN2(14, 32, 3, 32, 32, 16, 1)
#undef SPEC_PLAN2
// Plan id 113: 8192 points with 16-point threads (2 x 16 x 16 x 16, 512 threads per line), used by the fp64
// member of the family only (spec_v3d.h): a 32-point fp64 thread would need 128 registers for its line alone.
// The id stands where log2(nfft) stands in every template of this file; nothing derives N from it.
template <> struct Plan2<113> {
    static constexpr int E = 16, N = 8192, T = N / E, NPASS = 4;
    static constexpr int radix[4] = {2, 16, 16, 16};
    static constexpr int WG = T, LPW = 1;
    static constexpr bool WAVE_LOCAL = false;
    static constexpr int PADSH = 4;
    static constexpr int LINE = N + (N >> PADSH);
};

// Experiments kept out of this file (csrc/experiments/spec_v2_exp.h; build.py --variant v2rows / v2stamp): the 16384-point Welch plan
// 16 x (32 x 32) whose second exchange stays inside half a wave (Plan2<214>, measured slower: profiles/r05_rows.md), and the
// per-wave phase stamps of tools/v2_timeline.py.  The product sees the three fall-backs below.
#ifndef SPEC_V2_LATE_WAR
#define SPEC_V2_LATE_WAR 0
#endif
#if defined(SPEC_V2_ROWS) || defined(SPEC_V2_STAMPS)
#include "experiments/spec_v2_exp.h"
#else
template <int L> constexpr bool p2_rows() { return false; }
template <int L> constexpr bool p2_lds_twl() { return false; }  // last-pass twiddles from two LDS tables (experiment plan 314)
constexpr int P2_LDS_TWL_ENTRIES = 0;
template <int L> constexpr int p2_bin_reg(int m) { return m; }  // multiple of T in the bin index of register m after a transform
#define V2_STAMP(sp, id) do { (void)(sp); } while (0)
template <typename V> __device__ __forceinline__ void v2_fft_rows(V (&)[32], int, V *, const V *, V (&)[16], uint32_t *) {}
#endif

template <int L> constexpr int p2_P_of(int pass) {  // product of the radices before `pass`
    int p = 1;
    for (int q = 0; q < pass; ++q) p *= Plan2<L>::radix[q];
    return p;
}
template <int L, int PASS> constexpr int p2_P() { return p2_P_of<L>(PASS); }
// LDS twiddle tables of the middle passes 1 .. NPASS-2: entry (r, k) of pass z at
// off(z) + r*P_z + k, in v2f units
template <int L> constexpr int p2_tab_size(int pass) { return Plan2<L>::radix[pass] * p2_P_of<L>(pass); }
template <int L> constexpr int p2_tab_off_of(int pass) {
    int o = 0;
    for (int q = 1; q < pass; ++q) o += p2_tab_size<L>(q);
    return o;
}
template <int L, int PASS> constexpr int p2_tab_off() { return p2_tab_off_of<L>(PASS); }
template <int L> constexpr int p2_tab_entries() {
    if constexpr (p2_rows<L>()) return 1024;  // W_1024^(m l) of the rows' second pass
    else return p2_tab_off_of<L>(Plan2<L>::NPASS - 1);
}
template <int L, int ELEM = 8> constexpr size_t p2_lds_bytes() {  // ELEM: bytes per complex value (8 fp32, 16 fp64)
    return (size_t)Plan2<L>::LPW * Plan2<L>::LINE * ELEM + (size_t)(p2_tab_entries<L>() + (p2_lds_twl<L>() ? P2_LDS_TWL_ENTRIES : 0)) * ELEM;
}

// ---- raw sample formats (SS:40-59); SCALE is folded into the epilogue -------------
template <int KIND> struct Raw2;
template <> struct Raw2<K_CF32> {
    using type = u32x2;
    static constexpr int BPS = 8;
    static constexpr float SCALE = 1.0f;
    template <int AUX> static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int v, int s) {
        return __builtin_amdgcn_raw_buffer_load_b64(r, v, s, AUX);
    }
    template <typename V = v2f> static __device__ __forceinline__ V dec(type u) {
        return pk_make<V>(__uint_as_float(u.x), __uint_as_float(u.y));
    }
    static __device__ __forceinline__ type swap(type u) { return type{__builtin_bswap32(u.x), __builtin_bswap32(u.y)}; }
};
// cf64 (EDC:79-81): fp64 family only
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <> struct Raw2<K_CF64> {
    using type = u32x4;
    static constexpr int BPS = 16;
    static constexpr float SCALE = 1.0f;
    template <int AUX> static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int v, int s) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, v, s, AUX);
    }
    template <typename V = v2d> static __device__ __forceinline__ V dec(type u) {
        return pk_make<V>(__longlong_as_double((long long)(((uint64_t)u.y << 32) | u.x)),
                          __longlong_as_double((long long)(((uint64_t)u.w << 32) | u.z)));
    }
    static __device__ __forceinline__ type swap(type u) {  // reverse the eight bytes of each double
        return type{__builtin_bswap32(u.y), __builtin_bswap32(u.x), __builtin_bswap32(u.w), __builtin_bswap32(u.z)};
    }
};
template <> struct Raw2<K_CI16> {
    using type = uint32_t;
    static constexpr int BPS = 4;
    static constexpr float SCALE = 1.0f / 32768.0f;  // SS:44-45
    template <int AUX> static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int v, int s) {
        return __builtin_amdgcn_raw_buffer_load_b32(r, v, s, AUX);
    }
    template <typename V = v2f> static __device__ __forceinline__ V dec(type u) {
        using S = pk_scalar_t<V>;
        return V{(S)(int16_t)(u & 0xFFFFu), (S)((int32_t)u >> 16)};
    }
    // big-endian file: swap the two bytes of each 16-bit component
    static __device__ __forceinline__ type swap(type u) { return ((u & 0x00FF00FFu) << 8) | ((u >> 8) & 0x00FF00FFu); }
};
template <> struct Raw2<K_CI8> {
    using type = uint16_t;
    static constexpr int BPS = 2;
    static constexpr float SCALE = 1.0f / 128.0f;  // SS:58-59
    template <int AUX> static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int v, int s) {
        return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, v, s, AUX);
    }
    template <typename V = v2f> static __device__ __forceinline__ V dec(type u) {
        using S = pk_scalar_t<V>;
        return V{(S)(int8_t)(u & 0xFF), (S)(int8_t)(u >> 8)};
    }
    static __device__ __forceinline__ type swap(type u) { return u; }
};
template <> struct Raw2<K_CU8> {
    using type = uint16_t;
    static constexpr int BPS = 2;
    static constexpr float SCALE = 1.0f / 128.0f;  // SS:53-54: (b - 127.5) / 128
    template <int AUX> static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int v, int s) {
        return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, v, s, AUX);
    }
    template <typename V = v2f> static __device__ __forceinline__ V dec(type u) {
        using S = pk_scalar_t<V>;
        return V{(S)(u & 0xFF), (S)(u >> 8)} - pk_make<V>(127.5, 127.5);
    }
    static __device__ __forceinline__ type swap(type u) { return u; }
};

// LDS layout of an exchange written with a stride narrower than 16 elements: one
// pad element after every 16 (32 for the 32-point first pass), index a -> a + (a >> PADSH).
// The 16 lanes of a write group then hit 16 different 8-byte slots, the stride-T read
// loses at most one cycle to a single 2-way conflict (tools/lds_sim.py), and -- unlike an
// XOR swizzle -- every address splits into a per-thread base plus a compile-time
// immediate, so an exchange costs two address registers and no VALU work.
template <int SHIFT> constexpr int padn(int a) { return a + (a >> SHIFT); }
template <int SHIFT> __device__ __forceinline__ int padn_rt(int a) { return a + (a >> SHIFT); }

// One pass on the registers.  Middle passes take their twiddles from the LDS
// table `tab` (entry (r, k) at r*P + k); the last pass from
// the per-thread registers `twl` (W_N^(r t); the second butterfly of an E = 32
// thread sits T = N/32 further on: one more factor W_32^r, a compile-time constant).
template <int L, int PASS, typename V>
__device__ __forceinline__ void v2_pass(V (&v)[Plan2<L>::E], int t, const V *tab, V (&twl)[16]) {
    using PL = Plan2<L>;
    constexpr int E = PL::E, R = PL::radix[PASS], S = E / R, P = p2_P<L, PASS>();
    if constexpr (PASS > 0 && PASS < PL::NPASS - 1) {
        // k = (t + s T) mod P is the same for every s (P divides T): one table row for all butterflies
        static_assert(S == 1 || PL::T % P == 0, "butterflies of a thread share their twiddles");
        const V *row = tab + p2_tab_off<L, PASS>() + (t & (P - 1));
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const V q = row[r * P];
#pragma unroll
            for (int s = 0; s < S; ++s) v[s + r * S] = pk_cmul(v[s + r * S], q);
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        V u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = v[s + r * S];
        if constexpr (PASS == PL::NPASS - 1) {
            static_assert(R == 16 && S <= 2, "last pass: radix-16 butterflies");
            if constexpr (p2_lds_twl<L>()) {  // (experiment plan: the twiddles formed from two LDS tables behind the middle passes' tables, five at a time)
                static_assert(S == 1, "one butterfly per thread");
                const V *ta = tab + p2_tab_entries<L>() + (t & 31), *tb = ta - (t & 31) + 15 * 32 + ((t >> 5) & 31);
#pragma unroll
                for (int r0 = 1; r0 < 16; r0 += 5) {
#pragma unroll
                    for (int r = r0; r < r0 + 5; ++r) u[r] = pk_cmul(u[r], pk_cmul(ta[(r - 1) * 32], tb[(r - 1) * 32]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    u[r] = pk_cmul(u[r], twl[r]);
                    if (s == 1) u[r] = r == 8 ? pk_mul_mi(u[r]) : pk_cmul_const(u[r], kW32[r][0], kW32[r][1]);
                }
            }
        }
        pk_dft<R>(u);
#pragma unroll
        for (int r = 0; r < R; ++r) v[s + r * S] = u[r];
    }
}

// registers -> LDS after PASS.  Index of output r of butterfly i = t + s T:
//   hi*(R P) + r P + k,  k = i mod P, hi = i / P      (Stockham autosort)
template <int L, int PASS, typename V> __device__ __forceinline__ void v2_store(const V (&v)[Plan2<L>::E], int t, V *lds) {
    using PL = Plan2<L>;
    constexpr int R = PL::radix[PASS], S = PL::E / R, P = p2_P<L, PASS>(), SH = PL::PADSH;
    if constexpr (P >= 16) {  // wide stride: plain layout is conflict free
        static_assert(S == 1 || PL::T % P == 0, "butterfly s sits s*T*R elements further on");
        V *base = lds + ((t & ~(P - 1)) * R + (t & (P - 1)));
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) base[s * PL::T * R + r * P] = v[s + r * S];
    } else if constexpr (PASS == 0) {  // P == 1: a = i R + r, padded: pad(t R) + s*pad(T R) + r
        static_assert((PL::T * R) % (1 << SH) == 0 && R <= (1 << SH), "sub-line stride must be a multiple of the pad period");
        V *base = lds + padn_rt<SH>(t * R);
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) base[padn<SH>(s * PL::T * R) + r] = v[s + r * S];
    } else {  // 1 < P < 16, R = 16, S = 1: hi*(17 P) + k + pad(r P)
        static_assert(S == 1 && R == 16 && SH == 4, "middle passes are single radix-16 butterflies");
        V *base = lds + ((t / P) * (17 * P) + (t & (P - 1)));
#pragma unroll
        for (int r = 0; r < R; ++r) base[padn<4>(r * P)] = v[r];
    }
}
// LDS -> registers at stride T after the exchange written by PASS
// SPEC_V2_SINGLE_READS: the reads stay one ds_read_b64 each.  Left alone hipcc pairs reads whose addresses differ by a
// compile-time constant into ds_read2_b64 / ds_read2st64_b64, which take 8 LDS cycles per pair where two single reads take 4
// (MI355X_MICROARCH.md, LDS table: 128 against 256 B/clk/CU; tools/ldsbench.hip).  A volatile access is not paired; the
// address space is spelled out (a volatile access through a generic pointer would become a flat load).
#ifndef SPEC_V2_SINGLE_READS
#define SPEC_V2_SINGLE_READS 0
#endif
template <int L, int PASS, typename V> __device__ __forceinline__ void v2_load(V (&v)[Plan2<L>::E], int t, const V *lds) {
    using PL = Plan2<L>;
    constexpr int SH = PL::PADSH;
#if SPEC_V2_SINGLE_READS
    typedef const volatile __attribute__((address_space(3))) V *ro;
#else
    typedef const V *ro;
#endif
    if constexpr (p2_P<L, PASS>() >= 16) {
        const ro base = (ro)(lds + t);
#pragma unroll
        for (int m = 0; m < PL::E; ++m) v[m] = base[m * PL::T];
    } else {
        static_assert(PL::T < (1 << SH) || PL::T % (1 << SH) == 0, "pad(t + m T) = pad(t) + pad(m T)");
        const ro base = (ro)(lds + (PL::T >= (1 << SH) ? padn_rt<SH>(t) : t));
#pragma unroll
        for (int m = 0; m < PL::E; ++m) v[m] = base[padn<SH>(m * PL::T)];
    }
}

// W_N^(r k STEP) for the (r, k) of one middle pass
template <int L, int PASS, typename V> __device__ __forceinline__ void fill_tables(V *tab, const V *__restrict__ tw, int tid) {
    using PL = Plan2<L>;
    if constexpr (p2_rows<L>()) {  // entry (m, l) = W_1024^(m l) = W_N^(16 m l)
        for (int e = tid; e < 1024; e += PL::WG) tab[e] = tw[((e >> 5) * (e & 31) * 16) & (PL::N - 1)];
    } else if constexpr (PASS < PL::NPASS - 1) {
        constexpr int P = p2_P<L, PASS>(), R = PL::radix[PASS], STEP = PL::N / (P * R);
        for (int e = tid; e < R * P; e += PL::WG) {
            const V w0 = tw[(e / P) * (e % P) * STEP];
            tab[p2_tab_off<L, PASS>() + e] = w0;
        }
        fill_tables<L, PASS + 1>(tab, tw, tid);
    }
}

template <int L> __device__ __forceinline__ void v2_sync() {
    if constexpr (Plan2<L>::WAVE_LOCAL) {
        // the line's LDS region belongs to lanes of this wave only and a wave's DS
        // operations complete in issue order: a scheduling fence is all it takes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int L, int PASS = 0, typename V>
__device__ __forceinline__ void v2_fft(V (&v)[Plan2<L>::E], int t, V *lds, const V *tab, V (&twl)[16], uint32_t *sp = nullptr) {
    using PL = Plan2<L>;
    if constexpr (p2_rows<L>()) {
        v2_fft_rows(v, t, lds, tab, twl, sp);  // csrc/experiments/spec_v2_exp.h; never instantiated by the product
    } else {
        // SPEC_ABL_*: ablation builds of tools/ablate.sh (one stage removed, results wrong by construction,
        // DESIGN.md 4.7); never defined in a product build
#ifndef SPEC_ABL_NOFFT
        v2_pass<L, PASS>(v, t, tab, twl);
#endif
        V2_STAMP(sp, 2 + 5 * PASS);
        if constexpr (PASS + 1 < PL::NPASS) {
#ifndef SPEC_ABL_NOLDS
#if !defined(SPEC_ABL_NOBAR) && SPEC_V2_LATE_WAR
            v2_sync<L>();  // rounds 1-4: the write-after-read barrier in front of the stores (build.py --variant v2late)
#endif
            V2_STAMP(sp, 3 + 5 * PASS);
            v2_store<L, PASS>(v, t, lds);
            V2_STAMP(sp, 4 + 5 * PASS);
#ifndef SPEC_ABL_NOBAR
            v2_sync<L>();
#endif
            V2_STAMP(sp, 5 + 5 * PASS);
            v2_load<L, PASS>(v, t, lds);
#if !defined(SPEC_ABL_NOBAR) && !SPEC_V2_LATE_WAR
            // The write-after-read barrier for the NEXT exchange's stores, taken as soon as this exchange has been read (round 5)
            // instead of in front of those stores: the pass behind it -- and, after a line's last exchange, everything up to the
            // next line's first pass -- runs without a barrier between its arithmetic and its stores, so a wave that is ahead
            // stores while the others still compute.  Same number of barriers; cfg2 +3 %, cfg4 +2 %, 16384 / 32768 points +1 %
            // (profiles/r05_ab_raw.txt).  The very first exchange of a kernel has nothing in front of it to wait for.
            if constexpr (!PL::WAVE_LOCAL) v2_sync<L>();  // (its release fence waits for the loads: lgkmcnt(0))
#endif
            V2_STAMP(sp, 6 + 5 * PASS);
#endif
            v2_fft<L, PASS + 1>(v, t, lds, tab, twl, sp);
        }
    }
}

// 20 log10(|X| + 1e-10) of the spectrum
// scale * v (SS:80-81), one range test per thread
// BOUNDED: the input format cannot overflow |X|^2 in fp32 (integer samples): no upper range test
// p = |X|^2 of every bin, formed by the caller BEFORE it branches on the output format (v2_epilogue below): pk_norm is two
// spelled-out instructions per bin, which the compiler does not merge across the two arms of that branch
template <bool DB, bool BOUNDED, int E>
__device__ __forceinline__ void v2_epilogue_p(const v2f (&v)[E], const float (&p)[E], float scale, float (&d)[E]) {
    const float s2 = scale * scale;
    if constexpr (!DB) {
#pragma unroll
        for (int m = 0; m < E; ++m) d[m] = p[m] * s2;
    } else {
        float lo = fminf(fminf(p[0], p[1]), p[2]), hi = 0.0f;
#pragma unroll
        for (int m = 3; m + 1 < E; m += 2) lo = fminf(fminf(lo, p[m]), p[m + 1]);
        lo = fminf(lo, p[E - 1]);
        if constexpr (!BOUNDED) {
            hi = fmaxf(fmaxf(p[0], p[1]), p[2]);
#pragma unroll
            for (int m = 3; m + 1 < E; m += 2) hi = fmaxf(fmaxf(hi, p[m]), p[m + 1]);
            hi = fmaxf(hi, p[E - 1]);
        }
        constexpr float k10 = 3.01029995663981195f;
        const float off = k10 * __log2f(s2);
        if (lo * s2 > 1e-4f && hi < 1e37f) {  // a NaN fails the first test
#pragma unroll
            for (int m = 0; m < E; m += 2) {  // two bins per v_pk_fma_f32
                const v2f r = __builtin_elementwise_fma(v2f{__log2f(p[m]), __log2f(p[m + 1])}, v2f{k10, k10}, v2f{off, off});
                d[m] = r.x;
                d[m + 1] = r.y;
            }
        } else {
#pragma unroll
            for (int m = 0; m < E; ++m) d[m] = db20(cx<float>{v[m].x * scale, v[m].y * scale});
        }
    }
}

// db: OUT_DB20_F32, otherwise |X|^2 (OUT_POW_F32)
template <bool BOUNDED, int E>
__device__ __forceinline__ void v2_epilogue(const v2f (&v)[E], float scale, bool db, float (&d)[E]) {
    float p[E];
#pragma unroll
    for (int m = 0; m < E; ++m) p[m] = pk_norm(v[m]);
    if (db) v2_epilogue_p<true, BOUNDED, E>(v, p, scale, d);
    else v2_epilogue_p<false, BOUNDED, E>(v, p, scale, d);
}

#ifndef SPEC_V2_WIN_LDS
#define SPEC_V2_WIN_LDS 1
#endif
struct V2Args {
    const uint8_t *iq;      // first byte of unit 0, line 0
    uint64_t unit_stride;   // bytes between units (Welch: PSDs; spectrogram: one unit)
    uint32_t n_units, n_lines;  // lines (segments) per unit
    uint32_t hop, run;      // samples between lines; lines per sub-line run
    uint32_t wgs_per_unit;
    const void *tw, *win;
    int win_hann;           // 1: `win` is the periodic Hann table (32-point threads read a quarter of it from LDS, below); 2: all ones
    void *out;              // spectrogram: float[n_lines][N]; Welch: float slabs [unit][wg*LPW + q][N]
    int out_fmt;
    int be;                 // big-endian components
    const int32_t *sel;     // MODE 2: column of bin k (unshifted index) in the compact line, or -1
    uint32_t out_stride;    // MODE 2: floats per compact line
    void *final_out;        // MODE 1, one sub-line per unit: the finished PSDs (float[n_units][N]) instead of slabs
    double norm;            //         sum -> PSD factor (WelchArgs)
    int db;                 //         10 log10(psd + 1e-20)
    int rows;               // experiment library only (-DSPEC_V2_ROWS): MODE 1, 16384 points through Plan2<214>
};

// MODE 0: spectrogram lines (MC:980-999 around SS:33-85); MODE 1: Welch partial sums; MODE 2: spectrogram
// lines of which only the bins a later stage samples are stored (renderSpectrogram reads one bin per pixel
// row, MC:1280): a compact line of out_stride floats, bin k at column sel[k]
template <int L, int KIND, int SH, bool HAS_WIN, int MODE, bool BE, int OCC>
__global__ __launch_bounds__(Plan2<L>::WG, OCC) void v2_kernel(const V2Args a) {
    using PL = Plan2<L>;
    using RW = Raw2<KIND>;
    using raw_t = typename RW::type;
    constexpr int BPS = RW::BPS, N = PL::N, T = PL::T, E = PL::E;
    static_assert(!p2_rows<L>() || MODE == 1, "the row plan leaves a thread with bins sixteen apart: Welch sums only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, t = tid % T, q = tid / T;
    // bin of register m at the end of a transform: tb + m T (row plan: row g = t / 32, lane l = t % 32 -> g + 16 l + 512 m)
    const int tb = p2_rows<L>() ? (t >> 5) + 16 * (t & 31) : t;
    v2f *lds = reinterpret_cast<v2f *>(smem) + (size_t)q * PL::LINE;
    v2f *tab = reinterpret_cast<v2f *>(smem + (size_t)PL::LPW * PL::LINE * 8);
    const v2f *__restrict__ tw = static_cast<const v2f *>(a.tw);

    // ---- one-time set-up: LDS twiddle tables of the middle passes, last-pass registers
    if constexpr (PL::NPASS > 2) fill_tables<L, 1>(tab, tw, tid);
    v2f twl[16];
    if constexpr (p2_lds_twl<L>()) {
        v2f *tt = tab + p2_tab_entries<L>();
        for (int e = tid; e < 15 * 32; e += PL::WG) {
            tt[e] = tw[((e / 32 + 1) * (e % 32)) & (N - 1)];
            tt[15 * 32 + e] = tw[((e / 32 + 1) * (e % 32) * 32) & (N - 1)];
        }
    } else {
#pragma unroll
        for (int r = 1; r < 16; ++r) twl[r] = tw[(r * t) & (N - 1)];
    }
    // window values of this thread's samples: registers for 16-point threads, re-read from
    // the L2-resident table every line for 32-point threads (no room)
    constexpr bool WIN_REGS = HAS_WIN && E == 16;  // (cf32: 61.0 % with the window in registers and 8 spilled VGPRs, 59.3 % reloading it)
    const float *win = static_cast<const float *>(a.win);
    float w[E];
    if constexpr (WIN_REGS) {
#pragma unroll
        for (int m = 0; m < E; ++m) w[m] = win[t + m * T];
    }
    // Round 4: the Hann window of the 32-point threads from LDS.  w[N/2 - n] = 1 - w[n] and w[n + N/2] = 1 - w[n] leave N/4 + 1
    // distinct values (16 KiB at 16384 points: what the LDS has left); 16 LDS reads and 16 subtractions per line instead of 32
    // loads from L2 whose latency the two waves of a SIMD waited out in lock step at the top of every line.
    constexpr bool WIN_LDS = HAS_WIN && E == 32 && SPEC_V2_WIN_LDS != 0;
    float *wq = reinterpret_cast<float *>(smem + p2_lds_bytes<L>());
    const bool hann_lds = WIN_LDS && a.win_hann == 1;
    const bool ones = WIN_LDS && a.win_hann == 2;  // the rectangular Welch's table of ones: nothing to read
    if constexpr (WIN_LDS) {
        if (hann_lds)
            for (int e = tid; e <= N / 4; e += PL::WG) wq[e] = win[e];
    }
    // (the barrier that publishes the tables comes after the first line's loads have been issued)

    // ---- this workgroup's span: LPW consecutive runs of `run` lines of one unit
    const uint32_t unit = blockIdx.x / a.wgs_per_unit, wg = blockIdx.x % a.wgs_per_unit;
    const uint32_t line0 = wg * (uint32_t)PL::LPW * a.run;           // first line of the workgroup
    uint32_t lines_wg = a.n_lines - line0;                            // valid lines from there on
    if (lines_wg > (uint32_t)PL::LPW * a.run) lines_wg = PL::LPW * a.run;
    const uint32_t line_bytes = a.hop * BPS;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.iq) + (uint64_t)unit * a.unit_stride + (uint64_t)line0 * line_bytes, 0,
        (lines_wg - 1) * line_bytes + (uint32_t)N * BPS, 0x00020000);
    const int voff = (int)(q * a.run * line_bytes) + t * BPS;  // this sub-line's run, this thread's column
#ifndef SPEC_LD_AUX
#define SPEC_LD_AUX 2  // non-temporal loads
#endif
#ifndef SPEC_ST_AUX
#define SPEC_ST_AUX 2  // non-temporal stores
#endif
    constexpr int AUX = SPEC_LD_AUX, ST_AUX = SPEC_ST_AUX;
    // overlap shift: SH > 0 promises hop == SH * T, i.e. the line slides by SH registers
    constexpr int NEW = SH > 0 ? SH : E;

    raw_t raw[E];
#pragma unroll
    for (int m = 0; m < E; ++m) raw[m] = RW::template load<AUX>(src, voff, m * T * BPS);

    float acc[E];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int m = 0; m < E; ++m) acc[m] = 0.0f;
    }
    __amdgpu_buffer_rsrc_t dst;
    int ovoff = 0;
    if constexpr (MODE == 0) {
        dst = __builtin_amdgcn_make_buffer_rsrc(static_cast<float *>(a.out) + (uint64_t)line0 * N, 0,
                                                lines_wg * (uint32_t)N * 4u, 0x00020000);
        ovoff = (int)(q * a.run * (uint32_t)N * 4u) + t * 4;
    }
    int pxo[MODE == 2 ? E : 1];  // MODE 2: byte offset of each of this thread's bins in its compact line
    (void)pxo;
    if constexpr (MODE == 2) {
        dst = __builtin_amdgcn_make_buffer_rsrc(static_cast<float *>(a.out) + (uint64_t)line0 * a.out_stride, 0,
                                                lines_wg * a.out_stride * 4u, 0x00020000);
        ovoff = (int)(q * a.run * a.out_stride * 4u);
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int col = a.sel[t + m * T];
            pxo[m] = col >= 0 ? ovoff + col * 4 : 0x7FFFFFFF;  // past the descriptor: the store is dropped
        }
    }
    // sub-lines whose run lies past the end read zeros and store nothing (descriptor bounds)
    const uint32_t my_first = q * a.run;
    const uint32_t my_lines = my_first >= lines_wg ? 0 : (lines_wg - my_first < a.run ? lines_wg - my_first : a.run);
    // 50 % overlap, cf32 spectrogram: the two register halves swap roles from one line to the next
    // (two copies of the loop body) instead of being moved; every other variant shifts the
    // registers down (the doubled body costs them 8-40 spilled VGPRs, more than the moves)
    constexpr bool PINGPONG = SH * 2 == E && KIND == K_CF32 && !HAS_WIN && MODE == 0;
    // Whole-workgroup lines (LPW == 1, q == 0): the trip count is the same for every thread.  The ping-pong kernels SAY so
    // (readfirstlane): hipcc cannot see it through tid / T, and with the second copy of the body under `if (line + 1 < iters)` it
    // took the loop for a divergent one, kept the byte offset of the next line in a VGPR and wrapped every load of the line in a
    // waterfall loop (v_readfirstlane / v_cmp / s_and_saveexec / load / s_xor / s_cbranch_execnz: three vector and five scalar
    // instructions and a branch per load -- 24 of the 405 vector instructions of a 4096-point line, rounds 1-5; the build now
    // fails on such a loop, build.py waterfall_findings).  The single-copy kernels never had one and keep their code as measured
    // (with a scalar trip count the 16384-point Welch kernel is scheduled differently and loses 2 %).
#ifndef SPEC_V2_VECTOR_TRIP
#define SPEC_V2_VECTOR_TRIP 0  // 1: as in rounds 1-5 (build.py --variant v2wfall, for the A/B)
#endif
    const uint32_t iters = PL::WAVE_LOCAL ? a.run  // whole-workgroup lines: LPW == 1
                           : PINGPONG && !SPEC_V2_VECTOR_TRIP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)my_lines) : my_lines;

    if constexpr (PL::NPASS > 2) __syncthreads();  // LDS twiddle tables visible

#ifdef SPEC_V2_STAMPS
    // [wave][segment 0..2][16] words behind everything else in LDS; a.out_stride = first stamped segment, a.sel = where they go
    uint32_t *const stamp0 = reinterpret_cast<uint32_t *>(smem + ((p2_lds_bytes<L>() + ((size_t)N / 4 + 1) * 4 + 15) & ~(size_t)15)) + (tid >> 6) * 48;
#endif
    auto do_line = [&](uint32_t line, auto phase_tag) {
        constexpr int PH = PINGPONG ? decltype(phase_tag)::value : 0;  // physical register of sample m: (m + PH*SH) mod E
        uint32_t *sp = nullptr;
#ifdef SPEC_V2_STAMPS
        if (MODE == 1 && line >= a.out_stride && line < a.out_stride + 3) sp = stamp0 + (line - a.out_stride) * 16;
#endif
        V2_STAMP(sp, 0);
        v2f v[E];
        if constexpr (HAS_WIN && !WIN_REGS) {
            if (hann_lds) {
                // (q1: based at the LOWEST address read through it -- DS offsets are unsigned, and with negative ones hipcc kept
                // eight address registers, one per read, alive across the loop)
                const float *q0 = wq + t, *q1 = wq + (N / 2 - (E / 2 - 1) * T - t);
#pragma unroll
                for (int m = 0; m < E / 2; ++m) {  // n = t + m T < N/2;  w[N/2 - n] = 1 - w[n],  w[n + N/2] = 1 - w[n]
                    const float x = m < E / 4 ? q0[m * T] : q1[(E / 2 - 1 - m) * T];
                    // 1 - x spelled as an instruction: left to hipcc, its SLP vectorizer pairs two of these subtractions into a
                    // v_pk_add_f32 behind two v_mov and unpacks the result with two more (40 instructions per line instead of 16)
                    float y;
#if SPEC_PK_NORM_ASM
                    asm("v_sub_f32_e32 %0, 1.0, %1" : "=v"(y) : "v"(x));
#else
                    y = 1.0f - x;  // (build.py --variant slpnorm: the compiler's form, for the A/B)
#endif
                    w[m] = m < E / 4 ? x : y;
                    w[m + E / 2] = m < E / 4 ? y : x;
                }
            } else if (ones || p2_rows<L>()) {  // (the row plan is launched with the Hann window or the table of ones only: v2_launch_n)
#pragma unroll
                for (int m = 0; m < E; ++m) w[m] = 1.0f;
            } else {
                const float *wp = win;
                asm volatile("" : "+s"(wp));  // keep the loads inside the loop (LICM would pin E VGPRs)
#pragma unroll
                for (int m = 0; m < E; ++m) w[m] = wp[t + m * T];
            }
        }
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const raw_t r = raw[(m + PH * SH) % E];
            v[m] = RW::dec(BE ? RW::swap(r) : r);  // SMH:87-91 byte order
        }
        if constexpr (HAS_WIN) {
#pragma unroll
            for (int m = 0; m < E; ++m) v[m] *= v2f{w[m], w[m]};
        }
        if constexpr (SH > 0 && SH < E && !PINGPONG) {
#pragma unroll
            for (int m = 0; m < E - SH; ++m) raw[m] = raw[m + SH];
        }
        const int next_off = (int)((line + 1) * line_bytes);
#pragma unroll
        for (int m = E - NEW; m < E; ++m)  // the next line is one phase further on
            raw[PINGPONG ? (m + (1 - PH) * SH) % E : m] = RW::template load<AUX>(src, voff, next_off + m * T * BPS);

        V2_STAMP(sp, 1);
        v2_fft<L>(v, t, lds, tab, twl, sp);

        if constexpr (MODE == 1) {
            if (line < my_lines) {
#pragma unroll
                for (int m = 0; m < E; ++m)  // two chained FMAs per point (as a sum of pk_norm: multiply, FMA and an add)
                    acc[m] = pk_norm_acc(v[m], acc[m]);
            }
            V2_STAMP(sp, 15);
        } else {
            float d[E];
            constexpr bool BOUNDED = KIND != K_CF32;
            v2_epilogue<BOUNDED, E>(v, RW::SCALE, a.out_fmt == OUT_DB20_F32, d);
            if constexpr (MODE == 2) {
                const int row_off = (int)(line * a.out_stride * 4u);
#pragma unroll
                for (int m = 0; m < E; ++m)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d[m]), dst, pxo[m], row_off, ST_AUX);
                return;
            }
            const int out_off = (int)(line * (uint32_t)N * 4u);
#ifdef SPEC_ABL_NOSTORE
            const bool do_store = d[0] == 123.456f;
#else
            constexpr bool do_store = true;
#endif
            if (do_store)
            {
#pragma unroll
                for (int m = 0; m < E; ++m)  // (t + m T + N/2) mod N   (SS:78)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d[m]), dst, ovoff,
                                                          out_off + ((m + E / 2) & (E - 1)) * T * 4, ST_AUX);
            }
        }
    };
    // The first line's samples have landed BEFORE the loop is entered (a wait the first transform needs anyway).  Without it
    // hipcc's wait-count insertion merges the loop entry (16 / 32 loads in flight, no store) with the back edge (the next line's
    // loads in flight BEHIND them the line's 16 output stores) into the smaller count of the two, and every line began with
    // s_waitcnt vmcnt(0) ... vmcnt(12): it sat out the completion of the output stores it had issued an instant before
    // (rounds 1-5; with the entry path empty the header waits are vmcnt(16 + k): the loads only).
#ifndef SPEC_V2_ENTRY_WAIT
#define SPEC_V2_ENTRY_WAIT 1  // 0: as in rounds 1-5 (build.py --variant v2nowait, for the A/B)
#endif
    if constexpr (SPEC_V2_ENTRY_WAIT != 0 && MODE != 1) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched
    if constexpr (PINGPONG) {
        for (uint32_t line = 0; line < iters; line += 2) {
            do_line(line, std::integral_constant<int, 0>{});
            if (line + 1 < iters) do_line(line + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (uint32_t line = 0; line < iters; ++line) do_line(line, std::integral_constant<int, 0>{});
    }
#ifdef SPEC_V2_STAMPS
    if (MODE == 1 && a.sel && blockIdx.x < 8 && (tid & 63) < 48)
        const_cast<int32_t *>(a.sel)[(blockIdx.x * (PL::WG / 64) + (tid >> 6)) * 48 + (tid & 63)] = (int32_t)stamp0[tid & 63];
#endif
    if constexpr (MODE == 1) {
        constexpr float s2 = RW::SCALE * RW::SCALE;
        if constexpr (PL::LPW == 1) {
            if (a.final_out) {
                // this workgroup has summed every segment of its PSD: finish it here, with welch_finalize_kernel's
                // arithmetic for a single slab (fp32 sum -> double, * norm, fftshift, optional dB)
                float *psd = static_cast<float *>(a.final_out) + (uint64_t)unit * N;
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const double v = (double)(acc[m] * s2) * a.norm;
                    psd[(tb + ((p2_bin_reg<L>(m) + E / 2) & (E - 1)) * T)] = a.db ? (float)(10.0 * log10(v + 1e-20)) : (float)v;
                }
                return;
            }
        }
        // one fp32 slab per sub-line (zeros for idle ones); welch_finalize_kernel sums them in order
        float *slab = static_cast<float *>(a.out) + ((uint64_t)blockIdx.x * PL::LPW + q) * N;
#pragma unroll
        for (int m = 0; m < E; ++m) slab[tb + p2_bin_reg<L>(m) * T] = acc[m] * s2;
    }
}

template <int L, int KIND, int SH, bool HAS_WIN, int MODE, bool BE = false>
hipError_t v2_launch1(const V2Args &a, hipStream_t s) {
    using PL = Plan2<L>;
#ifdef SPEC_V2_STAMPS
    constexpr size_t lds = ((p2_lds_bytes<L>() + ((size_t)PL::N / 4 + 1) * 4 + 15) & ~(size_t)15) + (PL::WG / 64) * 48 * 4;
#else
    constexpr size_t lds = p2_lds_bytes<L>() + (HAS_WIN && PL::E == 32 && SPEC_V2_WIN_LDS != 0 ? ((size_t)PL::N / 4 + 1) * sizeof(float) : 0);
#endif
    static_assert(lds <= 160 * 1024, "one workgroup's LDS");
    // Minimum waves per SIMD asked of the register allocator, chosen so that the kernel does
    // not spill: about 104 VGPRs of FFT state + 32 (cf32) or 16 raw sample registers + 16 for a
    // window + 16 for Welch sums.  32-point threads (8192 / 16384 points) take the full 256.
    constexpr int NEED = 104 + (KIND == K_CF32 ? 32 : 16) + (HAS_WIN ? 16 : 0) + (MODE != 0 ? 16 : 0);
#ifdef SPEC_FORCE_OCC  // experiments only
    constexpr int WAVES_PER_SIMD = PL::E == 32 ? 2 : SPEC_FORCE_OCC;
#else
    constexpr int WAVES_PER_SIMD = PL::WG == 1024 ? 4 : PL::E == 32 ? 2 : NEED <= 120 ? 4 : NEED <= 152 ? 3 : 2;
#endif
    auto kern = v2_kernel<L, KIND, SH, HAS_WIN, MODE, BE, WAVES_PER_SIMD>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.n_units * a.wgs_per_unit), dim3(PL::WG), lds, s, a);
    return hipGetLastError();
}

// Instantiation matrix: register-reuse variants for hop = N/2 (50 % overlap) in every format, for
// hop = N/4 (75 %) in cf32 / ci16 and in every Welch kernel; every other hop takes SH = 0 (the overlap then
// comes from L2).  Big-endian files: the spectrogram kernels have the same register-reuse variants (the raw registers
// hold the file's bytes, the swap happens at decode: 50 % overlap without a window since round 3, the windowed and the
// 75 % ones since round 5), Welch and the redraw mode one SH = 0 variant each.
// Welch always multiplies by a window table (all ones for the rectangular window).
template <int L, int KIND, int MODE> hipError_t v2_launch_sh(const V2Args &a, hipStream_t s) {
    constexpr int N = Plan2<L>::N, E = Plan2<L>::E;
    constexpr bool WIDE = KIND == K_CF32 || KIND == K_CI16;  // formats with a byte order
    if constexpr (WIDE) {  // big-endian files: one variant per mode (no register reuse), byte swap at decode
        if (a.be) {
            if constexpr (MODE == 1) return v2_launch1<L, KIND, 0, true, 1, true>(a, s);
            else if constexpr (MODE == 0) {
                // (round 5: the windowed 50 % and both 75 % variants as well -- re-reading the overlap from L2 these cells ran at
                // 0.35 / 0.46 of 8 TB/s where the little-endian kernels reach 0.52 / 0.62: profiles/r05_cells.txt)
                if (a.hop == N / 2) return a.win ? v2_launch1<L, KIND, E / 2, true, 0, true>(a, s) : v2_launch1<L, KIND, E / 2, false, 0, true>(a, s);
                if (a.hop == N / 4) return a.win ? v2_launch1<L, KIND, E / 4, true, 0, true>(a, s) : v2_launch1<L, KIND, E / 4, false, 0, true>(a, s);
                return a.win ? v2_launch1<L, KIND, 0, true, 0, true>(a, s) : v2_launch1<L, KIND, 0, false, 0, true>(a, s);
            } else return a.win ? v2_launch1<L, KIND, 0, true, MODE, true>(a, s) : v2_launch1<L, KIND, 0, false, MODE, true>(a, s);
        }
    }
    if constexpr (MODE == 1) {
        if (a.hop == N / 2) return v2_launch1<L, KIND, E / 2, true, 1>(a, s);
        if (a.hop == N / 4) return v2_launch1<L, KIND, E / 4, true, 1>(a, s);
        return v2_launch1<L, KIND, 0, true, 1>(a, s);
    } else {
        const bool win = a.win != nullptr;
        if constexpr (MODE == 2) {  // the redraw path: the reference's hop == nfft and 50 % overlap, no window
            if (a.hop == N / 2 && !win) return v2_launch1<L, KIND, E / 2, false, 2>(a, s);
            return win ? v2_launch1<L, KIND, 0, true, 2>(a, s) : v2_launch1<L, KIND, 0, false, 2>(a, s);
        } else {
            if (a.hop == N / 2) return win ? v2_launch1<L, KIND, E / 2, true, 0>(a, s) : v2_launch1<L, KIND, E / 2, false, 0>(a, s);
            if constexpr (WIDE) {
                if (a.hop == N / 4) return win ? v2_launch1<L, KIND, E / 4, true, 0>(a, s) : v2_launch1<L, KIND, E / 4, false, 0>(a, s);
            }
            return win ? v2_launch1<L, KIND, 0, true, 0>(a, s) : v2_launch1<L, KIND, 0, false, 0>(a, s);
        }
    }
}

template <int L, int MODE> hipError_t v2_launch_kind(const V2Args &a, int kind, hipStream_t s) {
    switch (kind) {
    case K_CF32: return v2_launch_sh<L, K_CF32, MODE>(a, s);
    case K_CI16: return v2_launch_sh<L, K_CI16, MODE>(a, s);
    case K_CU8: return v2_launch_sh<L, K_CU8, MODE>(a, s);
    case K_CI8: return v2_launch_sh<L, K_CI8, MODE>(a, s);
    default: return hipErrorInvalidValue;
    }
}

template <int MODE> hipError_t v2_launch_n(const V2Args &a, int log2n, int kind, hipStream_t s) {
    switch (log2n) {
    case 8: return v2_launch_kind<8, MODE>(a, kind, s);
    case 9: return v2_launch_kind<9, MODE>(a, kind, s);
    case 10: return v2_launch_kind<10, MODE>(a, kind, s);
    case 11: return v2_launch_kind<11, MODE>(a, kind, s);
    case 12: return v2_launch_kind<12, MODE>(a, kind, s);
    case 13: if constexpr (MODE != 2) return v2_launch_kind<13, MODE>(a, kind, s); else return hipErrorInvalidValue;
    case 14:
#ifdef SPEC_V2_ROWS
        if constexpr (MODE == 1) {
            if (a.rows == 2) return v2_launch_kind<314, MODE>(a, kind, s);  // 1024 threads x 16 points
            if (a.rows && (a.win_hann == 1 || a.win_hann == 2)) return v2_launch_kind<214, MODE>(a, kind, s);
        }
#endif
        if constexpr (MODE != 2) return v2_launch_kind<14, MODE>(a, kind, s); else return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

}  // namespace specgpu
