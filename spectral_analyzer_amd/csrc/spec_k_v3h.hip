// spec_k_v3h.hip -- 16384-point fp64 lines in ONE workgroup: the strict-parity pipeline (cf64 recordings, SPEC_OUT_*_F64
// outputs; SpectralService.java:33-85 computes in double) one size above the fp64 family's largest plan.
//
// A 16384-point line of fp64 complex values is 256 KiB -- the same bytes as the 32768-point fp32 line of
// spec_k_v2h.hip, and the same answer: the first radix-2 step (decimation in frequency) in registers on the way in,
//     a[n] = x[n] + x[n + H],   b[n] = (x[n] - x[n + H]) W_N^n,   H = N/2 = 8192,  n < H
//     X[2k] = FFT_H(a)[k],      X[2k + 1] = FFT_H(b)[k]
// then two 8192-point transforms of the fp64 family (Plan2<113>: 512 threads, 16 points each, radix 2 x 16 x 16 x 16)
// one after the other through the same LDS buffer.  Until round 4 these lines took the four-step team kernel
// (spec_k_team.hip; 0.32 / 0.23 / 0.14 of the HBM roofline for cf64 -> f64, cf64 -> f32, cf32 -> f64).
//   * thread t owns n = t + 512 m (m < 16) of both halves: W_N^n = W_N^t W_32^m, one per-thread twiddle and 16 constants;
//   * the even bins wait as 16 finished doubles per thread while the odd half is transformed, then leave in PAIRS
//     (16-byte stores for double output, 8-byte for float), fftshift (SS:78) folded into the index;
//   * registers (256 at two waves per SIMD): transform state 64, parked 64 -- cf64: the difference lo - hi; every narrower
//     format: the raw samples (decoded twice), and at hop = N/2 the upper half stays as the next line's lower half.  The
//     family keeps the last pass's fifteen twiddles W_H^(r t) in 60 registers; here they are formed when needed from two
//     LDS tables, W_H^(r (t mod 32)) W_H^(32 r (t div 32)) (the second IS the third pass's table), and the 60 registers
//     hold the next line's samples in flight behind the second transform instead.
#include "spec_v3d.h"
#include "spec_v2h.h"  // kW64: the per-register constants of the paired kernel below

namespace specgpu {

namespace {

struct V3hArgs {
    const uint8_t *iq;     // first byte of line 0
    uint32_t n_lines, hop, run;  // run: consecutive lines per workgroup
    const void *tw_half;   // v2d W_8192^m
    const void *tw_full;   // v2d W_16384^m
    const void *win;       // non-null: Hann window (computed from the twiddles, the table is not read)
    void *out;
    int out_fmt;
};

// d * W_32^M * wt
template <int M> __device__ __forceinline__ v2d v3h_twiddle(v2d d, v2d wt) { return pk_cmul(pk_mul_w32<M>(d), wt); }
template <typename F, int... M> __device__ __forceinline__ void v3h_for_each(F &&f, std::integer_sequence<int, M...>) {
    (f(std::integral_constant<int, M>{}), ...);
}

// 20 log10(|X| + 1e-10) (SS:80-81) by the table logarithm of spec_fft.h, four bins at a time as in spec_v3d.h; or |X|^2
template <int E> __device__ __forceinline__ void v3h_epilogue(const v2d (&v)[E], double scale, bool db, const double *dbt, double (&d)[E]) {
    constexpr int G = 4;
#pragma unroll
    for (int g = 0; g < E; g += G) {
        double val[G];
        if (db) {
            cx<double> z[G];
#pragma unroll
            for (int j = 0; j < G; ++j) z[j] = cx<double>{v[g + j].x * scale, v[g + j].y * scale};
            db20_tab_n<G>(z, dbt, val);
        } else {
#pragma unroll
            for (int j = 0; j < G; ++j) val[j] = pk_norm(v[g + j]) * (scale * scale);
        }
#pragma unroll
        for (int j = 0; j < G; ++j) d[g + j] = val[j];
        __builtin_amdgcn_sched_barrier(0);  // one group's logarithms at a time (registers)
    }
}

// A 16-byte buffer store with its wait states attached.  MEASURED on gfx950 (ROCm 7.2): a vector-ALU write to one of the
// store's four data registers in the instruction slot behind a `buffer_store_dwordx4 ... s<offset> offen` reaches the
// store -- the last lanes of the wave leave with the NEW value (lanes 12-15 of every 16, second wave of a SIMD, one run in
// three).  hipcc assembles the (even, odd) pair in a borrowed register tuple and restores the borrowed half with a v_mov
// right behind the store; its hazard recogniser pads that pattern only for stores WITHOUT a scalar offset register.
// The empty-looking asm reads the tuple, so the restoring moves are ordered behind it, and carries the two wait states.
// tools/check_store_hazard.py checks the disassembly (tests/test_store_hazard.py).
template <int AUX> __device__ __forceinline__ void v3h_store_b128(u32x4 d, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, voff, soff, AUX);
    asm volatile("s_nop 1" : : "v"(d) : "memory");
}

// the family's passes (spec_v2.h v2_fft) with the last pass's twiddles from LDS: tab_lo[(r - 1) 32 + j] = W_H^(r j), j < 32
template <int PASS = 0>
__device__ __forceinline__ void v3h_fft(v2d (&v)[16], int t, v2d *lds, const v2d *tab, const v2d *tab_lo) {
    if constexpr (PASS < 3) {
        if constexpr (PASS == 0) {
            v2d none[16];  // (only the last pass reads its twiddle registers)
            v2_pass<113, 0>(v, t, tab, none);
        } else {  // v2_pass with the table reads five at a time (registers, as in the last pass below)
            constexpr int P = p2_P<113, PASS>();
            const v2d *row = tab + p2_tab_off<113, PASS>() + (t & (P - 1));
#pragma unroll
            for (int r0 = 1; r0 < 16; r0 += 5) {
#pragma unroll
                for (int r = r0; r < r0 + 5; ++r) v[r] = pk_cmul(v[r], row[r * P]);
                __builtin_amdgcn_sched_barrier(0);
            }
            pk_dft16(v);
        }
        __syncthreads();  // WAR: the previous exchange has been read by everyone
        v2_store<113, PASS>(v, t, lds);
        __syncthreads();
        v2_load<113, PASS>(v, t, lds);
        v3h_fft<PASS + 1>(v, t, lds, tab, tab_lo);
    } else {
        static_assert(p2_P<113, 2>() == 32 && Plan2<113>::radix[2] == 16, "third pass's table: entry (r, k) = W_H^(16 r k) at r 32 + k");
        const v2d *ra = tab_lo + (t & 31);
        const v2d *rb = tab + p2_tab_off<113, 2>() + 2 * (t >> 5);  // W_H^(32 r h) = entry (r, 2 h)
        // five at a time: left alone the scheduler requests all thirty table entries first (120 registers)
#pragma unroll
        for (int r0 = 1; r0 < 16; r0 += 5) {
#pragma unroll
            for (int r = r0; r < r0 + 5; ++r) v[r] = pk_cmul(v[r], pk_cmul(ra[(r - 1) * 32], rb[r * 32]));
            __builtin_amdgcn_sched_barrier(0);
        }
        pk_dft16(v);
    }
}

#ifndef V3H_EARLY_LO_FIRST
#define V3H_EARLY_LO_FIRST 1
#endif
#ifndef V3H_EARLY_REGS
#define V3H_EARLY_REGS 32  // registers' worth of the next line requested BEFORE the second transform (cf64: 8 of 32 samples)
#endif

// REUSE: hop == N/2 and a format whose raw samples are parked -- the raw upper half stays in registers
template <int KIND, bool HAS_WIN, bool BE, bool REUSE>
__global__ __launch_bounds__(Plan2<113>::T, 2) void v3h_kernel(const V3hArgs a) {
    using PL = Plan2<113>;
    using RW = Raw2<KIND>;
    using raw_t = typename RW::type;
    constexpr int BPS = RW::BPS, H = PL::N, N = 2 * H, T = PL::T, E = PL::E;
    constexpr bool PARK_RAW = KIND != K_CF64;
    constexpr int RAW_REGS = sizeof(raw_t) <= 4 ? 1 : (int)sizeof(raw_t) / 4;
    // the part of the next line requested BEFORE the second transform; the rest is requested behind the second epilogue.
    // The LOWER half first: at 50 % overlap it is the half this workgroup read one line ago, and the sooner it is read again
    // the more of it is still in L2 (each CU keeps 128 KiB in flight there, 4 MiB per XCD -- all of the L2; with the upper
    // half first, 1.5 line times between the two readings, none of it was: reads 1.9x the new samples, V3H_EARLY_LO_FIRST = 0)
    constexpr int N_EARLY = (V3H_EARLY_REGS - (REUSE ? E * RAW_REGS : 0)) / RAW_REGS;  // (REUSE: the kept half counts)
    constexpr bool LO_FIRST = V3H_EARLY_LO_FIRST && !REUSE;
    constexpr int EARLY_A = N_EARLY < E ? N_EARLY : E, EARLY_B = N_EARLY - EARLY_A < E ? N_EARLY - EARLY_A : E;
    constexpr int EARLY_LO = REUSE ? 0 : (LO_FIRST ? EARLY_A : EARLY_B);
    constexpr int EARLY_HI = LO_FIRST ? EARLY_B : EARLY_A;
    static_assert(!REUSE || PARK_RAW, "register reuse needs the raw halves parked");
    static_assert(E == 16 && N / T == 32, "16 points per thread and half: n = t + T m, W_N^(T m) = W_32^m");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    v2d *lds = reinterpret_cast<v2d *>(smem);
    v2d *tab = reinterpret_cast<v2d *>(smem + (size_t)PL::LINE * 16);
    const v2d *__restrict__ tw = static_cast<const v2d *>(a.tw_half);

    fill_tables<113, 1>(tab, tw, t);
    double *dbt_w = reinterpret_cast<double *>(smem + p2_lds_bytes<113, 16>());
    for (int e = t; e < DB20_TAB_DOUBLES; e += T) dbt_w[e] = DB20_TAB[e];
    const double *dbt = dbt_w;
    v2d *wtab = reinterpret_cast<v2d *>(dbt_w + DB20_TAB_DOUBLES);  // W_32^m = W_N^(512 m), m < 16 (the Hann window's cosine)
    if (HAS_WIN && t < 16) wtab[t] = static_cast<const v2d *>(a.tw_full)[T * t];
    v2d *tab_lo = wtab + 16;  // W_H^(r j), r = 1 .. 15, j < 32
    if (t < 15 * 32) tab_lo[t] = tw[(t / 32 + 1) * (t % 32)];
    const v2d wt = static_cast<const v2d *>(a.tw_full)[t];  // W_N^t
    const bool out64 = a.out_fmt == OUT_DB20_F64 || a.out_fmt == OUT_POW_F64;
    const bool db = a.out_fmt == OUT_DB20_F64 || a.out_fmt == OUT_DB20_F32;
    const uint32_t esz = out64 ? 8u : 4u;

    const uint32_t line0 = blockIdx.x * a.run;
    uint32_t lines_wg = a.n_lines - line0;
    if (lines_wg > a.run) lines_wg = a.run;
    const uint32_t line_bytes = a.hop * BPS;
    uint32_t lw = lines_wg;  // (behind an empty asm: see spec_k_v2h.hip -- keeps the descriptors' sizes in scalar registers)
    asm volatile("" : "+s"(lw));
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.iq) + (uint64_t)line0 * line_bytes, 0, (lw - 1) * line_bytes + (uint32_t)N * BPS, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<uint8_t *>(a.out) + (uint64_t)line0 * N * esz, 0, lw * (uint32_t)N * esz, 0x00020000);
    const int voff = t * BPS, ovoff = t * 2 * (int)esz;
    constexpr int AUX = 2, ST_AUX = 2;  // non-temporal, as the family
    // cf64 (no register reuse): the overlapped half is read again one line later -- cached on its first reading
    constexpr int AUX_HI = PARK_RAW ? 2 : 0;
    constexpr double scale = (double)RW::SCALE;

    raw_t rlo[E], rhi[E];
#pragma unroll
    for (int m = 0; m < E; ++m) rlo[m] = RW::template load<AUX>(src, voff, m * T * BPS);
#pragma unroll
    for (int m = 0; m < E; ++m) rhi[m] = RW::template load<AUX_HI>(src, voff, (m + E) * T * BPS);

    __syncthreads();  // LDS tables visible

    for (uint32_t line = 0; line < lines_wg; ++line) {
        const int next_off = (int)((line + 1) * line_bytes);
        v2d v[E], dd[PARK_RAW ? 1 : E];
        (void)dd;
        auto decode = [&](int m, v2d &lo, v2d &hi) {
            lo = RW::template dec<v2d>(BE ? RW::swap(rlo[m]) : rlo[m]);  // SMH:87-91 byte order
            hi = RW::template dec<v2d>(BE ? RW::swap(rhi[m]) : rhi[m]);
            if constexpr (HAS_WIN) {
                // Hann: w[n] = 1/2 - 1/2 cos(2 pi n / N), cos(2 pi n / N) = Re(W_N^t W_32^m) for n = t + 512 m; n + H turns
                // the cosine's sign.  (`win` only says that a window is wanted.)
                const v2d cs = wtab[m];
                const double c = __builtin_fma(wt.x, cs.x, -(wt.y * cs.y));
                const double w0 = __builtin_fma(-0.5, c, 0.5), w1 = __builtin_fma(0.5, c, 0.5);
                lo *= v2d{w0, w0};
                hi *= v2d{w1, w1};
            }
        };
        // ---- first radix-2 step, even half: a = lo + hi ----
#pragma unroll
        for (int m = 0; m < E; ++m) {
            v2d lo, hi;
            decode(m, lo, hi);
            v[m] = lo + hi;
            if constexpr (!PARK_RAW) {
                dd[m] = lo - hi;
                asm volatile("" : "+v"(dd[m]));  // computed here, not sunk behind the first transform (spec_k_v2h.hip)
            }
        }
        v3h_fft(v, t, lds, tab, tab_lo);
        double de[E];
        v3h_epilogue<E>(v, scale, db, dbt, de);

        // ---- odd half: b = (lo - hi) W_N^(t + 512 m) = d W_32^m W_N^t ----
        v3h_for_each([&](auto mt) {
            constexpr int m = decltype(mt)::value;
            if constexpr (PARK_RAW) {
                v2d lo, hi;
                asm volatile("" : "+v"(rlo[m]), "+v"(rhi[m]));  // decoded a second time from the parked raw registers
                decode(m, lo, hi);
                v[m] = v3h_twiddle<m>(lo - hi, wt);
            } else {
                v[m] = v3h_twiddle<m>(dd[m], wt);
            }
        }, std::make_integer_sequence<int, E>{});
        if constexpr (REUSE) {
#pragma unroll
            for (int m = 0; m < E; ++m) rlo[m] = rhi[m];
        }
#pragma unroll
        for (int m = 0; m < EARLY_LO; ++m) rlo[m] = RW::template load<AUX>(src, voff, next_off + m * T * BPS);
#pragma unroll
        for (int m = 0; m < EARLY_HI; ++m) rhi[m] = RW::template load<AUX_HI>(src, voff, next_off + (m + E) * T * BPS);
        v3h_fft(v, t, lds, tab, tab_lo);
        double dq[E];
        v3h_epilogue<E>(v, scale, db, dbt, dq);
        if constexpr (!REUSE) {  // (the spectrum's registers are free again) the rest of the next line
#pragma unroll
            for (int m = EARLY_LO; m < E; ++m) rlo[m] = RW::template load<AUX>(src, voff, next_off + m * T * BPS);
        }
#pragma unroll
        for (int m = EARLY_HI; m < E; ++m) rhi[m] = RW::template load<AUX_HI>(src, voff, next_off + (m + E) * T * BPS);
        // ---- bins 2k, 2k + 1 (k = t + 512 m) at columns 2c, 2c + 1, c = (k + H/2) mod H   (SS:78) ----
        const int out_off = (int)(line * (uint32_t)N * esz);
        if (out64) {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const unsigned long long b0 = (unsigned long long)__double_as_longlong(de[m]);
                const unsigned long long b1 = (unsigned long long)__double_as_longlong(dq[m]);
                v3h_store_b128<ST_AUX>(u32x4{(uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32)}, dst, ovoff,
                                       out_off + ((m + E / 2) & (E - 1)) * T * 16);
            }
        } else {
#pragma unroll
            for (int m = 0; m < E; ++m)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint((float)de[m]), __float_as_uint((float)dq[m])}, dst, ovoff,
                                                      out_off + ((m + E / 2) & (E - 1)) * T * 8, ST_AUX);
        }
    }
}

template <int KIND, bool HAS_WIN, bool BE, bool REUSE> hipError_t v3h_launch1(const V3hArgs &a, hipStream_t s) {
    constexpr size_t lds = p2_lds_bytes<113, 16>() + DB20_TAB_DOUBLES * sizeof(double) + (16 + 15 * 32) * sizeof(v2d);
    static_assert(lds <= 160 * 1024, "one workgroup's LDS");
    auto kern = v3h_kernel<KIND, HAS_WIN, BE, REUSE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((a.n_lines + a.run - 1) / a.run), dim3(Plan2<113>::T), lds, s, a);
    return hipGetLastError();
}

template <int KIND, bool BE> hipError_t v3h_launch_kind(const V3hArgs &a, hipStream_t s) {
    const bool win = a.win != nullptr;
    if constexpr (KIND != K_CF64) {
        if (a.hop == (uint32_t)Plan2<113>::N) return win ? v3h_launch1<KIND, true, BE, true>(a, s) : v3h_launch1<KIND, false, BE, true>(a, s);
    }
    return win ? v3h_launch1<KIND, true, BE, false>(a, s) : v3h_launch1<KIND, false, BE, false>(a, s);
}


// ================= 32768-point fp64 lines: a PAIR of workgroups per line (round 5) =======================================
// The fp64 twin of spec_k_v2q.hip.  A 32768-point line of fp64 complex values is 512 KiB; a radix-4 step in registers on the
// way in (Q = N/4 = 8192, x_q = x[n + q Q]; formulas in spec_k_v2q.hip) leaves four 8192-point transforms of the fp64 family:
// the first workgroup of a pair runs y0 and y1 (bins 4k, 4k + 1: one 16- or 8-byte store per register), the second y2 and y3.
// Workgroups b and b + 8 of the grid (the same XCD) form a pair, the sixteen pairs of an XCD walk one block of lines
// together; nothing waits for anything.  Until round 5 these lines took the four-step team kernel.
//   * thread t owns n = t + 512 m (m < 16) of all four quarters: W_N^(p n) = W_N^(p t) W_64^(p m);
//   * registers: the second transform's input is parked (64); the next line's 64 samples per thread are requested in the
//     order they are consumed, in instalments (cf64: 16 registers per m -- four fit beside the second epilogue, ten at the
//     top of the next line).
struct V3qArgs {
    const uint8_t *iq;
    uint32_t n_lines, hop, run, ilv;  // as V2qArgs (spec_k_v2q.hip)
    const void *tw_q;      // v2d W_8192^m
    const void *tw_full;   // v2d W_32768^m
    const void *win;
    void *out;
    int out_fmt;
};

// a * W_64^J for a compile-time J (kW64 holds the upper half circle)
template <int J> __device__ __forceinline__ v2d v3q_mul_w64(v2d a) {
    constexpr int K = J & 63;
    if constexpr (K == 0) return a;
    else if constexpr (K == 16) return pk_mul_mi(a);
    else if constexpr (K == 32) return v2d{-a.x, -a.y};
    else if constexpr (K == 48) return pk_mul_mi(v2d{-a.x, -a.y});
    else if constexpr (K < 32) return pk_cmul_const(a, kW64[K][0], kW64[K][1]);
    else return pk_cmul_const(a, -kW64[K - 32][0], -kW64[K - 32][1]);
}

#ifndef V3Q_M0
#define V3Q_M0 -1  // request schedule overrides (experiments)
#define V3Q_M1 -1
#define V3Q_M2 -1
#endif

template <int KIND, bool HAS_WIN, bool BE, int HALF>
__device__ __forceinline__ void v3q_body(const V3qArgs &a, uint32_t line0, uint32_t lines_wg, unsigned char *smem) {
    using PL = Plan2<113>;
    using RW = Raw2<KIND>;
    using raw_t = typename RW::type;
    constexpr int BPS = RW::BPS, Q = PL::N, N = 4 * Q, T = PL::T, E = PL::E;
    static_assert(E == 16 && T == 512 && N / T == 64, "n = t + 512 m, W_N^(512 m) = W_64^m");
    constexpr int REG = sizeof(raw_t) <= 4 ? 1 : (int)sizeof(raw_t) / 4;  // registers per raw sample
    // m (four samples each) of the next line in flight: behind the second transform's input / behind its epilogue / at the top
    // of the next line; the rest when the first SPLIT have been combined
    constexpr int M0 = V3Q_M0 >= 0 ? V3Q_M0 : (REG == 4 ? 1 : REG == 2 ? 3 : 8), M1 = V3Q_M1 >= 0 ? V3Q_M1 : (REG == 4 ? 5 : REG == 2 ? 10 : 16),
                  M2 = V3Q_M2 >= 0 ? V3Q_M2 : (REG == 4 ? 9 : REG == 2 ? 14 : 16), SPLIT = 4;
    static_assert(M0 <= M1 && M1 <= M2 && M2 <= E, "request schedule");
    const int t = threadIdx.x;
    v2d *lds = reinterpret_cast<v2d *>(smem);
    v2d *tab = reinterpret_cast<v2d *>(smem + (size_t)PL::LINE * 16);
    const v2d *__restrict__ tw = static_cast<const v2d *>(a.tw_q);
    const v2d *__restrict__ twf = static_cast<const v2d *>(a.tw_full);

    fill_tables<113, 1>(tab, tw, t);
    double *dbt_w = reinterpret_cast<double *>(smem + p2_lds_bytes<113, 16>());
    for (int e = t; e < DB20_TAB_DOUBLES; e += T) dbt_w[e] = DB20_TAB[e];
    const double *dbt = dbt_w;
    v2d *wtab = reinterpret_cast<v2d *>(dbt_w + DB20_TAB_DOUBLES);  // W_64^m = W_N^(512 m), m < 16 (the Hann window)
    if (HAS_WIN && t < 16) wtab[t] = twf[T * t];
    v2d *tab_lo = wtab + 16;  // W_Q^(r j), r = 1 .. 15, j < 32
    if (t < 15 * 32) tab_lo[t] = tw[(t / 32 + 1) * (t % 32)];
    constexpr int PA = 2 * HALF, PB = 2 * HALF + 1;  // y_PA through the first transform, y_PB through the second
    const v2d wA = twf[(PA * t) & (N - 1)], wB = twf[(PB * t) & (N - 1)];
    v2d wt = v2d{1.0, 0.0};
    if constexpr (HAS_WIN) wt = twf[t];
    const bool out64 = a.out_fmt == OUT_DB20_F64 || a.out_fmt == OUT_POW_F64;
    const bool db = a.out_fmt == OUT_DB20_F64 || a.out_fmt == OUT_DB20_F32;
    const uint32_t esz = out64 ? 8u : 4u;

    const uint32_t line_bytes = a.hop * BPS * a.ilv, line_out = (uint32_t)N * a.ilv;  // from one of the pair's lines to its next
    uint32_t lw = lines_wg;
    asm volatile("" : "+s"(lw));
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.iq) + (uint64_t)line0 * a.hop * BPS, 0, (lw - 1) * line_bytes + (uint32_t)N * BPS, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<uint8_t *>(a.out) + (uint64_t)line0 * N * esz, 0, ((lw - 1) * line_out + (uint32_t)N) * esz, 0x00020000);
    const int voff = t * BPS, ovoff = (t * 4 + HALF * 2) * (int)esz;
    constexpr int AUX = 0, ST_AUX = 0;  // default policy: every byte is read by two workgroups, every output line written by two
    constexpr double scale = (double)RW::SCALE;

    raw_t r[4][E];
    auto request = [&](int off, auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
        r[0][m] = RW::template load<AUX>(src, voff, off + (m * T) * BPS);
        r[2][m] = RW::template load<AUX>(src, voff, off + (m * T + 2 * Q) * BPS);
        r[1][m] = RW::template load<AUX>(src, voff, off + (m * T + Q) * BPS);
        r[3][m] = RW::template load<AUX>(src, voff, off + (m * T + 3 * Q) * BPS);
    };
    auto request_range = [&](int off, auto lo_tag, auto hi_tag) {
        constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
        v3h_for_each([&](auto mt) { request(off, std::integral_constant<int, LO + decltype(mt)::value>{}); },
                     std::make_integer_sequence<int, HI - LO>{});
    };
    using I0 = std::integral_constant<int, 0>;
    using IM0 = std::integral_constant<int, M0>;
    using IM1 = std::integral_constant<int, M1>;
    using IM2 = std::integral_constant<int, M2>;
    using IE = std::integral_constant<int, E>;
    request_range(0, I0{}, IM2{});

    __syncthreads();  // LDS tables visible

    for (uint32_t line = 0; line < lines_wg; ++line) {
        const int next_off = (int)((line + 1) * line_bytes);
        v2d v[E], dd[E];
        v3h_for_each([&](auto m_tag) {
            constexpr int m = decltype(m_tag)::value;
            if constexpr (m == SPLIT && M2 < E) request_range((int)(line * line_bytes), IM2{}, IE{});
            v2d x0 = RW::template dec<v2d>(BE ? RW::swap(r[0][m]) : r[0][m]);  // SMH:87-91 byte order
            v2d x2 = RW::template dec<v2d>(BE ? RW::swap(r[2][m]) : r[2][m]);
            v2d x1 = RW::template dec<v2d>(BE ? RW::swap(r[1][m]) : r[1][m]);
            v2d x3 = RW::template dec<v2d>(BE ? RW::swap(r[3][m]) : r[3][m]);
            if constexpr (HAS_WIN) {  // Hann: the four quarters see cos, -sin, -cos, sin of 2 pi n / N (spec_k_v2q.hip)
                const v2d cs = wtab[m];
                const double c = __builtin_fma(wt.x, cs.x, -(wt.y * cs.y)), ms = __builtin_fma(wt.x, cs.y, wt.y * cs.x);
                const double w0 = __builtin_fma(-0.5, c, 0.5), w2 = __builtin_fma(0.5, c, 0.5);
                const double w1 = __builtin_fma(-0.5, ms, 0.5), w3 = __builtin_fma(0.5, ms, 0.5);
                x0 *= v2d{w0, w0}; x1 *= v2d{w1, w1}; x2 *= v2d{w2, w2}; x3 *= v2d{w3, w3};
            }
            const v2d ls = x0 + x2, ld = x0 - x2, hs = x1 + x3, hd = x1 - x3;
            if constexpr (HALF == 0) {
                v[m] = ls + hs;              // y0
                dd[m] = pk_add_mi(ld, hd);   // y1 before its twiddle: ld - i hd
            } else {
                v[m] = pk_cmul(v3q_mul_w64<PA * m>(ls - hs), wA);  // y2
                dd[m] = pk_sub_mi(ld, hd);                          // y3 before its twiddle: ld + i hd
            }
            asm volatile("" : "+v"(dd[m]));  // computed here (spec_k_v2h.hip)
        }, std::make_integer_sequence<int, E>{});

        v3h_fft(v, t, lds, tab, tab_lo);
        double de[E];
        v3h_epilogue<E>(v, scale, db, dbt, de);

        v3h_for_each([&](auto mt) {
            constexpr int m = decltype(mt)::value;
            v[m] = pk_cmul(v3q_mul_w64<PB * m>(dd[m]), wB);
        }, std::make_integer_sequence<int, E>{});
        request_range(next_off, I0{}, IM0{});
        v3h_fft(v, t, lds, tab, tab_lo);
        double dq[E];
        v3h_epilogue<E>(v, scale, db, dbt, dq);
        request_range(next_off, IM0{}, IM1{});
        // ---- bins 4k + 2h, 4k + 2h + 1 (k = t + 512 m) at columns 4c + 2h, 4c + 2h + 1, c = (k + Q/2) mod Q   (SS:78) ----
        const int out_off = (int)(line * line_out * esz);
        if (out64) {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const unsigned long long b0 = (unsigned long long)__double_as_longlong(de[m]);
                const unsigned long long b1 = (unsigned long long)__double_as_longlong(dq[m]);
                v3h_store_b128<ST_AUX>(u32x4{(uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32)}, dst, ovoff,
                                       out_off + ((m + E / 2) & (E - 1)) * T * 32);
            }
        } else {
#pragma unroll
            for (int m = 0; m < E; ++m)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint((float)de[m]), __float_as_uint((float)dq[m])}, dst, ovoff,
                                                      out_off + ((m + E / 2) & (E - 1)) * T * 16, ST_AUX);
        }
        __builtin_amdgcn_sched_barrier(0);  // (the last instalment really behind the stores: spec_k_v2q.hip)
        if constexpr (M1 < M2) request_range(next_off, IM1{}, IM2{});
    }
}

template <int KIND, bool HAS_WIN, bool BE>
__global__ __launch_bounds__(Plan2<113>::T, 2) void v3q_kernel(const V3qArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the pair: workgroups 16 g + i (first) and 16 g + 8 + i (second), both on XCD i under the round-robin dispatch
    const uint32_t half = (blockIdx.x >> 3) & 1u, g = blockIdx.x >> 4, i = blockIdx.x & 7u;
    uint32_t line0, lines_wg;
    if (a.ilv == 1) {
        line0 = (g * 8u + i) * a.run;
        if (line0 >= a.n_lines) return;
        lines_wg = a.n_lines - line0 < a.run ? a.n_lines - line0 : a.run;
    } else {
        const uint32_t slot = g & 15u, block = (g >> 4) * 8u + i;
        line0 = block * 16u * a.run + slot;
        if (line0 >= a.n_lines) return;
        lines_wg = (a.n_lines - line0 + 15u) / 16u;
        if (lines_wg > a.run) lines_wg = a.run;
    }
    if (half == 0) v3q_body<KIND, HAS_WIN, BE, 0>(a, line0, lines_wg, smem);
    else v3q_body<KIND, HAS_WIN, BE, 1>(a, line0, lines_wg, smem);
}

template <int KIND, bool HAS_WIN, bool BE> hipError_t v3q_launch1(const V3qArgs &a, hipStream_t s) {
    constexpr size_t lds = p2_lds_bytes<113, 16>() + DB20_TAB_DOUBLES * sizeof(double) + (16 + 15 * 32) * sizeof(v2d);
    static_assert(lds <= 160 * 1024, "one workgroup's LDS");
    auto kern = v3q_kernel<KIND, HAS_WIN, BE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint32_t grid;
    if (a.ilv == 1) grid = ((a.n_lines + a.run - 1) / a.run + 7) / 8 * 16;
    else grid = ((a.n_lines + 16 * a.run - 1) / (16 * a.run) + 7) / 8 * 16 * 16;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Plan2<113>::T), lds, s, a);
    return hipGetLastError();
}

template <int KIND, bool BE> hipError_t v3q_launch_kind(const V3qArgs &a, hipStream_t s) {
    return a.win ? v3q_launch1<KIND, true, BE>(a, s) : v3q_launch1<KIND, false, BE>(a, s);
}

}  // namespace

bool v3h_applicable(int log2n, int kind, uint64_t n_lines, uint32_t hop) {
    if (log2n != 14) return false;
    if (kind != K_CF64 && kind != K_CF32 && kind != K_CI16 && kind != K_CU8 && kind != K_CI8) return false;
    return n_lines > 0 && n_lines < (1ull << 31) && hop <= (8u << log2n);
}

hipError_t launch_v3h_spectro(const WfArgs &w, const void *tw_half, uint32_t run, hipStream_t s) {
#ifdef V3Q_ONLY_KIND
    return hipErrorInvalidValue;
#endif
    V3hArgs a{};
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.run = run;
    a.tw_half = tw_half; a.tw_full = w.tw; a.win = w.win; a.out = w.out; a.out_fmt = w.out_fmt;
    switch (w.kind) {
    case K_CF64: return w.be ? v3h_launch_kind<K_CF64, true>(a, s) : v3h_launch_kind<K_CF64, false>(a, s);
    case K_CF32: return w.be ? v3h_launch_kind<K_CF32, true>(a, s) : v3h_launch_kind<K_CF32, false>(a, s);
    case K_CI16: return w.be ? v3h_launch_kind<K_CI16, true>(a, s) : v3h_launch_kind<K_CI16, false>(a, s);
    case K_CU8: return v3h_launch_kind<K_CU8, false>(a, s);
    case K_CI8: return v3h_launch_kind<K_CI8, false>(a, s);
    default: return hipErrorInvalidValue;
    }
}

bool v3q_applicable(int log2n, int kind, uint64_t n_lines, uint32_t hop) {
    if (log2n != 15) return false;
    if (kind != K_CF64 && kind != K_CF32 && kind != K_CI16 && kind != K_CU8 && kind != K_CI8) return false;
    return n_lines > 0 && n_lines < (1ull << 31) && hop <= (8u << log2n);
}

hipError_t launch_v3q_spectro(const WfArgs &w, const void *tw_q, uint32_t run, int interleave, hipStream_t s) {
    V3qArgs a{};
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.run = run; a.ilv = interleave ? 16u : 1u;
    a.tw_q = tw_q; a.tw_full = w.tw; a.win = w.win; a.out = w.out; a.out_fmt = w.out_fmt;
#ifdef V3Q_ONLY_KIND  // development: one instantiation per compile (register experiments)
    return v3q_launch1<V3Q_ONLY_KIND, V3Q_ONLY_WIN, V3Q_ONLY_BE>(a, s);
#endif
    switch (w.kind) {
    case K_CF64: return w.be ? v3q_launch_kind<K_CF64, true>(a, s) : v3q_launch_kind<K_CF64, false>(a, s);
    case K_CF32: return w.be ? v3q_launch_kind<K_CF32, true>(a, s) : v3q_launch_kind<K_CF32, false>(a, s);
    case K_CI16: return w.be ? v3q_launch_kind<K_CI16, true>(a, s) : v3q_launch_kind<K_CI16, false>(a, s);
    case K_CU8: return v3q_launch_kind<K_CU8, false>(a, s);
    case K_CI8: return v3q_launch_kind<K_CI8, false>(a, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace specgpu
