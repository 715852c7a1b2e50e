// spec_k_v2h.hip -- 32768-point fp32 lines in ONE workgroup (round 4).  The top of the reference's NFFT slider
// (main-scene.fxml:129-132: 2^6 ... 2^16) around SpectralService.java:33-85.
//
// A 32768-point cf32 line is 256 KiB: it does not fit the LDS (160 KiB), and until round 3 it went through the
// four-step team kernel (spec_k_team.hip: sixteen + sixteen workgroups per line, the intermediate handed over in L2,
// 0.26 of the HBM roofline -- bound by the hand-off chain).  It does fit ONE workgroup's registers and
// LDS if the first radix-2 step is taken in registers on the way in (decimation in frequency):
//     a[n] = x[n] + x[n + H],   b[n] = (x[n] - x[n + H]) W_N^n,   H = N/2 = 16384,  n < H
//     X[2k] = FFT_H(a)[k],      X[2k + 1] = FFT_H(b)[k]
// and the two 16384-point transforms run one after the other through the SAME LDS buffer with the family's own
// passes (Plan2<14>: 512 threads, 32 points each, radix 32 x 32 x 16, two exchanges).  No other workgroup is waited
// for, nothing but the line's samples and bins crosses the CU's boundary.
//   * thread t owns n = t + 512 m (m < 32) of both halves, so W_N^n = W_N^t W_64^m: one per-thread twiddle and 32
//     compile-time constants;
//   * the even bins wait as 32 finished floats per thread while the odd half is transformed, then leave in PAIRS
//     (X[2k], X[2k+1]: 8-byte stores, 512 bytes per wave and instruction), fftshift (SS:78) folded into the index;
//   * register budget (256 at two waves per SIMD): transform state 64 + 64, plus what is parked.  cf32: the difference
//     d = lo - hi (64 registers) during the first transform; the next line's lower half is requested at the start of the
//     first transform, the upper half in two pieces around the second -- no room to keep the 50 %-overlap half in
//     registers, it comes back from L2 / the Infinity Cache.  2- and 4-byte formats: the raw samples themselves are
//     parked (32 + 32 registers, decoded twice) and at hop = N/2 the upper half STAYS as the next line's lower half.
#include "spec_v2h.h"

namespace specgpu {

namespace {

struct V2hArgs {
    const uint8_t *iq;     // first byte of line 0
    uint32_t n_lines, hop, run;  // run: consecutive lines per workgroup
    const void *tw_half;   // v2f W_16384^m
    const void *tw_full;   // v2f W_32768^m
    const void *tw_full64; // v2d W_N^m: the Hann window's cosine is formed in fp64 (non-null with `win`)
    const void *win;       // non-null: Hann window (computed from the twiddles, the table is not read)
    float *out;
    int out_fmt;
};

#ifndef V2H_PF
#define V2H_PF 16  // cf32: samples of the next line's lower half requested at the start of the second transform
#endif

// REUSE: hop == N/2 and a 2- / 4-byte format -- the raw upper half stays in registers as the next line's lower half
// L: log2 of the HALF line (14: 32768-point lines, one 512-thread workgroup per CU; 13: 16384-point lines, 256-thread
// workgroups, two per CU -- two independent barrier domains on every SIMD; 12: 8192-point lines, 16 points per thread and
// half, 256-thread workgroups at V2H_WAVES_12 waves per SIMD)
#ifndef V2H_WAVES_12
#define V2H_WAVES_12 3
#endif
template <int L, int KIND, bool HAS_WIN, bool BE, bool REUSE>
__global__ __launch_bounds__(Plan2<L>::T, L == 12 ? V2H_WAVES_12 : 2) void v2h_kernel(const V2hArgs a) {
    using PL = Plan2<L>;
    using RW = Raw2<KIND>;
    using raw_t = typename RW::type;
    constexpr int BPS = RW::BPS, H = PL::N, N = 2 * H, T = PL::T, E = PL::E;
    constexpr bool PARK_RAW = KIND != K_CF32;  // park the raw samples (E + E registers) instead of the decoded difference
    constexpr int PF = V2H_PF < E ? V2H_PF : E;
    constexpr int NT = N / T;
    static_assert(!REUSE || PARK_RAW, "register reuse needs the raw halves parked");
    static_assert((E == 32 && NT == 64) || (E == 16 && NT == 32), "n = t + T m, W_N^(T m) = W_NT^m, m < NT / 2");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    v2f *lds = reinterpret_cast<v2f *>(smem);
    v2f *tab = reinterpret_cast<v2f *>(smem + (size_t)PL::LINE * 8);
    const v2f *__restrict__ tw = static_cast<const v2f *>(a.tw_half);

    fill_tables<L, 1>(tab, tw, t);
    v2d *wtab = reinterpret_cast<v2d *>(tab + p2_tab_entries<L>());  // W_NT^m = W_N^(T m), m < E, in fp64 (the Hann window's cosine, below)
    v2d *wt64s = wtab + 32;  // W_N^t of every thread, in LDS: four registers only while a line is decoded
    if constexpr (HAS_WIN) {
        if (t < E) wtab[t] = static_cast<const v2d *>(a.tw_full64)[T * t];
        wt64s[t] = static_cast<const v2d *>(a.tw_full64)[t];
    }
    v2f twl[16];
#pragma unroll
    for (int r = 1; r < 16; ++r) twl[r] = tw[(r * t) & (H - 1)];
    const v2f wt = static_cast<const v2f *>(a.tw_full)[t];  // W_N^t
    const bool db = a.out_fmt == OUT_DB20_F32;

    const uint32_t line0 = blockIdx.x * a.run;
    uint32_t lines_wg = a.n_lines - line0;
    if (lines_wg > a.run) lines_wg = a.run;
    const uint32_t line_bytes = a.hop * BPS;
    // (lw: the same number behind an empty asm.  Left visible, hipcc merges "lines_wg - 1" with the loop guard
    // "lines_wg == 0" into ONE v_sub_co_u32 -- a VALU instruction -- the descriptor's size becomes a vector register and
    // every buffer access of the kernel a readfirstlane loop.)
    uint32_t lw = lines_wg;
    asm volatile("" : "+s"(lw));
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.iq) + (uint64_t)line0 * line_bytes, 0, (lw - 1) * line_bytes + (uint32_t)N * BPS, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(a.out + (uint64_t)line0 * N, 0, lw * (uint32_t)N * 4u, 0x00020000);
    const int voff = t * BPS, ovoff = t * 8;
    constexpr int AUX = 2, ST_AUX = 2;  // non-temporal, as the family
    // cf32 without register reuse: the overlapped half is read a second time one line later -- cached (L2 / MALL) on
    // its first reading, non-temporal on its last
    constexpr int AUX_HI = PARK_RAW ? 2 : 0;

    raw_t rlo[E], rhi[E];
#pragma unroll
    for (int m = 0; m < E; ++m) rlo[m] = RW::template load<AUX>(src, voff, m * T * BPS);
#pragma unroll
    for (int m = 0; m < E; ++m) rhi[m] = RW::template load<AUX_HI>(src, voff, (m + E) * T * BPS);

    __syncthreads();  // LDS twiddle table visible

    for (uint32_t line = 0; line < lines_wg; ++line) {
        const int next_off = (int)((line + 1) * line_bytes);
        v2f v[E], dd[PARK_RAW ? 1 : E];
        (void)dd;
        auto decode = [&](int m, v2f &lo, v2f &hi, v2d wt64) {
            lo = RW::dec(BE ? RW::swap(rlo[m]) : rlo[m]);  // SMH:87-91 byte order
            hi = RW::dec(BE ? RW::swap(rhi[m]) : rhi[m]);
            if constexpr (HAS_WIN) {
                // Hann (the only window of the ABI): w[n] = 1/2 - 1/2 cos(2 pi n / N), and cos(2 pi n / N) = Re W_N^n =
                // Re(W_N^t W_64^m) for n = t + 512 m -- from the twiddle the thread holds and W_64^m (a 32-entry LDS table), where the
                // table cost two loads per sample from L2 for each half (0.32 of 8 TB/s against 0.43 without a window);
                // n + H turns the cosine's sign.  (`win` only says that a window is wanted.)
                // The cosine in fp64 (round 4, after the extended random runs): formed from the fp32 twiddles its error, ~1.5e-7 and
                // the same for all of a thread's samples, reached the tolerance of the 1e-4 M tier (5.4e-3 dB seen at 16384 points
                // against 3.0e-3 with the table, tools/err_window.py); now the window is the table's value to fp32 rounding.
                const v2d cs = wtab[m];  // one broadcast LDS read
                const double c = __builtin_fma(wt64.x, cs.x, -(wt64.y * cs.y));
                const float w0 = (float)__builtin_fma(-0.5, c, 0.5), w1 = (float)__builtin_fma(0.5, c, 0.5);
                lo *= v2f{w0, w0};
                hi *= v2f{w1, w1};
            }
        };
        // ---- first radix-2 step, even half: a = lo + hi ----
        v2d wt64 = v2d{1.0, 0.0};
        if constexpr (HAS_WIN) wt64 = wt64s[t];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            v2f lo, hi;
            decode(m, lo, hi, wt64);
            v[m] = lo + hi;
            if constexpr (!PARK_RAW) {
                dd[m] = lo - hi;
                // computed HERE: left alone, hipcc sinks the subtraction to its use behind the first transform and keeps
                // lo and hi -- 128 registers instead of 64 -- alive across it
                asm volatile("" : "+v"(dd[m]));
            }
        }
        v2_fft<L>(v, t, lds, tab, twl);
        float de[E];
        constexpr bool BOUNDED = KIND != K_CF32;
        v2h_epilogue<BOUNDED, E>(v, RW::SCALE, db, de);

        // ---- odd half: b = (lo - hi) W_N^(t + 512 m) = d W_64^m W_N^t ----
        if constexpr (HAS_WIN && PARK_RAW) wt64 = wt64s[t];  // (read again: not kept across the transform)
        v2h_for_each([&](auto mt) {
            constexpr int m = decltype(mt)::value;
            if constexpr (PARK_RAW) {
                v2f lo, hi;
                // decoded a second time from the parked raw registers; the empty asm keeps hipcc from re-using the first
                // decode's floats instead (128 registers alive across the first transform)
                asm volatile("" : "+v"(rlo[m]), "+v"(rhi[m]));
                decode(m, lo, hi, wt64);
                v[m] = v2h_twiddle<m, NT>(lo - hi, wt);
            } else {
                v[m] = v2h_twiddle<m, NT>(dd[m], wt);
            }
        }, std::make_integer_sequence<int, E>{});
        if constexpr (PARK_RAW) {  // the whole next line is requested here and lands behind the second transform
            if constexpr (REUSE) {
#pragma unroll
                for (int m = 0; m < E; ++m) rlo[m] = rhi[m];
            } else {
#pragma unroll
                for (int m = 0; m < E; ++m) rlo[m] = RW::template load<AUX>(src, voff, next_off + m * T * BPS);
            }
#pragma unroll
            for (int m = 0; m < E; ++m) rhi[m] = RW::template load<AUX_HI>(src, voff, next_off + (m + E) * T * BPS);
        } else {  // cf32: PF samples of the next line's lower half fit beside the second transform
#pragma unroll
            for (int m = 0; m < PF; ++m) rlo[m] = RW::template load<AUX>(src, voff, next_off + m * T * BPS);
        }
        v2_fft<L>(v, t, lds, tab, twl);
        float dq[E];
        v2h_epilogue<BOUNDED, E>(v, RW::SCALE, db, dq);
        if constexpr (!PARK_RAW) {  // (the spectrum's registers are free again) the rest of the next line
#pragma unroll
            for (int m = PF; m < E; ++m) rlo[m] = RW::template load<AUX>(src, voff, next_off + m * T * BPS);
#pragma unroll
            for (int m = 0; m < E; ++m) rhi[m] = RW::template load<AUX_HI>(src, voff, next_off + (m + E) * T * BPS);
        }
        // ---- bins 2k, 2k + 1 (k = t + 512 m) at columns 2c, 2c + 1, c = (k + H/2) mod H   (SS:78) ----
        const int out_off = (int)(line * (uint32_t)N * 4u);
#pragma unroll
        for (int m = 0; m < E; ++m)
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(de[m]), __float_as_uint(dq[m])}, dst, ovoff,
                                                  out_off + ((m + E / 2) & (E - 1)) * T * 8, ST_AUX);
    }
}

template <int L, int KIND, bool HAS_WIN, bool BE, bool REUSE> hipError_t v2h_launch1(const V2hArgs &a, hipStream_t s) {
    constexpr size_t lds = p2_lds_bytes<L>() + (32 + (HAS_WIN ? Plan2<L>::T : 0)) * sizeof(v2d);  // + the window's W_NT table and per-thread W_N^t (fp64)
    static_assert(lds <= 160 * 1024, "one workgroup's LDS");
    auto kern = v2h_kernel<L, KIND, HAS_WIN, BE, REUSE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((a.n_lines + a.run - 1) / a.run), dim3(Plan2<L>::T), lds, s, a);
    return hipGetLastError();
}

template <int L, int KIND, bool BE> hipError_t v2h_launch_kind(const V2hArgs &a, hipStream_t s) {
    const bool win = a.win != nullptr;
    if constexpr (KIND != K_CF32) {
        if (a.hop == (uint32_t)Plan2<L>::N) return win ? v2h_launch1<L, KIND, true, BE, true>(a, s) : v2h_launch1<L, KIND, false, BE, true>(a, s);
    }
    return win ? v2h_launch1<L, KIND, true, BE, false>(a, s) : v2h_launch1<L, KIND, false, BE, false>(a, s);
}

template <int L> hipError_t v2h_launch_l(const V2hArgs &a, int kind, int be, hipStream_t s) {
    switch (kind) {
    case K_CF32: return be ? v2h_launch_kind<L, K_CF32, true>(a, s) : v2h_launch_kind<L, K_CF32, false>(a, s);
    case K_CI16: return be ? v2h_launch_kind<L, K_CI16, true>(a, s) : v2h_launch_kind<L, K_CI16, false>(a, s);
    case K_CU8: return v2h_launch_kind<L, K_CU8, false>(a, s);
    case K_CI8: return v2h_launch_kind<L, K_CI8, false>(a, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool v2h_applicable(int log2n, int kind, int out_fmt, uint64_t n_lines, uint32_t hop) {
    if (log2n != 15 && log2n != 14 && log2n != 13) return false;
    if (kind != K_CF32 && kind != K_CI16 && kind != K_CU8 && kind != K_CI8) return false;
    if (out_fmt != OUT_DB20_F32 && out_fmt != OUT_POW_F32) return false;
    return n_lines > 0 && n_lines < (1ull << 31) && hop <= (8u << log2n);  // a workgroup's span stays far below 4 GiB
}

hipError_t launch_v2h_spectro(const WfArgs &w, int log2n, const void *tw_half, const void *tw_full64, uint32_t run, hipStream_t s) {
    V2hArgs a{};
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.run = run;
    a.tw_half = tw_half; a.tw_full = w.tw; a.tw_full64 = tw_full64; a.win = tw_full64 ? w.win : nullptr;
    a.out = static_cast<float *>(w.out); a.out_fmt = w.out_fmt;
    return log2n == 15 ? v2h_launch_l<14>(a, w.kind, w.be, s) : log2n == 14 ? v2h_launch_l<13>(a, w.kind, w.be, s) : v2h_launch_l<12>(a, w.kind, w.be, s);
}

}  // namespace specgpu
