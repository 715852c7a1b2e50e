// spec_k_v2q.hip -- 65536-point fp32 lines, the top of the reference's NFFT slider (main-scene.fxml:129-132: 2^6 ... 2^16) around
// SpectralService.java:33-85, WITHOUT the four-step team kernel: a PAIR of workgroups per line, each of them a single-workgroup
// kernel of the spec_k_v2h.hip kind (round 5).
//
// A 65536-point cf32 line is 512 KiB; one workgroup holds a quarter of it.  Two nested radix-2 steps, decimation in frequency,
// are one radix-4 step in registers on the way in (Q = N/4 = 16384, n < Q, x_q = x[n + q Q]):
//     y0 =  (x0 + x2) +    (x1 + x3)              -> X[4k]     = FFT_Q(y0)[k]
//     y2 = ((x0 + x2) -    (x1 + x3)) W_N^(2n)    -> X[4k + 2] = FFT_Q(y2)[k]
//     y1 = ((x0 - x2) - i  (x1 - x3)) W_N^n       -> X[4k + 1] = FFT_Q(y1)[k]
//     y3 = ((x0 - x2) + i  (x1 - x3)) W_N^(3n)    -> X[4k + 3] = FFT_Q(y3)[k]
// The FIRST workgroup of a pair forms y0 and y1 from the whole line and runs two 16384-point transforms of the packed family
// through its LDS buffer (Plan2<14>: 512 threads x 32 points, as spec_k_v2h.hip); the SECOND does the same with y2 and y3.
// (y0 with y1, not with y2: the two bins a thread finishes per k are then neighbours, 4k + 2h and 4k + 2h + 1, and leave as ONE
// 8-byte store; the first build paired y0 with y2 -- two 4-byte stores 8 bytes apart, twice the store instructions.)
// Both read the same 512 KiB at about the same time and write interleaved halves of the same 256 KiB of output.  They are
// workgroups b and b + 8 of the grid -- the same XCD under the round-robin dispatch -- so the second reader of a byte finds it in
// that XCD's L2 and the two halves of an output cache line meet there before they leave.  NOTHING waits for anything: no
// counters, no residency assumption; another placement changes the L2 hit rate and the write merging, never the result.
//   * thread t owns n = t + 512 m (m < 32) of all four quarters: W_N^(p n) = W_N^(p t) W_128^(p m) -- one per-thread twiddle per
//     transform and 32 compile-time constants (p = 0 / 1 in the first workgroup, 2 / 3 in the second);
//   * registers: the second transform's input waits as 32 complex values (64 registers) behind the first; the next line's
//     samples are requested in the order they are consumed -- a few behind the second transform's input, most of them behind
//     its epilogue, the rest while the first are being combined -- because the whole line (128 samples per thread) does not fit;
//     no register reuse of the 50 % overlap: what a line shares with the previous one comes back from L2 / the Infinity Cache;
//   * bins 4k + 2h and 4k + 2h + 1 (h = which workgroup of the pair) leave as one 8-byte store, fftshift (SS:78) folded into the index.
#include "spec_v2h.h"

namespace specgpu {

namespace {

struct V2qArgs {
    const uint8_t *iq;            // first byte of line 0
    uint32_t n_lines, hop, run;   // run: lines per workgroup PAIR
    uint32_t ilv;                 // 16: the sixteen pairs of an XCD share a block of 16 * run lines, pair `slot` taking lines slot, slot + 16, ...
                                  // (they then work on sixteen CONSECUTIVE lines at a time: the half a line shares with the next is
                                  // being read by the neighbouring pair and comes from that XCD's L2); 1: `run` consecutive lines per pair
    const void *tw_q;             // v2f W_16384^m: the transforms
    const void *tw_full;          // v2f W_65536^m
    const void *tw_full64;        // v2d W_65536^m: the Hann window's cosine is formed in fp64 (non-null with `win`)
    const void *win;              // non-null: Hann window (computed from the twiddles, the table is not read)
    float *out;
    int out_fmt;
};

// W_128^j = exp(-2 pi i j / 128), j = 0 .. 127 (40-digit arithmetic, rounded once by the compiler)
__device__ static constexpr double kW128[128][2] = {
    {1.0, 0.0}, {0.998795456205172392714772, -0.049067674327418014254955},
    {0.995184726672196886244837, -0.0980171403295606019941956}, {0.989176509964780973451674, -0.14673047445536175165885},
    {0.980785280403230449126182, -0.195090322016128267848285}, {0.970031253194543992603984, -0.242980179903263889948274},
    {0.956940335732208864935798, -0.290284677254462367636192}, {0.941544065183020778412509, -0.336889853392220050689253},
    {0.923879532511286756128183, -0.38268343236508977172846}, {0.9039892931234433315862, -0.427555093430282094320967},
    {0.881921264348355029712757, -0.471396736825997648556388}, {0.85772861000027206990227, -0.514102744193221726593694},
    {0.831469612302545237078788, -0.555570233019602224742831}, {0.803207531480644909806677, -0.595699304492433343467037},
    {0.773010453362736960810907, -0.634393284163645498215172}, {0.740951125354959091175617, -0.671558954847018400625377},
    {0.707106781186547524400844, -0.707106781186547524400844}, {0.671558954847018400625377, -0.740951125354959091175617},
    {0.634393284163645498215172, -0.773010453362736960810907}, {0.595699304492433343467037, -0.803207531480644909806677},
    {0.555570233019602224742831, -0.831469612302545237078788}, {0.514102744193221726593694, -0.85772861000027206990227},
    {0.471396736825997648556388, -0.881921264348355029712757}, {0.427555093430282094320967, -0.9039892931234433315862},
    {0.38268343236508977172846, -0.923879532511286756128183}, {0.336889853392220050689253, -0.941544065183020778412509},
    {0.290284677254462367636192, -0.956940335732208864935798}, {0.242980179903263889948274, -0.970031253194543992603984},
    {0.195090322016128267848285, -0.980785280403230449126182}, {0.14673047445536175165885, -0.989176509964780973451674},
    {0.0980171403295606019941956, -0.995184726672196886244837}, {0.049067674327418014254955, -0.998795456205172392714772},
    {0.0, -1.0}, {-0.049067674327418014254955, -0.998795456205172392714772},
    {-0.0980171403295606019941956, -0.995184726672196886244837}, {-0.14673047445536175165885, -0.989176509964780973451674},
    {-0.195090322016128267848285, -0.980785280403230449126182}, {-0.242980179903263889948274, -0.970031253194543992603984},
    {-0.290284677254462367636192, -0.956940335732208864935798}, {-0.336889853392220050689253, -0.941544065183020778412509},
    {-0.38268343236508977172846, -0.923879532511286756128183}, {-0.427555093430282094320967, -0.9039892931234433315862},
    {-0.471396736825997648556388, -0.881921264348355029712757}, {-0.514102744193221726593694, -0.85772861000027206990227},
    {-0.555570233019602224742831, -0.831469612302545237078788}, {-0.595699304492433343467037, -0.803207531480644909806677},
    {-0.634393284163645498215172, -0.773010453362736960810907}, {-0.671558954847018400625377, -0.740951125354959091175617},
    {-0.707106781186547524400844, -0.707106781186547524400844}, {-0.740951125354959091175617, -0.671558954847018400625377},
    {-0.773010453362736960810907, -0.634393284163645498215172}, {-0.803207531480644909806677, -0.595699304492433343467037},
    {-0.831469612302545237078788, -0.555570233019602224742831}, {-0.85772861000027206990227, -0.514102744193221726593694},
    {-0.881921264348355029712757, -0.471396736825997648556388}, {-0.9039892931234433315862, -0.427555093430282094320967},
    {-0.923879532511286756128183, -0.38268343236508977172846}, {-0.941544065183020778412509, -0.336889853392220050689253},
    {-0.956940335732208864935798, -0.290284677254462367636192}, {-0.970031253194543992603984, -0.242980179903263889948274},
    {-0.980785280403230449126182, -0.195090322016128267848285}, {-0.989176509964780973451674, -0.14673047445536175165885},
    {-0.995184726672196886244837, -0.0980171403295606019941956}, {-0.998795456205172392714772, -0.049067674327418014254955},
    {-1.0, 0.0}, {-0.998795456205172392714772, 0.049067674327418014254955},
    {-0.995184726672196886244837, 0.0980171403295606019941956}, {-0.989176509964780973451674, 0.14673047445536175165885},
    {-0.980785280403230449126182, 0.195090322016128267848285}, {-0.970031253194543992603984, 0.242980179903263889948274},
    {-0.956940335732208864935798, 0.290284677254462367636192}, {-0.941544065183020778412509, 0.336889853392220050689253},
    {-0.923879532511286756128183, 0.38268343236508977172846}, {-0.9039892931234433315862, 0.427555093430282094320967},
    {-0.881921264348355029712757, 0.471396736825997648556388}, {-0.85772861000027206990227, 0.514102744193221726593694},
    {-0.831469612302545237078788, 0.555570233019602224742831}, {-0.803207531480644909806677, 0.595699304492433343467037},
    {-0.773010453362736960810907, 0.634393284163645498215172}, {-0.740951125354959091175617, 0.671558954847018400625377},
    {-0.707106781186547524400844, 0.707106781186547524400844}, {-0.671558954847018400625377, 0.740951125354959091175617},
    {-0.634393284163645498215172, 0.773010453362736960810907}, {-0.595699304492433343467037, 0.803207531480644909806677},
    {-0.555570233019602224742831, 0.831469612302545237078788}, {-0.514102744193221726593694, 0.85772861000027206990227},
    {-0.471396736825997648556388, 0.881921264348355029712757}, {-0.427555093430282094320967, 0.9039892931234433315862},
    {-0.38268343236508977172846, 0.923879532511286756128183}, {-0.336889853392220050689253, 0.941544065183020778412509},
    {-0.290284677254462367636192, 0.956940335732208864935798}, {-0.242980179903263889948274, 0.970031253194543992603984},
    {-0.195090322016128267848285, 0.980785280403230449126182}, {-0.14673047445536175165885, 0.989176509964780973451674},
    {-0.0980171403295606019941956, 0.995184726672196886244837}, {-0.049067674327418014254955, 0.998795456205172392714772},
    {0.0, 1.0}, {0.049067674327418014254955, 0.998795456205172392714772},
    {0.0980171403295606019941956, 0.995184726672196886244837}, {0.14673047445536175165885, 0.989176509964780973451674},
    {0.195090322016128267848285, 0.980785280403230449126182}, {0.242980179903263889948274, 0.970031253194543992603984},
    {0.290284677254462367636192, 0.956940335732208864935798}, {0.336889853392220050689253, 0.941544065183020778412509},
    {0.38268343236508977172846, 0.923879532511286756128183}, {0.427555093430282094320967, 0.9039892931234433315862},
    {0.471396736825997648556388, 0.881921264348355029712757}, {0.514102744193221726593694, 0.85772861000027206990227},
    {0.555570233019602224742831, 0.831469612302545237078788}, {0.595699304492433343467037, 0.803207531480644909806677},
    {0.634393284163645498215172, 0.773010453362736960810907}, {0.671558954847018400625377, 0.740951125354959091175617},
    {0.707106781186547524400844, 0.707106781186547524400844}, {0.740951125354959091175617, 0.671558954847018400625377},
    {0.773010453362736960810907, 0.634393284163645498215172}, {0.803207531480644909806677, 0.595699304492433343467037},
    {0.831469612302545237078788, 0.555570233019602224742831}, {0.85772861000027206990227, 0.514102744193221726593694},
    {0.881921264348355029712757, 0.471396736825997648556388}, {0.9039892931234433315862, 0.427555093430282094320967},
    {0.923879532511286756128183, 0.38268343236508977172846}, {0.941544065183020778412509, 0.336889853392220050689253},
    {0.956940335732208864935798, 0.290284677254462367636192}, {0.970031253194543992603984, 0.242980179903263889948274},
    {0.980785280403230449126182, 0.195090322016128267848285}, {0.989176509964780973451674, 0.14673047445536175165885},
    {0.995184726672196886244837, 0.0980171403295606019941956}, {0.998795456205172392714772, 0.049067674327418014254955}};

// a * W_128^J for a compile-time J
template <int J> __device__ __forceinline__ v2f v2q_mul_w128(v2f a) {
    constexpr int K = J & 127;
    if constexpr (K == 0) return a;
    else if constexpr (K == 32) return pk_mul_mi(a);          // -i
    else if constexpr (K == 64) return v2f{-a.x, -a.y};
    else if constexpr (K == 96) return pk_mul_mi(v2f{-a.x, -a.y});
    else return pk_cmul_const(a, kW128[K][0], kW128[K][1]);
}

#ifndef V2Q_WIN_CHUNK
#define V2Q_WIN_CHUNK 4
#endif
#ifndef V2Q_ST_AUX
#define V2Q_ST_AUX 0  // output stores: default cache policy -- the pair's halves of every line meet in L2 (WRITE_SIZE 1.00x the output);
#endif                // non-temporal (2) lets a quarter of the lines go out half full (1.29x) and is +-4 % either way depending on the line order
#ifndef V2Q_M0
#define V2Q_M0 -1  // request schedule overrides (experiments): see v2q_body
#define V2Q_M1 -1
#define V2Q_M2 -1
#endif
#ifndef V2Q_ROLLING
#define V2Q_ROLLING 2  // request schedule of the instalments that do not fit (v2q_body): 2 = rolling with the Hann window only, 1 = always, 0 = never
#endif
#ifndef V2Q_LD_AUX
#define V2Q_LD_AUX 0  // sample loads: default policy (every byte is read four times: two workgroups, two lines at 50 % overlap)
#endif

// One workgroup of a pair: HALF = 0 forms y0 / y1, HALF = 1 forms y2 / y3 (the header's formulas).  Which one is a template
// parameter of the BODY, chosen by one wave-uniform branch at the top of the kernel: each side is straight-line code with
// compile-time twiddle constants, and no value of the line loop meets its twin of the other side.
template <int KIND, bool HAS_WIN, bool BE, int HALF>
__device__ __forceinline__ void v2q_body(const V2qArgs &a, uint32_t line0, uint32_t lines_wg, unsigned char *smem) {
    using PL = Plan2<14>;
    using RW = Raw2<KIND>;
    using raw_t = typename RW::type;
    constexpr int BPS = RW::BPS, Q = PL::N, N = 4 * Q, T = PL::T, E = PL::E;
    static_assert(E == 32 && T == 512 && N / T == 128, "n = t + 512 m, W_N^(512 m) = W_128^m");
    // how many m (four samples each) of the next line are in flight: behind the second transform's input (52 registers are
    // free there) / behind its epilogue (150) / at the top of the next line (214); the rest follows when the first SPLIT have
    // been combined -- 8 registers per m for cf32, 4 for the 2- and 4-byte formats
    constexpr int REG = BPS == 8 ? 2 : 1;  // registers per raw sample
    // (chosen per format from the register allocator's verdicts -- tools/kernel_resources.py: cf32 1 / 12 / 20 spills nothing
    // at all, 2 / 12 / 20 eleven registers, 2 / 12 / 22 twenty-three; ci16 and cu8 hold a whole line; ci8's sign extensions cost a few)
    constexpr bool TIGHT = KIND == K_CI8;
    constexpr int M0 = V2Q_M0 >= 0 ? V2Q_M0 : (REG == 2 ? 0 : TIGHT ? 4 : 12), M1 = V2Q_M1 >= 0 ? V2Q_M1 : (REG == 2 ? 8 : TIGHT ? 20 : 32),
                  M2 = V2Q_M2 >= 0 ? V2Q_M2 : (REG == 2 ? 14 : TIGHT ? 24 : 32), SPLIT = 8;
    static_assert(M0 <= M1 && M1 <= M2 && M2 <= E, "request schedule");
    const int t = threadIdx.x;
    v2f *lds = reinterpret_cast<v2f *>(smem);
    v2f *tab = reinterpret_cast<v2f *>(smem + (size_t)PL::LINE * 8);
    const v2f *__restrict__ tw = static_cast<const v2f *>(a.tw_q);
    const v2f *__restrict__ twf = static_cast<const v2f *>(a.tw_full);

    fill_tables<14, 1>(tab, tw, t);
    v2d *wtab = reinterpret_cast<v2d *>(tab + p2_tab_entries<14>());  // W_128^m in fp64 (the Hann window's cosine and sine)
    v2d *wt64s = wtab + 32;                                           // W_N^t of every thread
    if constexpr (HAS_WIN) {
        if (t < 32) wtab[t] = static_cast<const v2d *>(a.tw_full64)[T * t];
        wt64s[t] = static_cast<const v2d *>(a.tw_full64)[t];
    }
    v2f twl[16];
#pragma unroll
    for (int r = 1; r < 16; ++r) twl[r] = tw[(r * t) & (Q - 1)];
    constexpr int PA = 2 * HALF, PB = 2 * HALF + 1;  // y_PA through the first transform, y_PB through the second
    const v2f wA = twf[(PA * t) & (N - 1)];          // W_N^(PA t)
    const v2f wB = twf[(PB * t) & (N - 1)];          // W_N^(PB t)
    const bool db = a.out_fmt == OUT_DB20_F32;

    const uint32_t line_bytes = a.hop * BPS * a.ilv, line_floats = (uint32_t)N * a.ilv;  // from one of the pair's lines to its next
    uint32_t lw = lines_wg;
    asm volatile("" : "+s"(lw));  // (keeps the descriptor's size scalar: spec_k_v2h.hip)
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.iq) + (uint64_t)line0 * a.hop * BPS, 0, (lw - 1) * line_bytes + (uint32_t)N * BPS, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(a.out + (uint64_t)line0 * N, 0, ((lw - 1) * line_floats + (uint32_t)N) * 4u, 0x00020000);
    const int voff = t * BPS, ovoff = t * 16 + HALF * 8;
    constexpr int AUX = V2Q_LD_AUX, ST_AUX = V2Q_ST_AUX;

    raw_t r[4][E];  // r[q][m] = x[t + 512 m + q Q] of the line being loaded
    // one m = four samples, in the order they are combined: x0, x2, x1, x3
    auto request = [&](int off, auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
#ifdef V2Q_ABL_NOLOAD  // ablation (results wrong by construction): every line re-reads the workgroup's first
        off = 0;
#endif
        r[0][m] = RW::template load<AUX>(src, voff, off + (m * T) * BPS);
        r[2][m] = RW::template load<AUX>(src, voff, off + (m * T + 2 * Q) * BPS);
        r[1][m] = RW::template load<AUX>(src, voff, off + (m * T + Q) * BPS);
        r[3][m] = RW::template load<AUX>(src, voff, off + (m * T + 3 * Q) * BPS);
    };
    auto request_range = [&](int off, auto lo_tag, auto hi_tag) {
        constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
        v2h_for_each([&](auto mt) { request(off, std::integral_constant<int, LO + decltype(mt)::value>{}); },
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
        v2f v[E], dd[E];
        // ---- the radix-4 step: v = first transform's input, dd = second transform's input before its twiddle ----
        v2d wt64 = v2d{1.0, 0.0};
        if constexpr (HAS_WIN) wt64 = wt64s[t];
        v2h_for_each([&](auto m_tag) {
            constexpr int m = decltype(m_tag)::value;
            // the rest of the line (cf32: the whole line does not fit the registers).  Without a window: all E - M2 instalments at once
            // at step SPLIT, into the registers of the first SPLIT.  With the Hann window (its fp64 cosines take registers of their
            // own): one instalment per step into the registers the step before has just freed, M2 in flight all the way -- 119 -> 41
            // spilled registers, 10.10 -> 7.95 ms on the n65536f shape (0.213 -> 0.270 of 8 TB/s; without a window the rolling
            // schedule spills 5 and loses 0.8 %: profiles/r05_v2q.txt, variant v2qroll)
            constexpr bool ROLL = V2Q_ROLLING == 1 || (V2Q_ROLLING == 2 && HAS_WIN);
            if constexpr (ROLL) {
                if constexpr (m >= 1 && M2 + m - 1 < E) request((int)(line * line_bytes), std::integral_constant<int, M2 + m - 1>{});
            } else {
                if constexpr (m == SPLIT && M2 < E) request_range((int)(line * line_bytes), IM2{}, IE{});
            }
            v2f x0 = RW::dec(BE ? RW::swap(r[0][m]) : r[0][m]);  // SMH:87-91 byte order
            v2f x2 = RW::dec(BE ? RW::swap(r[2][m]) : r[2][m]);
            v2f x1 = RW::dec(BE ? RW::swap(r[1][m]) : r[1][m]);
            v2f x3 = RW::dec(BE ? RW::swap(r[3][m]) : r[3][m]);
            if constexpr (HAS_WIN) {
                // Hann: w[n] = 1/2 - 1/2 cos(2 pi n / N); W_N^n = W_N^t W_128^m = (cos, -sin) in fp64 (spec_k_v2h.hip: formed from
                // fp32 twiddles the cosine's error reached the tolerance); the four quarters see cos, -sin, -cos, sin
                const v2d cs = wtab[m];  // one broadcast LDS read
                const double c = __builtin_fma(wt64.x, cs.x, -(wt64.y * cs.y)), ms = __builtin_fma(wt64.x, cs.y, wt64.y * cs.x);
                const float w0 = (float)__builtin_fma(-0.5, c, 0.5), w2 = (float)__builtin_fma(0.5, c, 0.5);
                const float w1 = (float)__builtin_fma(-0.5, ms, 0.5), w3 = (float)__builtin_fma(0.5, ms, 0.5);
                x0 *= v2f{w0, w0}; x1 *= v2f{w1, w1}; x2 *= v2f{w2, w2}; x3 *= v2f{w3, w3};
            }
            const v2f ls = x0 + x2, ld = x0 - x2, hs = x1 + x3, hd = x1 - x3;
            if constexpr (HALF == 0) {
                v[m] = ls + hs;              // y0
                dd[m] = pk_add_mi(ld, hd);   // y1 before its twiddle: ld - i hd
            } else {
                v[m] = pk_cmul(v2q_mul_w128<PA * m>(ls - hs), wA);  // y2 = (ls - hs) W_128^(2 m) W_N^(2 t)
                dd[m] = pk_sub_mi(ld, hd);                          // y3 before its twiddle: ld + i hd
            }
            asm volatile("" : "+v"(dd[m]));  // computed HERE (spec_k_v2h.hip: left alone hipcc keeps the samples alive instead)
            // (window: the fp64 cosines four registers at a time -- left alone the scheduler reads all 32 table entries first and
            // forms all 128 window values before the first sample is touched: ~100 registers spilled)
            if constexpr (HAS_WIN && (m % V2Q_WIN_CHUNK) == V2Q_WIN_CHUNK - 1) __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, E>{});

        v2_fft<14>(v, t, lds, tab, twl);
        float de[E];
        constexpr bool BOUNDED = KIND != K_CF32;
        v2h_epilogue<BOUNDED, E>(v, RW::SCALE, db, de);

        // ---- second transform: dd W_128^(PB m) W_N^(PB t) ----
        v2h_for_each([&](auto mt) {
            constexpr int m = decltype(mt)::value;
            v[m] = pk_cmul(v2q_mul_w128<PB * m>(dd[m]), wB);
        }, std::make_integer_sequence<int, E>{});
        request_range(next_off, I0{}, IM0{});
        v2_fft<14>(v, t, lds, tab, twl);
        float dq[E];
        v2h_epilogue<BOUNDED, E>(v, RW::SCALE, db, dq);
        request_range(next_off, IM0{}, IM1{});
        // ---- bins 4k + 2h, 4k + 2h + 1 (k = t + 512 m) at columns 4c + 2h, 4c + 2h + 1, c = (k + Q/2) mod Q   (SS:78) ----
        const int out_off = (int)(line * line_floats * 4u);
#ifdef V2Q_ABL_NOSTORE  // ablation: no output (the test keeps the epilogues alive)
        if (de[0] == 123.456f)
#endif
#pragma unroll
        for (int m = 0; m < E; ++m)
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(de[m]), __float_as_uint(dq[m])}, dst, ovoff,
                                                  out_off + ((m + E / 2) & (E - 1)) * T * 16, ST_AUX);
        // (behind the stores for real: left to itself the scheduler hoists these requests above them, where the 64 registers of
        // the finished bins are still taken -- eleven registers spilled)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M1 < M2) request_range(next_off, IM1{}, IM2{});
    }
}

template <int KIND, bool HAS_WIN, bool BE>
__global__ __launch_bounds__(Plan2<14>::T, 2) void v2q_kernel(const V2qArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the pair: workgroups 16 g + i (first) and 16 g + 8 + i (second), both on XCD i under the round-robin dispatch
    const uint32_t half = (blockIdx.x >> 3) & 1u, g = blockIdx.x >> 4, i = blockIdx.x & 7u;
    uint32_t line0, lines_wg;
    if (a.ilv == 1) {  // pair 8 g + i: `run` consecutive lines
        line0 = (g * 8u + i) * a.run;
        if (line0 >= a.n_lines) return;  // (the whole workgroup: the grid is rounded up)
        lines_wg = a.n_lines - line0 < a.run ? a.n_lines - line0 : a.run;
    } else {           // block (g / 16) * 8 + i of 16 * run lines, shared by the sixteen pairs g % 16 of XCD i
        const uint32_t slot = g & 15u, block = (g >> 4) * 8u + i;
        line0 = block * 16u * a.run + slot;
        if (line0 >= a.n_lines) return;
        lines_wg = (a.n_lines - line0 + 15u) / 16u;  // lines line0, line0 + 16, ... below n_lines
        if (lines_wg > a.run) lines_wg = a.run;
    }
    if (half == 0) v2q_body<KIND, HAS_WIN, BE, 0>(a, line0, lines_wg, smem);
    else v2q_body<KIND, HAS_WIN, BE, 1>(a, line0, lines_wg, smem);
}

template <int KIND, bool HAS_WIN, bool BE> hipError_t v2q_launch1(const V2qArgs &a, hipStream_t s) {
    constexpr size_t lds = p2_lds_bytes<14>() + (32 + (HAS_WIN ? Plan2<14>::T : 0)) * sizeof(v2d);
    static_assert(lds <= 160 * 1024, "one workgroup's LDS");
    auto kern = v2q_kernel<KIND, HAS_WIN, BE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint32_t grid;
    if (a.ilv == 1) grid = ((a.n_lines + a.run - 1) / a.run + 7) / 8 * 16;                  // pairs rounded up to eight, two workgroups each
    else grid = ((a.n_lines + 16 * a.run - 1) / (16 * a.run) + 7) / 8 * 16 * 16;           // blocks rounded up to eight, sixteen pairs each
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Plan2<14>::T), lds, s, a);
    return hipGetLastError();
}

template <int KIND, bool BE> hipError_t v2q_launch_kind(const V2qArgs &a, hipStream_t s) {
    return a.win ? v2q_launch1<KIND, true, BE>(a, s) : v2q_launch1<KIND, false, BE>(a, s);
}

}  // namespace

bool v2q_applicable(int log2n, int kind, int out_fmt, uint64_t n_lines, uint32_t hop) {
    if (log2n != 16) return false;
    if (kind != K_CF32 && kind != K_CI16 && kind != K_CU8 && kind != K_CI8) return false;
    if (out_fmt != OUT_DB20_F32 && out_fmt != OUT_POW_F32) return false;
    return n_lines > 0 && n_lines < (1ull << 31) && hop <= (8u << log2n);
}

hipError_t launch_v2q_spectro(const WfArgs &w, const void *tw_q, const void *tw_full64, uint32_t run, int interleave, hipStream_t s) {
    V2qArgs a{};
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.run = run; a.ilv = interleave ? 16u : 1u;
    a.tw_q = tw_q; a.tw_full = w.tw; a.tw_full64 = tw_full64; a.win = tw_full64 ? w.win : nullptr;
    a.out = static_cast<float *>(w.out); a.out_fmt = w.out_fmt;
#ifdef V2Q_ONLY_KIND  // development: one instantiation per compile (register experiments)
    return v2q_launch1<V2Q_ONLY_KIND, V2Q_ONLY_WIN, V2Q_ONLY_BE>(a, s);
#else
    switch (w.kind) {
    case K_CF32: return w.be ? v2q_launch_kind<K_CF32, true>(a, s) : v2q_launch_kind<K_CF32, false>(a, s);
    case K_CI16: return w.be ? v2q_launch_kind<K_CI16, true>(a, s) : v2q_launch_kind<K_CI16, false>(a, s);
    case K_CU8: return v2q_launch_kind<K_CU8, false>(a, s);
    case K_CI8: return v2q_launch_kind<K_CI8, false>(a, s);
    default: return hipErrorInvalidValue;
    }
#endif
}

}  // namespace specgpu
