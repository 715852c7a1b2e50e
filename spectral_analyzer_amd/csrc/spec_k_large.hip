// spec_k_large.hip -- lines longer than the LDS holds (fp32: nfft >= 32768,
// fp64: nfft >= 16384; BASELINE configs[4] is 65536-point cf64).
//
// Four-step decomposition N = N1 * N2 with n = N2 n1 + n2, k = k1 + N1 k2:
//   kernel A ("columns"): for a tile of C adjacent columns n2, the N1-point
//       FFTs over n1 (input stride N2), times W_N^(n2 k1), written to a
//       scratch line in [n2][k1] order;
//   kernel B ("rows"):    for a tile of C adjacent k1, the N2-point FFTs over
//       n2 (scratch stride N1), then the epilogue; X[k1 + N1 k2] leaves with
//       k1 fastest, i.e. in runs of C bins.
// Each kernel flips its thread roles at the first LDS exchange
// (fft_line_remap): the side that touches global memory with the line index
// fastest gets coalesced runs, the other side gets the FFT's own stride-T order.
// Scratch is sized per chunk of lines (<= 256 MiB) so that it lives in the
// 256 MiB Infinity Cache between the two kernels.
#include <type_traits>

#include "spec_kernels.h"

namespace specgpu {

namespace {

template <typename R, int L1, int L2> struct Large {
    static constexpr int N1 = 1 << L1, N2 = 1 << L2, N = N1 * N2;
    using PA = Plan<L1>;  // column FFTs
    using PB = Plan<L2>;  // row FFTs
    static constexpr int CA = PA::LPW, CB = PB::LPW;          // lines (columns / rows) per workgroup
    static constexpr int SA = PA::N + 1, SB = PB::N + 1;      // padded LDS line strides (elements)
    // line buffers + the sub-FFT's own twiddle table W_N1 / W_N2 (global twiddle loads on the
    // critical path of every pass were the main cost of the first version of these kernels)
    static constexpr size_t LDS_A = ((size_t)CA * SA + PA::N) * sizeof(cx<R>), LDS_B = ((size_t)CB * SB + PB::N) * sizeof(cx<R>);
};

struct LargeArgs {
    const uint8_t *iq;  // first byte of line 0 of this chunk
    uint32_t n_lines;
    uint32_t hop, bps;
    int kind, be;
    const void *tw1, *tw2;        // W_N1, W_N2 tables (cx<R>)
    const void *twn;              // W_N table, always fp64 (cx<double>): inter-step twiddles by recurrence
    const void *win;              // R[N] or nullptr
    void *scratch;                // cx<R>[n_lines][N], [n2][k1] order
    void *out;
    int out_fmt;
    const uint32_t *run_if;       // nullptr, or: run only when *run_if != 0 (fall-back behind spec_k_team.hip)
};

// Both kernels walk TPW consecutive tiles per workgroup and request the next tile's operands before the
// FFT of the current one (the same software prefetch as the LDS-resident family): with 70 KiB of LDS per
// workgroup only two workgroups share a CU, too few to hide the load latency by occupancy alone.
constexpr uint32_t LARGE_TPW = 4;  // row kernel: consecutive tiles per workgroup
constexpr uint32_t LARGE_LPW = 8;  // column kernel: consecutive lines (of one column tile) per workgroup

// DIRECT: the samples already are cx<R> in memory (little-endian cf64 for double, cf32 for float), so a load
// needs no decode and can stay in flight behind the current tile's FFT
template <typename R, int L1, int L2, bool DIRECT>
__global__ __launch_bounds__(Plan<L1>::WG, 2) void large_cols_kernel(const LargeArgs a) {
    using LG = Large<R, L1, L2>;
    using PA = typename LG::PA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<R> *lds = reinterpret_cast<cx<R> *>(smem);
    const int tid = threadIdx.x;
    if (a.run_if && *a.run_if == 0) return;
    const int q0 = tid % LG::CA, t0 = tid / LG::CA;  // loads: columns fastest (contiguous samples)
    const int t1 = tid % PA::T, q1 = tid / PA::T;    // stores: k1 fastest (contiguous scratch)
    constexpr uint32_t TILES = LG::N2 / LG::CA;
    // a workgroup walks LARGE_LPW consecutive LINES of one column tile: at 50 % overlap line l+1's rows
    // n1 < N1/2 are line l's rows n1 + N1/2, i.e. this thread's registers m + E/2 -- half of every tile
    // after the first is a register move instead of a load
    const uint32_t c0 = (blockIdx.x % TILES) * LG::CA;
    const uint32_t l0 = (blockIdx.x / TILES) * LARGE_LPW, l1 = l0 + LARGE_LPW < a.n_lines ? l0 + LARGE_LPW : a.n_lines;
    const bool half = (uint64_t)a.hop * 2 == (uint64_t)LG::N;  // uniform
    const R *__restrict__ win = static_cast<const R *>(a.win);

    // sub-FFT twiddles W_N1 into LDS
    cx<R> *tab = lds + (size_t)LG::CA * LG::SA;
    for (int e = tid; e < PA::N; e += PA::WG) tab[e] = static_cast<const cx<R> *>(a.tw1)[e];

    auto load_rows = [&](uint32_t line, cx<R> (&x)[PA::E], auto first_tag) {  // rows m >= FIRST of a line's tile
        constexpr int FIRST = decltype(first_tag)::value;
        const uint8_t *src = a.iq + (uint64_t)line * a.hop * a.bps;
#pragma unroll
        for (int m = FIRST; m < PA::E; ++m) {
            const uint32_t n = (uint32_t)(t0 + m * PA::T) * LG::N2 + c0 + q0;
            if constexpr (DIRECT) x[m] = *reinterpret_cast<const cx<R> *>(src + (uint64_t)n * sizeof(cx<R>));
            else x[m] = decode_sample<R>(src + (uint64_t)n * a.bps, a.kind, a.be != 0);
        }
    };
    cx<R> nxt[PA::E];
    load_rows(l0, nxt, std::integral_constant<int, 0>{});
    __syncthreads();  // table visible
    const cx<double> *__restrict__ twn = static_cast<const cx<double> *>(a.twn);
    for (uint32_t line = l0; line < l1; ++line) {
        cx<R> v[PA::E];
#pragma unroll
        for (int m = 0; m < PA::E; ++m) v[m] = nxt[m];
        if (line + 1 < l1) {
            if (half) {
#pragma unroll
                for (int m = 0; m < PA::E / 2; ++m) nxt[m] = nxt[m + PA::E / 2];
                load_rows(line + 1, nxt, std::integral_constant<int, PA::E / 2>{});
            } else {
                load_rows(line + 1, nxt, std::integral_constant<int, 0>{});
            }
        }
        if (win) {
#pragma unroll
            for (int m = 0; m < PA::E; ++m) {
                const R w = win[(uint32_t)(t0 + m * PA::T) * LG::N2 + c0 + q0];
                v[m].x *= w;
                v[m].y *= w;
            }
        }
        fft_line_remap<R, L1>(v, t0, lds + (size_t)q0 * LG::SA, t1, lds + (size_t)q1 * LG::SA, tab);
        // inter-step twiddle W_N^(n2 k1), k1 = t1 + m T: W^(n2 t1) * (W^(n2 T))^m by recurrence in
        // fp64 (two table reads per thread instead of sixteen scattered ones; 15 roundings of 1e-16)
        const uint32_t n2 = c0 + q1;
        cx<double> w = twn[n2 * (uint32_t)t1];
        const cx<double> step = twn[n2 * (uint32_t)PA::T];
        cx<R> *dst = static_cast<cx<R> *>(a.scratch) + (uint64_t)line * LG::N + (uint64_t)n2 * LG::N1;
#pragma unroll
        for (int m = 0; m < PA::E; ++m) {
            const cx<double> z = cmul(cx<double>{(double)v[m].x, (double)v[m].y}, w);
            dst[t1 + m * PA::T] = cx<R>{(R)z.x, (R)z.y};
            w = cmul(w, step);
        }
    }
}

// two workgroups (LDS) of four waves per CU = 2 waves per SIMD: keep the allocation within 256 registers
template <typename R, int L1, int L2>
__global__ __launch_bounds__(Plan<L2>::WG, 2) void large_rows_kernel(const LargeArgs a) {
    using LG = Large<R, L1, L2>;
    using PB = typename LG::PB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<R> *lds = reinterpret_cast<cx<R> *>(smem);
    const int tid = threadIdx.x;
    if (a.run_if && *a.run_if == 0) return;
    const int q0 = tid % LG::CB, t0 = tid / LG::CB;  // rows k1 fastest: both the scratch reads and the final stores
    constexpr uint32_t TILES = LG::N1 / LG::CB;
    const uint32_t total = a.n_lines * TILES;
    const uint32_t g0 = blockIdx.x * LARGE_TPW, g1 = g0 + LARGE_TPW < total ? g0 + LARGE_TPW : total;

    cx<R> *tab = lds + (size_t)LG::CB * LG::SB;
    for (int e = tid; e < PB::N; e += PB::WG) tab[e] = static_cast<const cx<R> *>(a.tw2)[e];
    auto load_tile = [&](uint32_t g, auto &x) {
        const uint32_t line = g / TILES, r0 = (g % TILES) * LG::CB;
        const cx<R> *src = static_cast<const cx<R> *>(a.scratch) + (uint64_t)line * LG::N + r0 + q0;
#pragma unroll
        for (int m = 0; m < PB::E; ++m) x[m] = src[(uint64_t)(t0 + m * PB::T) * LG::N1];  // [n2][k1]
    };
    // fp64: the dB epilogue already needs most of the 256 registers, a second tile in flight spills 65 of them;
    // the tile loop alone (table set-up amortised) is kept
    constexpr bool PREFETCH = sizeof(R) == 4;
    cx<R> nxt[PREFETCH ? PB::E : 1];
    (void)nxt;
    if constexpr (PREFETCH) load_tile(g0, nxt);
    __syncthreads();  // table visible
    for (uint32_t g = g0; g < g1; ++g) {
        const uint32_t line = g / TILES, r0 = (g % TILES) * LG::CB;
        cx<R> v[PB::E];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int m = 0; m < PB::E; ++m) v[m] = nxt[m];
            if (g + 1 < g1) load_tile(g + 1, nxt);
        } else {
            load_tile(g, v);
        }
        // no role change is needed here (k1 stays the fast index), only the padded line stride
        fft_line<R, L2>(v, t0, lds + (size_t)q0 * LG::SB, tab);
        const uint64_t base = (uint64_t)line * LG::N;
#pragma unroll
        for (int m = 0; m < PB::E; ++m) {
            const uint32_t k = (r0 + q0) + (uint32_t)LG::N1 * (t0 + m * PB::T);
            store_bin<R>(a.out, base + ((k + LG::N / 2) & (LG::N - 1)), v[m], a.out_fmt);  // SS:78
        }
    }
}

// ---- the guarded fall-back behind the persistent team kernel: ONE launch for any number of lines ----------------
// The team kernel (spec_k_team.hip) needs all its workgroups resident at once; on a shared or partitioned GPU its
// bounded waits time out, it raises the abort word and leaves.  What runs then must not need residency of its own,
// and it should be ONE guarded launch, not one pair per chunk of lines (64 empty launches per 32 767-line call cost
// 2-3 % of the step).  Here every workgroup owns whole lines: it runs the column step of all of a line's tiles into
// a line-sized intermediate of its own (a.scratch: one per workgroup of the grid), then the row step of all tiles
// from it -- no other workgroup is ever waited for.  Its own stores reach L2 through the CU's write-through L1
// (s_waitcnt vmcnt(0) + barrier), the reads bypass that L1 (non-temporal), so a line's intermediate is never read
// stale although the buffer is rewritten line after line.  No prefetch, no overlap reuse: a correctness path.
template <typename R, int L1, int L2, bool DIRECT>
__global__ __launch_bounds__(256, 2) void large_solo_kernel(const LargeArgs a) {
    using LG = Large<R, L1, L2>;
    using PA = typename LG::PA;
    using PB = typename LG::PB;
    static_assert(PA::WG == 256 && PB::WG == 256, "both steps use 256-thread workgroups");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<R> *lds = reinterpret_cast<cx<R> *>(smem);
    const int tid = threadIdx.x;
    if (a.run_if && *a.run_if == 0) return;
    constexpr size_t LINES = (size_t)LG::CA * LG::SA > (size_t)LG::CB * LG::SB ? (size_t)LG::CA * LG::SA : (size_t)LG::CB * LG::SB;
    cx<R> *tab_a = lds + LINES, *tab_b = tab_a + PA::N;
    for (int e = tid; e < PA::N; e += 256) tab_a[e] = static_cast<const cx<R> *>(a.tw1)[e];
    for (int e = tid; e < PB::N; e += 256) tab_b[e] = static_cast<const cx<R> *>(a.tw2)[e];
    __syncthreads();
    const R *__restrict__ win = static_cast<const R *>(a.win);
    const cx<double> *__restrict__ twn = static_cast<const cx<double> *>(a.twn);
    cx<R> *mid = static_cast<cx<R> *>(a.scratch) + (uint64_t)blockIdx.x * LG::N;  // [n2][k1], this workgroup's own
    for (uint32_t line = blockIdx.x; line < a.n_lines; line += gridDim.x) {
        const uint8_t *src = a.iq + (uint64_t)line * a.hop * a.bps;
        {   // column step: N1-point transforms over n1 for every column n2, times W_N^(n2 k1)
            const int q0 = tid % LG::CA, t0 = tid / LG::CA, t1 = tid % PA::T, q1 = tid / PA::T;
            for (uint32_t c0 = 0; c0 < (uint32_t)LG::N2; c0 += LG::CA) {
                cx<R> v[PA::E];
#pragma unroll
                for (int m = 0; m < PA::E; ++m) {
                    const uint32_t n = (uint32_t)(t0 + m * PA::T) * LG::N2 + c0 + q0;
                    if constexpr (DIRECT) v[m] = *reinterpret_cast<const cx<R> *>(src + (uint64_t)n * sizeof(cx<R>));
                    else v[m] = decode_sample<R>(src + (uint64_t)n * a.bps, a.kind, a.be != 0);
                    if (win) { const R w = win[n]; v[m].x *= w; v[m].y *= w; }
                }
                fft_line_remap<R, L1>(v, t0, lds + (size_t)q0 * LG::SA, t1, lds + (size_t)q1 * LG::SA, tab_a);
                const uint32_t n2 = c0 + q1;
                cx<double> w = twn[n2 * (uint32_t)t1];
                const cx<double> step = twn[n2 * (uint32_t)PA::T];
                cx<R> *dst = mid + (uint64_t)n2 * LG::N1;
#pragma unroll
                for (int m = 0; m < PA::E; ++m) {
                    const cx<double> z = cmul(cx<double>{(double)v[m].x, (double)v[m].y}, w);
                    dst[t1 + m * PA::T] = cx<R>{(R)z.x, (R)z.y};
                    w = cmul(w, step);
                }
                __syncthreads();  // the line buffers are rewritten by the next tile
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's intermediate is in L2
        __syncthreads();
        {   // row step: N2-point transforms over n2 for every row k1, epilogue (SS:76-82)
            const int q0 = tid % LG::CB, t0 = tid / LG::CB;
            const uint64_t base = (uint64_t)line * LG::N;
            for (uint32_t r0 = 0; r0 < (uint32_t)LG::N1; r0 += LG::CB) {
                cx<R> v[PB::E];
                typedef R r2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int m = 0; m < PB::E; ++m) {  // L1 bypassed: the buffer was rewritten since this CU last read it
                    const r2 u = __builtin_nontemporal_load(reinterpret_cast<const r2 *>(mid + (uint64_t)(t0 + m * PB::T) * LG::N1 + r0 + q0));
                    v[m] = cx<R>{u.x, u.y};
                }
                fft_line<R, L2>(v, t0, lds + (size_t)q0 * LG::SB, tab_b);
#pragma unroll
                for (int m = 0; m < PB::E; ++m) {
                    const uint32_t k = (r0 + q0) + (uint32_t)LG::N1 * (t0 + m * PB::T);
                    store_bin<R>(a.out, base + ((k + LG::N / 2) & (LG::N - 1)), v[m], a.out_fmt);  // SS:78
                }
                __syncthreads();
            }
        }
        // the row step's reads of `mid` are complete (consumed by its transforms) before the next line's stores
    }
}

template <typename R, int L1, int L2> size_t solo_lds_bytes() {
    using LG = Large<R, L1, L2>;
    const size_t lines = (size_t)LG::CA * LG::SA > (size_t)LG::CB * LG::SB ? (size_t)LG::CA * LG::SA : (size_t)LG::CB * LG::SB;
    return (lines + LG::PA::N + LG::PB::N) * sizeof(cx<R>);
}

template <typename R, int L1, int L2> hipError_t launch_solo(const LargeArgs &a, uint32_t grid, hipStream_t s) {
    const bool direct = !a.be && a.kind == (sizeof(R) == 8 ? K_CF64 : K_CF32);
    auto fn = direct ? &large_solo_kernel<R, L1, L2, true> : &large_solo_kernel<R, L1, L2, false>;
    const size_t lds = solo_lds_bytes<R, L1, L2>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <typename R, int L1, int L2> hipError_t launch_large(const LargeArgs &a, hipStream_t s) {
    using LG = Large<R, L1, L2>;
    // samples that already are cx<R> in memory need no decode
    const bool direct = !a.be && a.kind == (sizeof(R) == 8 ? K_CF64 : K_CF32);
    auto cols = direct ? &large_cols_kernel<R, L1, L2, true> : &large_cols_kernel<R, L1, L2, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cols), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LG::LDS_A);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&large_rows_kernel<R, L1, L2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LG::LDS_B);
    if (e != hipSuccess) return e;
    const uint32_t wgs_a = ((a.n_lines + LARGE_LPW - 1) / LARGE_LPW) * (LG::N2 / LG::CA), tiles_b = a.n_lines * (LG::N1 / LG::CB);
    hipLaunchKernelGGL(cols, dim3(wgs_a), dim3(LG::PA::WG), LG::LDS_A, s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((large_rows_kernel<R, L1, L2>), dim3((tiles_b + LARGE_TPW - 1) / LARGE_TPW), dim3(LG::PB::WG),
                       LG::LDS_B, s, a);
    return hipGetLastError();
}

}  // namespace

bool large_split(int log2n, bool f64, int *l1, int *l2) {
    if (f64 ? (log2n < 14) : (log2n < 15)) return false;
    if (log2n > 16) return false;
    *l1 = log2n == 16 ? 8 : 7;
    *l2 = log2n - *l1;
    return true;
}

size_t large_scratch_bytes_per_line(int log2n, bool f64) { return ((size_t)1 << log2n) * (f64 ? 16 : 8); }

// One guarded launch over all n_lines (n_lines < 2^32): `scratch` holds grid line-sized intermediates.
hipError_t launch_spectro_large_solo(const WfArgs &w, int log2n, bool f64, const void *tw1, const void *tw2, void *scratch,
                                     uint32_t grid, hipStream_t s, const uint32_t *run_if) {
    LargeArgs a{};
    a.run_if = run_if;
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.bps = w.bps; a.kind = w.kind; a.be = w.be;
    a.tw1 = tw1; a.tw2 = tw2; a.twn = w.tw; a.win = w.win; a.scratch = scratch; a.out = w.out; a.out_fmt = w.out_fmt;
    if (f64) {
        switch (log2n) {
        case 14: return launch_solo<double, 7, 7>(a, grid, s);
        case 15: return launch_solo<double, 7, 8>(a, grid, s);
        case 16: return launch_solo<double, 8, 8>(a, grid, s);
        }
    } else {
        switch (log2n) {
        case 15: return launch_solo<float, 7, 8>(a, grid, s);
        case 16: return launch_solo<float, 8, 8>(a, grid, s);
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_spectro_large(const WfArgs &w, int log2n, bool f64, const void *tw1, const void *tw2,
                                void *scratch, hipStream_t s, const uint32_t *run_if) {
    LargeArgs a{};
    a.run_if = run_if;
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.bps = w.bps; a.kind = w.kind; a.be = w.be;
    a.tw1 = tw1; a.tw2 = tw2; a.twn = w.tw; a.win = w.win; a.scratch = scratch; a.out = w.out; a.out_fmt = w.out_fmt;
    if (f64) {
        switch (log2n) {
        case 14: return launch_large<double, 7, 7>(a, s);
        case 15: return launch_large<double, 7, 8>(a, s);
        case 16: return launch_large<double, 8, 8>(a, s);
        }
    } else {
        switch (log2n) {
        case 15: return launch_large<float, 7, 8>(a, s);
        case 16: return launch_large<float, 8, 8>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace specgpu
