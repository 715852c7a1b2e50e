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
};

template <typename R, int L1, int L2>
__global__ __launch_bounds__(Plan<L1>::WG) void large_cols_kernel(const LargeArgs a) {
    using LG = Large<R, L1, L2>;
    using PA = typename LG::PA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<R> *lds = reinterpret_cast<cx<R> *>(smem);
    const int tid = threadIdx.x;
    const int q0 = tid % LG::CA, t0 = tid / LG::CA;  // loads: columns fastest (contiguous samples)
    const int t1 = tid % PA::T, q1 = tid / PA::T;    // stores: k1 fastest (contiguous scratch)
    constexpr int TILES = LG::N2 / LG::CA;
    const uint32_t line = blockIdx.x / TILES, c0 = (blockIdx.x % TILES) * LG::CA;
    const uint8_t *src = a.iq + (uint64_t)line * a.hop * a.bps;
    const R *__restrict__ win = static_cast<const R *>(a.win);

    // sub-FFT twiddles W_N1 into LDS
    cx<R> *tab = lds + (size_t)LG::CA * LG::SA;
    for (int e = tid; e < PA::N; e += PA::WG) tab[e] = static_cast<const cx<R> *>(a.tw1)[e];

    cx<R> v[PA::E];
#pragma unroll
    for (int m = 0; m < PA::E; ++m) {
        const uint32_t n = (uint32_t)(t0 + m * PA::T) * LG::N2 + c0 + q0;
        v[m] = decode_sample<R>(src + (uint64_t)n * a.bps, a.kind, a.be != 0);
        if (win) { const R w = win[n]; v[m].x *= w; v[m].y *= w; }
    }
    __syncthreads();  // table visible
    fft_line_remap<R, L1>(v, t0, lds + (size_t)q0 * LG::SA, t1, lds + (size_t)q1 * LG::SA, tab);
    // inter-step twiddle W_N^(n2 k1), k1 = t1 + m T: W^(n2 t1) * (W^(n2 T))^m by recurrence in
    // fp64 (two table reads per thread instead of sixteen scattered ones; 15 roundings of 1e-16)
    const cx<double> *__restrict__ twn = static_cast<const cx<double> *>(a.twn);
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

template <typename R, int L1, int L2>
__global__ __launch_bounds__(Plan<L2>::WG) void large_rows_kernel(const LargeArgs a) {
    using LG = Large<R, L1, L2>;
    using PB = typename LG::PB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<R> *lds = reinterpret_cast<cx<R> *>(smem);
    const int tid = threadIdx.x;
    const int q0 = tid % LG::CB, t0 = tid / LG::CB;  // rows k1 fastest: both the scratch reads and the final stores
    constexpr int TILES = LG::N1 / LG::CB;
    const uint32_t line = blockIdx.x / TILES, r0 = (blockIdx.x % TILES) * LG::CB;
    const cx<R> *src = static_cast<const cx<R> *>(a.scratch) + (uint64_t)line * LG::N + r0 + q0;

    cx<R> *tab = lds + (size_t)LG::CB * LG::SB;
    for (int e = tid; e < PB::N; e += PB::WG) tab[e] = static_cast<const cx<R> *>(a.tw2)[e];
    cx<R> v[PB::E];
#pragma unroll
    for (int m = 0; m < PB::E; ++m) v[m] = src[(uint64_t)(t0 + m * PB::T) * LG::N1];  // [n2][k1]
    __syncthreads();  // table visible
    // no role change is needed here (k1 stays the fast index), only the padded line stride
    fft_line<R, L2>(v, t0, lds + (size_t)q0 * LG::SB, tab);
    const uint64_t base = (uint64_t)line * LG::N;
#pragma unroll
    for (int m = 0; m < PB::E; ++m) {
        const uint32_t k = (r0 + q0) + (uint32_t)LG::N1 * (t0 + m * PB::T);
        store_bin<R>(a.out, base + ((k + LG::N / 2) & (LG::N - 1)), v[m], a.out_fmt);  // SS:78
    }
}

template <typename R, int L1, int L2> hipError_t launch_large(const LargeArgs &a, hipStream_t s) {
    using LG = Large<R, L1, L2>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&large_cols_kernel<R, L1, L2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LG::LDS_A);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&large_rows_kernel<R, L1, L2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LG::LDS_B);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((large_cols_kernel<R, L1, L2>), dim3(a.n_lines * (LG::N2 / LG::CA)), dim3(LG::PA::WG), LG::LDS_A, s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((large_rows_kernel<R, L1, L2>), dim3(a.n_lines * (LG::N1 / LG::CB)), dim3(LG::PB::WG), LG::LDS_B, s, a);
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

hipError_t launch_spectro_large(const WfArgs &w, int log2n, bool f64, const void *tw1, const void *tw2,
                                void *scratch, hipStream_t s) {
    LargeArgs a{};
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
