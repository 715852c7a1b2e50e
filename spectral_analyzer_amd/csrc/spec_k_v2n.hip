// spec_k_v2n.hip -- spectrogram lines of 64 and 128 points (the low end of the reference's NFFT slider,
// main-scene.fxml:129-132) on the packed-fp32 FFT core of spec_v2.h.
//
// With 16 points per thread such a line is only T = 4 or 8 lanes wide: left to itself every lane group would touch
// global memory in 32- or 64-byte pieces (the family's own kernel was measured slower than the generic one for these
// sizes).  Here a WAVE works on LB = 64 / T CONSECUTIVE lines at a time and does all global traffic together:
//   * the block's input span ((LB - 1) hop + N samples, contiguous) is read with 16 bytes per lane, 1 KiB per
//     instruction, and parked in the wave's own LDS region with one line-group's width of padding per line stride
//     (the lanes of neighbouring lines then start in neighbouring banks); overlapping lines share it -- every
//     input byte is requested once per block;
//   * every lane group picks its line's 16 samples per thread out of the span (stride T, as the FFT wants them);
//   * the transform is spec_v2.h's: radix 4 (or 8) in registers, ONE wave-local exchange through the same LDS
//     region (no s_barrier anywhere in this kernel), radix 16 with its twiddles in registers, the family's epilogue;
//   * the LB finished lines -- contiguous in the output -- are laid out in LDS (fftshift folded into the index) and
//     leave with 16 bytes per lane, 1 KiB per instruction.
// Any hop from 16 bytes' worth of samples up to N (the reference's own hop), any of the byte / 16-bit / float formats,
// either byte order, optional window, dB or power output.
#include "spec_v2.h"

namespace specgpu {

namespace {

struct V2nArgs {
    const uint8_t *iq;   // first byte of line 0
    uint32_t n_lines, hop;
    uint32_t pad_shift;  // floor(log2(hop * BPS)): one pad unit per 2^pad_shift bytes of span
    uint32_t in_bytes;   // v2n_dma_kernel: bytes of a wave's landing buffer for this call's hop
    uint32_t chunk;      // v2n_dma_kernel: consecutive blocks a wave takes before it jumps gridDim.x chunks ahead (0: automatic = 4; "lines_per_wg" sets it)
    const void *tw, *win;
    float *out;
    int out_fmt;
};

typedef uint32_t vu4 __attribute__((ext_vector_type(4)));
#ifndef V2N_DMA
#define V2N_DMA 1  // 0: v2n_kernel for every hop (build.py --variant v2nold)
#endif
#ifndef V2N_OCC
#define V2N_OCC 3  // waves per SIMD asked of the register allocator (147 registers: no spills at 3)
#endif

// bytes of a wave's LDS region: the larger of the padded input span (maximised over every admissible hop), the
// exchange buffer of the block's lines and their output staging
template <int L, int BPS> constexpr int v2n_region_bytes() {
    using PL = Plan2<L>;
    constexpr int N = PL::N, T = PL::T, LB = 64 / T, PADU = T * BPS;
    int in_max = 0;
    for (int hop = (16 + BPS - 1) / BPS; hop <= N; ++hop) {
        const int lb = hop * BPS;
        int sh = 0;
        while ((2 << sh) <= lb) ++sh;
        const int span = 15 + (LB - 1) * lb + N * BPS;  // worst misalignment
        const int padded = span + ((span >> sh) + 1) * PADU;
        if (padded > in_max) in_max = padded;
    }
    const int ex = LB * PL::LINE * 8, out = LB * (N + 4) * 4;
    int m = in_max > ex ? in_max : ex;
    if (out > m) m = out;
    return (m + 15) & ~15;
}

template <int L, int KIND, bool BE>
__global__ __launch_bounds__(256, V2N_OCC) void v2n_kernel(const V2nArgs a) {
    using PL = Plan2<L>;
    using RW = Raw2<KIND>;
    constexpr int N = PL::N, T = PL::T, E = PL::E, BPS = RW::BPS;
    constexpr int LB = 64 / T;                      // lines per wave and block
    constexpr int PADU = T * BPS;                   // pad unit: the bytes one line's lanes read per instruction
    constexpr int OUT_STRIDE = N + 4;               // floats per staged output line: neighbouring lines 4 banks apart
    constexpr int REGION = v2n_region_bytes<L, BPS>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *region = smem + (size_t)wave * REGION;
    const int t = lane % T, j = lane / T;           // this lane: butterfly index t of the block's line j
    const v2f *__restrict__ tw = static_cast<const v2f *>(a.tw);
    v2f twl[16];
#pragma unroll
    for (int r = 1; r < 16; ++r) twl[r] = tw[(r * t) & (N - 1)];
    const float *win = static_cast<const float *>(a.win);
    float w[E];
    if (win) {
#pragma unroll
        for (int m = 0; m < E; ++m) w[m] = win[t + m * T];
    }
    const uint32_t n_blocks = (a.n_lines + LB - 1) / LB;
    const uint32_t waves_total = gridDim.x * 4u, wave_id = blockIdx.x * 4u + (uint32_t)wave;
    // consecutive blocks to one wave: the N - hop samples two blocks share are then re-read from L2 by the same CU
    const uint32_t per_wave = (n_blocks + waves_total - 1) / waves_total;
    const uint32_t b0 = wave_id * per_wave, b1 = b0 + per_wave < n_blocks ? b0 + per_wave : n_blocks;
    const uint32_t line_bytes = a.hop * BPS, psh = a.pad_shift;
    auto padded = [&](uint32_t x) -> uint32_t { return x + (x >> psh) * (uint32_t)PADU; };
    // (Requesting the next block's span before the current block is transformed -- five 16-byte pieces per lane in
    // flight behind the FFT -- was measured SLOWER, 47 % -> 38 % of the peak at 64 points: 168 registers with 10 dwords
    // spilled, and a wait at the top of the loop that also drains the previous block's output stores.  Twelve waves
    // per CU hide the latency better than one wave's prefetch does.)
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t l0 = b * LB;
        const uint32_t nb = a.n_lines - l0 < (uint32_t)LB ? a.n_lines - l0 : (uint32_t)LB;  // valid lines of this block
        // ---- input span -> LDS (16 bytes per lane; the first chunk starts at the 16-byte boundary below the span) ----
        const uint8_t *first = a.iq + (uint64_t)l0 * line_bytes;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(first) & 15u);
        const uint32_t span = mis + (nb - 1) * line_bytes + (uint32_t)N * BPS;  // bytes from the aligned base
        const uint32_t chunks = (span + 15u) / 16u;
        const vu4 *gsrc = reinterpret_cast<const vu4 *>(first - mis);
        for (uint32_t c = (uint32_t)lane; c < chunks; c += 64u) {
            const vu4 u = __builtin_nontemporal_load(gsrc + c);
            *reinterpret_cast<vu4 *>(region + padded(16u * c)) = u;
        }
        v2_sync<L>();  // wave-local: the span is in LDS for every lane of this wave
        // ---- this lane's 16 samples: x[j hop + t + m T] ----
        v2f v[E];
        const uint32_t base = mis + (uint32_t)j * line_bytes + (uint32_t)t * BPS;
        // SPLIT (cf32 only): the recording starts at 4 mod 8 bytes (include/specgpu.h promises component alignment only),
        // so a sample's two words may lie on either side of a pad gap: each word gets its own padded address.  `mis & 4`
        // is the same for every lane and block of the call (line strides and lane offsets are multiples of 8).
        auto pick = [&](auto split_tag) {
            constexpr bool SPLIT = decltype(split_tag)::value;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const uint32_t x = base + (uint32_t)(m * T * BPS);
                const unsigned char *p = region + padded(x);
                typename RW::type r;
                if constexpr (BPS == 8) {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);  // 4-byte aligned (checked by the host)
                    if constexpr (SPLIT) r = typename RW::type{q[0], *reinterpret_cast<const uint32_t *>(region + padded(x + 4u))};
                    else r = typename RW::type{q[0], q[1]};
                } else if constexpr (BPS == 4) {
                    r = *reinterpret_cast<const uint32_t *>(p);
                } else {
                    r = *reinterpret_cast<const uint16_t *>(p);
                }
                v[m] = RW::dec(BE ? RW::swap(r) : r);  // SMH:87-91 byte order
            }
        };
        if (BPS == 8 && (mis & 4u)) pick(std::true_type{});
        else pick(std::false_type{});
        if (win) {
#pragma unroll
            for (int m = 0; m < E; ++m) v[m] *= v2f{w[m], w[m]};
        }
        v2_sync<L>();  // everybody has its samples: the region becomes the exchange buffer
        v2_fft<L>(v, t, reinterpret_cast<v2f *>(region) + (size_t)j * PL::LINE, static_cast<const v2f *>(nullptr), twl);
        float d[E];
        constexpr bool BOUNDED = KIND != K_CF32;
        v2_epilogue<BOUNDED, E>(v, RW::SCALE, a.out_fmt == OUT_DB20_F32, d);
        v2_sync<L>();  // the exchange has been read: the region becomes the output staging
        float *stage = reinterpret_cast<float *>(region);
#pragma unroll
        for (int m = 0; m < E; ++m)  // bin k = t + m T at column (k + N/2) mod N   (SS:78)
            stage[j * OUT_STRIDE + ((t + m * T + N / 2) & (N - 1))] = d[m];
        v2_sync<L>();
        // ---- the block's nb lines are contiguous in the output: 16 bytes per lane ----
        float *dst = a.out + (uint64_t)l0 * N;
        const uint32_t out_chunks = nb * (uint32_t)(N / 4);
        for (uint32_t c = (uint32_t)lane; c < out_chunks; c += 64u) {
            const uint32_t row = c / (uint32_t)(N / 4), col = (c % (uint32_t)(N / 4)) * 4u;
            const vu4 u = *reinterpret_cast<const vu4 *>(stage + row * OUT_STRIDE + col);
            __builtin_nontemporal_store(u, reinterpret_cast<vu4 *>(dst) + c);
        }
        v2_sync<L>();  // the staging has been read: the next block's span may land
    }
}

// ---- the same blocks with the NEXT block's input span already on its way (round 5) ------------------------------------
// v2n_kernel's waves spend 73 % of their cycles waiting (profiles/r05_n64_summary.md: twelve waves per CU do not cover a
// block's load -> transform -> store chain), and a prefetch into registers had been measured slower (above).  Here the span
// of block b + 1 is requested as soon as block b's samples have been picked out of the landing buffer, by LDS-DMA
// (global_load_lds_dword: 256 contiguous bytes per wave instruction straight into LDS, no register, nothing for the
// compiler to spill) into a landing buffer of the wave's own beside its exchange / staging region; it lands while block b is
// transformed and stored, and the wait at the top of the loop is a COUNTED one that leaves block b's output stores in
// flight.  A DMA instruction cannot pad inside its 256 bytes: line strides of >= 256 bytes only (cf32 from hop 32,
// ci16 from hop 64, the byte formats from hop 128; what is below keeps v2n_kernel).  The span starts at the 4-byte boundary
// below the first sample, so a cf32 sample never straddles a pad gap and is one aligned ds_read_b64.  One wave per
// workgroup: the LDS (exchange region + landing buffer, sized per call from the hop) then fits eleven waves per CU at
// 50 % overlap instead of eight with four-wave workgroups.
__device__ __forceinline__ uint32_t v2n_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}
// 4 bytes per lane from gsrc (per lane) to LDS lds_dst + 4 * lane (lds_dst wave-uniform); M0 saved and restored inside the
// statement; the leading lgkmcnt(0): this wave's reads of the landing buffer have left the LDS queue
__device__ __forceinline__ void v2n_glds4(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void v2n_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int L> constexpr int v2n_ex_bytes() {  // exchange buffer of a block's lines / their output staging
    using PL = Plan2<L>;
    constexpr int LB = 64 / PL::T;
    const int ex = LB * PL::LINE * 8, out = LB * (PL::N + 4) * 4;
    return ((ex > out ? ex : out) + 15) & ~15;
}

template <int L, int KIND, bool BE>
__global__ __launch_bounds__(64, V2N_OCC) void v2n_dma_kernel(const V2nArgs a) {
    using PL = Plan2<L>;
    using RW = Raw2<KIND>;
    constexpr int N = PL::N, T = PL::T, E = PL::E, BPS = RW::BPS;
    constexpr int LB = 64 / T, PADU = T * BPS, OUT_STRIDE = N + 4;
    constexpr int STORES = LB * (N / 4) / 64;  // output store instructions of a full block
    static_assert(STORES == 4, "the counted wait below");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    unsigned char *region = smem;                      // exchange, then output staging
    unsigned char *landing = smem + v2n_ex_bytes<L>();  // the next block's input span
    const int t = lane % T, j = lane / T;
    const v2f *__restrict__ tw = static_cast<const v2f *>(a.tw);
    v2f twl[16];
#pragma unroll
    for (int r = 1; r < 16; ++r) twl[r] = tw[(r * t) & (N - 1)];
    const float *win = static_cast<const float *>(a.win);
    float w[E];
    if (win) {
#pragma unroll
        for (int m = 0; m < E; ++m) w[m] = win[t + m * T];
    }
    const uint32_t n_blocks = (a.n_lines + LB - 1) / LB;
    const uint32_t per_wave = (n_blocks + gridDim.x - 1) / gridDim.x;
    // this wave's blocks: chunks of `chunk` consecutive blocks, gridDim.x chunks apart.  All of a wave's blocks in ONE piece (rounds
    // 1-5: the samples two blocks share come back from L2) spreads the waves that run together over the whole recording; chunks of 4
    // keep them on neighbouring blocks: +3 ... +8 % at 64 / 128 points and for the 256-point cells of this kernel, at hop = nfft and
    // at 50 % overlap, 2^28 and 2^30 samples (tools/bench_coop_chunk.py, profiles/r05_run_len.txt; chunks of 1 lose, 2 ... 8 are close)
    const uint32_t want = a.chunk ? a.chunk : 4u, chunk = want < per_wave ? want : per_wave, jump = (gridDim.x - 1) * chunk;
    const uint32_t b0 = blockIdx.x * chunk, b1 = n_blocks;
    if (b0 >= b1) return;
    auto next_block = [&](uint32_t b) -> uint32_t { return (b + 1) % chunk ? b + 1 : b + 1 + jump; };
    const uint32_t line_bytes = a.hop * BPS, psh = a.pad_shift;  // psh >= 8 (the launcher's condition)
    auto padded = [&](uint32_t x) -> uint32_t { return x + (x >> psh) * (uint32_t)PADU; };
    const uint32_t landing_lds = v2n_lds_addr(landing);
    // request block b's span: dword 64 c + lane of it to landing + padded(256 c) + 4 lane
    auto request = [&](uint32_t b) {
        const uint32_t l0 = b * LB;
        const uint32_t nb = a.n_lines - l0 < (uint32_t)LB ? a.n_lines - l0 : (uint32_t)LB;
        const uint8_t *first = a.iq + (uint64_t)l0 * line_bytes;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(first) & 3u);
        const uint32_t dwords = (mis + (nb - 1) * line_bytes + (uint32_t)N * BPS + 3u) / 4u;
        const uint8_t *g = first - mis + 4u * (uint32_t)lane;
        for (uint32_t c = 0; c * 64u < dwords; ++c)
            if (c * 64u + (uint32_t)lane < dwords) v2n_glds4(g + 256u * c, landing_lds + padded(256u * c));
    };
    request(b0);
    bool full_prev = false;  // the previous block issued STORES output stores behind this block's request
    for (uint32_t b = b0; b < b1; b = next_block(b)) {
        const uint32_t l0 = b * LB;
        const uint32_t nb = a.n_lines - l0 < (uint32_t)LB ? a.n_lines - l0 : (uint32_t)LB;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(a.iq + (uint64_t)l0 * line_bytes) & 3u);
        // this block's span has landed; the previous block's output stores (younger) may still be in flight
        if (full_prev) v2n_vm_wait<STORES>();
        else v2n_vm_wait<0>();
        v2_sync<L>();
        v2f v[E];
        const uint32_t base = mis + (uint32_t)j * line_bytes + (uint32_t)t * BPS;
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const unsigned char *p = landing + padded(base + (uint32_t)(m * T * BPS));
            typename RW::type r;
            if constexpr (BPS == 8) r = *reinterpret_cast<const typename RW::type *>(p);  // 8-byte aligned: mis = 0 for cf32
            else if constexpr (BPS == 4) r = *reinterpret_cast<const uint32_t *>(p);
            else r = *reinterpret_cast<const uint16_t *>(p);
            v[m] = RW::dec(BE ? RW::swap(r) : r);  // SMH:87-91 byte order
        }
        if (next_block(b) < b1) request(next_block(b));  // (its lgkmcnt(0): the picks above have been read)
        if (win) {
#pragma unroll
            for (int m = 0; m < E; ++m) v[m] *= v2f{w[m], w[m]};
        }
        v2_fft<L>(v, t, reinterpret_cast<v2f *>(region) + (size_t)j * PL::LINE, static_cast<const v2f *>(nullptr), twl);
        float d[E];
        constexpr bool BOUNDED = KIND != K_CF32;
        v2_epilogue<BOUNDED, E>(v, RW::SCALE, a.out_fmt == OUT_DB20_F32, d);
        v2_sync<L>();  // the exchange has been read: the region becomes the output staging
        float *stage = reinterpret_cast<float *>(region);
#pragma unroll
        for (int m = 0; m < E; ++m) stage[j * OUT_STRIDE + ((t + m * T + N / 2) & (N - 1))] = d[m];  // SS:78
        v2_sync<L>();
        float *dst = a.out + (uint64_t)l0 * N;
        if (nb == (uint32_t)LB) {  // exactly STORES store instructions (the counted wait relies on it)
#pragma unroll
            for (int i = 0; i < STORES; ++i) {
                const uint32_t c = (uint32_t)lane + 64u * i;
                const uint32_t row = c / (uint32_t)(N / 4), col = (c % (uint32_t)(N / 4)) * 4u;
                const vu4 u = *reinterpret_cast<const vu4 *>(stage + row * OUT_STRIDE + col);
                __builtin_nontemporal_store(u, reinterpret_cast<vu4 *>(dst) + c);
            }
            full_prev = true;
        } else {
            const uint32_t out_chunks = nb * (uint32_t)(N / 4);
            for (uint32_t c = (uint32_t)lane; c < out_chunks; c += 64u) {
                const uint32_t row = c / (uint32_t)(N / 4), col = (c % (uint32_t)(N / 4)) * 4u;
                const vu4 u = *reinterpret_cast<const vu4 *>(stage + row * OUT_STRIDE + col);
                __builtin_nontemporal_store(u, reinterpret_cast<vu4 *>(dst) + c);
            }
            full_prev = false;
        }
        v2_sync<L>();  // the staging has been read: the next block's exchange may be written
    }
}

// bytes of the landing buffer for one hop: the padded span at its worst misalignment
template <int L, int BPS> uint32_t v2n_landing_bytes(uint32_t hop, uint32_t psh) {
    using PL = Plan2<L>;
    constexpr uint32_t LB = 64 / PL::T, PADU = PL::T * BPS;
    const uint32_t span = 3u + (LB - 1) * hop * BPS + (uint32_t)PL::N * BPS;
    const uint32_t padded = ((span + 255u) & ~255u) + ((span >> psh) + 2u) * PADU;  // whole 256-byte pieces land
    return (padded + 15u) & ~15u;
}

template <int L, int KIND, bool BE> hipError_t v2n_dma_launch(V2nArgs a, int n_cu, hipStream_t s) {
    using PL = Plan2<L>;
    using RW = Raw2<KIND>;
    constexpr int LB = 64 / PL::T;
    a.in_bytes = v2n_landing_bytes<L, RW::BPS>(a.hop, a.pad_shift);
    const size_t lds = (size_t)v2n_ex_bytes<L>() + a.in_bytes;
    auto kern = v2n_dma_kernel<L, KIND, BE>;
    const uint32_t n_blocks = (a.n_lines + LB - 1) / LB;
    uint32_t wgs = n_blocks;  // one wave each; enough to fill the chip a few times over, consecutive blocks per wave
    const uint32_t cap = (uint32_t)n_cu * 64u;
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64), lds, s, a);
    return hipGetLastError();
}

template <int L, int KIND, bool BE> hipError_t v2n_launch(const V2nArgs &a, int n_cu, hipStream_t s) {
    using PL = Plan2<L>;
    using RW = Raw2<KIND>;
    constexpr int T = PL::T, LB = 64 / T;
    constexpr int REGION = v2n_region_bytes<L, RW::BPS>();
    constexpr size_t lds = (size_t)4 * REGION;
    auto kern = v2n_kernel<L, KIND, BE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t n_blocks = (a.n_lines + LB - 1) / LB;
    // enough waves to fill the chip a few times over, consecutive blocks per wave
    uint32_t wgs = (n_blocks + 3) / 4;
    const uint32_t cap = (uint32_t)n_cu * 16u;
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int L, int KIND> hipError_t v2n_launch_be(const V2nArgs &a, int be, int n_cu, hipStream_t s) {
    const bool dma = V2N_DMA && a.pad_shift >= 8;  // line stride >= 256 bytes (v2n_dma_kernel)
    if constexpr (KIND == K_CF32 || KIND == K_CI16) {
        if (be) return dma ? v2n_dma_launch<L, KIND, true>(a, n_cu, s) : v2n_launch<L, KIND, true>(a, n_cu, s);
    }
    return dma ? v2n_dma_launch<L, KIND, false>(a, n_cu, s) : v2n_launch<L, KIND, false>(a, n_cu, s);
}

template <int L> hipError_t v2n_launch_kind(const V2nArgs &a, int kind, int be, int n_cu, hipStream_t s) {
    switch (kind) {
    case K_CF32: return v2n_launch_be<L, K_CF32>(a, be, n_cu, s);
    case K_CI16: return v2n_launch_be<L, K_CI16>(a, be, n_cu, s);
    case K_CU8: return v2n_launch_be<L, K_CU8>(a, be, n_cu, s);
    case K_CI8: return v2n_launch_be<L, K_CI8>(a, be, n_cu, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

// 64 / 128 points, fp32 outputs, hop between 16 bytes' worth of samples and N, input aligned to min(4, bytes per sample)
bool v2n_applicable(int log2n, int kind, int out_fmt, uint64_t n_lines, uint32_t hop, const void *first, int max_log2n) {
    if (log2n < 6 || log2n > max_log2n || log2n > 8) return false;
    if (kind != K_CF32 && kind != K_CI16 && kind != K_CU8 && kind != K_CI8) return false;
    if (out_fmt != OUT_DB20_F32 && out_fmt != OUT_POW_F32) return false;
    const uint32_t bps = kind == K_CF32 ? 8u : kind == K_CI16 ? 4u : 2u;
    if (hop > (1u << log2n) || (uint64_t)hop * bps < 16u) return false;
    if (reinterpret_cast<uintptr_t>(first) % (bps < 4u ? bps : 4u)) return false;
    return n_lines > 0 && n_lines < (1ull << 31);
}

hipError_t launch_v2n_spectro(const WfArgs &w, int log2n, int n_cu, hipStream_t s) {
    V2nArgs a{};
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.tw = w.tw; a.win = w.win;
    a.out = static_cast<float *>(w.out); a.out_fmt = w.out_fmt;
    a.chunk = w.lines_per_wg;  // (blocks; 0 = automatic)
    uint32_t lb = w.hop * w.bps, sh = 0;
    while ((2u << sh) <= lb) ++sh;
    a.pad_shift = sh;
    return log2n == 6 ? v2n_launch_kind<6>(a, w.kind, w.be, n_cu, s) : log2n == 7 ? v2n_launch_kind<7>(a, w.kind, w.be, n_cu, s)
                                                                                    : v2n_launch_kind<8>(a, w.kind, w.be, n_cu, s);
}

}  // namespace specgpu
