// spec_kernels.h -- the generic spectrogram kernel (any nfft in the plan table, any
// datatype, either byte order, fp32 or fp64), instantiated per precision in
// spec_k_f32.hip and spec_k_f64.hip.  The packed-fp32 product path is spec_v2.h.
#pragma once
#include "spec_fft.h"
#include "spec_internal.h"

namespace specgpu {

// Load the E samples of one line owned by thread t:  v[m] = x[t + m*T].
template <typename R, int LOG2N>
__device__ __forceinline__ void load_line(cx<R> (&v)[Plan<LOG2N>::E], const uint8_t *__restrict__ line0,
                                          int t, uint32_t bps, int kind, bool be) {
    using PL = Plan<LOG2N>;
#pragma unroll
    for (int m = 0; m < PL::E; ++m)
        v[m] = decode_sample<R>(line0 + (uint64_t)(t + m * PL::T) * bps, kind, be);
}

template <typename R> __device__ __forceinline__ void store_bin(void *out, uint64_t idx, cx<R> z, int fmt) {
    if constexpr (sizeof(R) == 4) {
        if (fmt == OUT_DB20_F32) static_cast<float *>(out)[idx] = db20(z);
        else static_cast<float *>(out)[idx] = z.x * z.x + z.y * z.y;
    } else {
        switch (fmt) {
        case OUT_DB20_F32: static_cast<float *>(out)[idx] = (float)db20(z); break;
        case OUT_POW_F32: static_cast<float *>(out)[idx] = (float)(z.x * z.x + z.y * z.y); break;
        case OUT_DB20_F64: static_cast<double *>(out)[idx] = db20(z); break;
        default: static_cast<double *>(out)[idx] = z.x * z.x + z.y * z.y; break;
        }
    }
}

// Spectrogram: MainController.java:980-999 (line loop) around
// SpectralService.java:33-85 (one line).  A workgroup owns `lines_per_wg`
// consecutive lines and walks them LPW at a time, so overlapping spans
// (hop < nfft) are re-read from this CU's L1 / the XCD's L2, not from HBM.
// fp64: two workgroups per CU wherever the LDS allows it -- left alone the allocator takes 256 VGPRs plus
// up to 112 AGPRs for the sixteen inlined dB epilogues and halves the residency
template <typename R, int LOG2N> constexpr int spectro_min_waves() {
    return sizeof(R) == 8 && Plan<LOG2N>::WG == 256 && (size_t)Plan<LOG2N>::LPW * Plan<LOG2N>::N * sizeof(cx<R>) <= 80 * 1024 ? 2 : 1;
}

template <typename R, int LOG2N>
__global__ __launch_bounds__(Plan<LOG2N>::WG, (spectro_min_waves<R, LOG2N>())) void spectro_kernel(const WfArgs a) {
    using PL = Plan<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, t = tid % PL::T, q = tid / PL::T;
    cx<R> *lds = reinterpret_cast<cx<R> *>(smem) + (size_t)q * PL::N;
    const cx<R> *__restrict__ tw = static_cast<const cx<R> *>(a.tw);
    const R *__restrict__ win = static_cast<const R *>(a.win);

    R w[PL::E];
    if (win) {
#pragma unroll
        for (int m = 0; m < PL::E; ++m) w[m] = win[t + m * PL::T];
    }
    const uint64_t first = (uint64_t)blockIdx.x * a.lines_per_wg;
    uint64_t last = first + a.lines_per_wg;
    if (last > a.n_lines) last = a.n_lines;
    const uint64_t line_bytes = (uint64_t)a.hop * a.bps;

    for (uint64_t g = first; g < last; g += PL::LPW) {
        const uint64_t line = g + q;
        const bool active = line < last;
        const uint64_t lc = active ? line : last - 1;  // idle sub-lines redo the last one, store nothing
        cx<R> v[PL::E];
        load_line<R, LOG2N>(v, a.iq + lc * line_bytes, t, a.bps, a.kind, a.be != 0);
        if (win) {
#pragma unroll
            for (int m = 0; m < PL::E; ++m) { v[m].x *= w[m]; v[m].y *= w[m]; }
        }
        fft_line<R, LOG2N>(v, t, lds, tw);
        if (active) {
            const uint64_t base = line * (uint64_t)PL::N;
#pragma unroll
            for (int m = 0; m < PL::E; ++m) {  // fftshift: (k + N/2) mod N   (SS:78)
                store_bin<R>(a.out, base + (uint64_t)((t + m * PL::T + PL::N / 2) & (PL::N - 1)), v[m], a.out_fmt);
                // fp64: finish one bin's sqrt / log series before starting the next (the scheduler would
                // interleave all sixteen and need their temporaries at once)
                if constexpr (sizeof(R) == 8) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <typename R, int LOG2N> hipError_t launch_spectro_one(const WfArgs &a, hipStream_t s) {
    using PL = Plan<LOG2N>;
    const size_t lds = (size_t)PL::LPW * PL::N * sizeof(cx<R>);
    if (lds > 64 * 1024) {  // per device, so set on every launch of the big-LDS plans
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&spectro_kernel<R, LOG2N>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint64_t n_wg = (a.n_lines + a.lines_per_wg - 1) / a.lines_per_wg;
    hipLaunchKernelGGL((spectro_kernel<R, LOG2N>), dim3((unsigned)n_wg), dim3(PL::WG), lds, s, a);
    return hipGetLastError();
}

}  // namespace specgpu
