// spec_v2h.h -- pieces shared by the single-workgroup kernels for lines longer than the LDS (spec_k_v2h.hip: 8192 ... 32768
// points as two transforms of half the length; spec_k_v2q.hip: 65536 points as a PAIR of workgroups, two transforms of a
// quarter of the length each): the per-register twiddle constants of the radix-2 step and the dB epilogue.
#pragma once
#include "spec_v2.h"

namespace specgpu {

namespace {

// W_64^m = exp(-2 pi i m / 64), m = 0 .. 31
__device__ static constexpr double kW64[32][2] = {
    {1.0, 0.0},
    {0.995184726672196886245, -0.0980171403295606019942},
    {0.980785280403230449126, -0.195090322016128267848},
    {0.956940335732208864936, -0.290284677254462367636},
    {0.923879532511286756128, -0.382683432365089771728},
    {0.881921264348355029713, -0.471396736825997648556},
    {0.831469612302545237079, -0.555570233019602224743},
    {0.773010453362736960811, -0.634393284163645498215},
    {0.707106781186547524401, -0.707106781186547524401},
    {0.634393284163645498215, -0.773010453362736960811},
    {0.555570233019602224743, -0.831469612302545237079},
    {0.471396736825997648556, -0.881921264348355029713},
    {0.382683432365089771728, -0.923879532511286756128},
    {0.290284677254462367636, -0.956940335732208864936},
    {0.195090322016128267848, -0.980785280403230449126},
    {0.0980171403295606019942, -0.995184726672196886245},
    {0.0, -1.0},
    {-0.0980171403295606019942, -0.995184726672196886245},
    {-0.195090322016128267848, -0.980785280403230449126},
    {-0.290284677254462367636, -0.956940335732208864936},
    {-0.382683432365089771728, -0.923879532511286756128},
    {-0.471396736825997648556, -0.881921264348355029713},
    {-0.555570233019602224743, -0.831469612302545237079},
    {-0.634393284163645498215, -0.773010453362736960811},
    {-0.707106781186547524401, -0.707106781186547524401},
    {-0.773010453362736960811, -0.634393284163645498215},
    {-0.831469612302545237079, -0.555570233019602224743},
    {-0.881921264348355029713, -0.471396736825997648556},
    {-0.923879532511286756128, -0.382683432365089771728},
    {-0.956940335732208864936, -0.290284677254462367636},
    {-0.980785280403230449126, -0.195090322016128267848},
    {-0.995184726672196886245, -0.0980171403295606019942}};

// d * W_NT^M * wt, NT = N / T = 64 (32 points per thread and half) or 32 (16 points)
template <int M, int NT> __device__ __forceinline__ v2f v2h_twiddle(v2f d, v2f wt) {
    static_assert(NT == 64 || NT == 32, "per-thread stride of the line");
    if constexpr (M == 0) return pk_cmul(d, wt);
    else if constexpr (M == NT / 4) return pk_cmul(pk_mul_mi(d), wt);
    else if constexpr (NT == 64) return pk_cmul(pk_cmul_const(d, kW64[M][0], kW64[M][1]), wt);
    else return pk_cmul(pk_cmul_const(d, kW32[M][0], kW32[M][1]), wt);
}
template <typename F, int... M> __device__ __forceinline__ void v2h_for_each(F &&f, std::integer_sequence<int, M...>) {
    (f(std::integral_constant<int, M>{}), ...);
}

// ---- epilogue: 20 log10(|X| + 1e-10) (SS:80-81), one range test per thread as in the family (spec_v2.h v2_epilogue) ----
// The family inlines the exact form db20() -- three branches -- once per bin behind the range test.  In this kernel the
// allocator then keeps the spectrum alive through all 32 of them and the HOT path loses 40 registers, exactly the room
// the next line's samples need.  Here the forms behind the fast one are straight-line code (selects, no branches):
//   * from |X|^2 alone while nothing can overflow (every integer format; cf32 with every |X|^2 < 1e37):
//       p > 1e-4 ? 10 log10(p) : 20 log10(sqrt(p) + 1e-10)            (|X| + 1e-10 == |X| in fp32 above that)
//   * cf32 with a huge, infinite or NaN value somewhere in the thread's bins: additionally the rescaled form
//       10 log10((x 2^-64)^2 + (y 2^-64)^2) + 10 log10(2^128)          for p >= 1e37
// The values are db20()'s (spec_fft.h) bin for bin.
template <bool BOUNDED, int E> __device__ __forceinline__ void v2h_epilogue(const v2f (&v)[E], float scale, bool db, float (&d)[E]) {
    float p[E];
#pragma unroll
    for (int m = 0; m < E; ++m) p[m] = pk_norm(v[m]);
    const float s2 = scale * scale;  // a power of two for the integer formats, 1 for cf32: p s2 is exact
    if (!db) {
#pragma unroll
        for (int m = 0; m < E; ++m) d[m] = p[m] * s2;
        return;
    }
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
    constexpr float k10 = 3.01029995663981195f;  // 10 log10(2)
    const float off = k10 * __log2f(s2);
    if (lo * s2 > 1e-4f && hi < 1e37f) {  // a NaN fails the first test
#pragma unroll
        for (int m = 0; m < E; m += 2) {  // two bins per v_pk_fma_f32
            const v2f r = __builtin_elementwise_fma(v2f{__log2f(p[m]), __log2f(p[m + 1])}, v2f{k10, k10}, v2f{off, off});
            d[m] = r.x;
            d[m + 1] = r.y;
        }
        return;
    }
    bool from_p = true;
    if constexpr (!BOUNDED) from_p = hi < 1e37f;  // (fmaxf drops NaNs: a NaN bin is carried by the select below)
    if (from_p) {
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const float ps = p[m] * s2;
            d[m] = ps > 1e-4f ? k10 * __log2f(ps) : 2.0f * k10 * __log2f(sqrtf(ps) + 1e-10f);
        }
    } else {
        if constexpr (!BOUNDED) {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const float ps = p[m];
                const float xs = v[m].x * (1.0f / 1.8446744e19f), ys = v[m].y * (1.0f / 1.8446744e19f);  // 2^-64
                const float r3 = k10 * (__log2f(__builtin_fmaf(xs, xs, ys * ys)) + 128.0f);
                const float r12 = ps > 1e-4f ? k10 * __log2f(ps) : 2.0f * k10 * __log2f(sqrtf(ps) + 1e-10f);
                d[m] = ps >= 1e37f ? r3 : r12;
            }
        }
    }
}

}  // namespace

}  // namespace specgpu
