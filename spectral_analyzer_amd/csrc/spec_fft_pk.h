// spec_fft_pk.h -- fp32 complex arithmetic written on 2-wide vectors so that one
// complex add / twiddle half is ONE packed VALU instruction (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32 with op_sel / neg modifiers).
//
// Why: on gfx950 a VALU wave-instruction holds its SIMD for 4 cycles whether it
// is scalar or packed (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.1
// cycles on the scalar-source build, profiles/r01), and the 4096-point
// spectrogram kernel is VALU-issue bound before it is HBM bound.  A complex
// value therefore lives in an aligned VGPR pair from the load to the epilogue
// and no lane shuffling (v_mov) is spent on packing.
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

namespace specgpu {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_swap(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }

// b + a*(-i) and b - a*(-i).  Written as an FMA with the constant (1,-1) on the
// swapped operand: hipcc folds the swap into op_sel and keeps the constant in an
// SGPR pair, so each is ONE v_pk_fma_f32; written as b + (a.y, -a.x) it spends a
// v_xor and a v_mov on the per-lane negation first.
__device__ __forceinline__ v2f pk_add_mi(v2f b, v2f a) { return __builtin_elementwise_fma(pk_swap(a), v2f{1.0f, -1.0f}, b); }
__device__ __forceinline__ v2f pk_sub_mi(v2f b, v2f a) { return __builtin_elementwise_fma(pk_swap(a), v2f{-1.0f, 1.0f}, b); }

// a * w for a run-time twiddle w = (c, d):  (a.x, a.y)*(c, c) + (a.y, a.x)*(-d, d).
// Two instructions: the swap is op_sel and the per-lane negation the neg_lo modifier of
// v_pk_fma_f32 -- hipcc does not select neg_lo from IR (it builds (-d, d) with an extra
// v_pk_mul or v_xor/v_mov), hence the asm; plain "v" operands, no side effects, so the
// scheduler and the waitcnt insertion treat both like any other VALU instruction.
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f w) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(r));
    return r;
}
// a * (c - i s) for compile-time constants
__device__ __forceinline__ v2f pk_cmul_const(v2f a, float c, float d) {
    return __builtin_elementwise_fma(pk_swap(a), v2f{-d, d}, a * v2f{c, c});
}

__device__ __forceinline__ void pk_dft4(v2f &x0, v2f &x1, v2f &x2, v2f &x3) {
    const v2f t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3, d = x1 - x3;
    x0 = t0 + t2;
    x2 = t0 - t2;
    x1 = pk_add_mi(t1, d);   // t1 + (-i) d
    x3 = pk_sub_mi(t1, d);   // t1 - (-i) d
}
// |a|^2 with scalar mul + fma: two instructions per bin.  (Left as v2f arithmetic hipcc packs two
// bins per v_pk_mul_f32 and then spends four v_mov on the transposition: 3.5 per bin.)
__device__ __forceinline__ float pk_norm(v2f a) { return __builtin_fmaf(a.x, a.x, a.y * a.y); }

__device__ __forceinline__ void pk_dft2(v2f &a, v2f &b) {
    const v2f t = a - b;
    a = a + b;
    b = t;
}

// 8-point forward DFT in place, natural order out: n = 2 n1 + n2, k = k1 + 4 k2
__device__ __forceinline__ void pk_dft8(v2f (&u)[8]) {
    constexpr float h = 0.70710678118654752440f;
    pk_dft4(u[0], u[2], u[4], u[6]);  // A0[k1] -> slots 0,2,4,6
    pk_dft4(u[1], u[3], u[5], u[7]);  // A1[k1] -> slots 1,3,5,7
    // A1[1] W8^1 = h(1 - i) A1[1] and A1[3] W8^3 = -h(1 + i) A1[3]: the factor h rides in the FMA of the
    // final additions instead of a multiply of its own
    const v2f q1 = pk_add_mi(u[3], u[3]);                // (1 - i) A1[1]
    const v2f q3 = pk_sub_mi(u[7], u[7]);                // (1 + i) A1[3]
    v2f y[8];
    y[0] = u[0] + u[1];
    y[4] = u[0] - u[1];
    y[1] = __builtin_elementwise_fma(q1, v2f{h, h}, u[2]);
    y[5] = __builtin_elementwise_fma(q1, v2f{-h, -h}, u[2]);
    y[2] = pk_add_mi(u[4], u[5]);                        // A1[2] W8^2 = -i folded in
    y[6] = pk_sub_mi(u[4], u[5]);
    y[3] = __builtin_elementwise_fma(q3, v2f{-h, -h}, u[6]);
    y[7] = __builtin_elementwise_fma(q3, v2f{h, h}, u[6]);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = y[k];
}

// 16-point forward DFT in place, natural order out (same index algebra as
// dft16 in spec_fft.h)
__device__ __forceinline__ void pk_dft16(v2f (&u)[16]) {
    constexpr float h = 0.70710678118654752440f;
    constexpr float c1 = 0.92387953251128675613f;  // cos(pi/8)
    constexpr float s1 = 0.38268343236508977173f;  // sin(pi/8)
    pk_dft4(u[0], u[4], u[8], u[12]);
    pk_dft4(u[1], u[5], u[9], u[13]);
    pk_dft4(u[2], u[6], u[10], u[14]);
    pk_dft4(u[3], u[7], u[11], u[15]);
    // slot 4*k1 + n2 holds A[n2][k1]; multiply by W16^(n2*k1).  The four factors of modulus h,
    // W16^2 = h(1 - i) and W16^6 = -h(1 + i), are applied as (1 -+ i) here and as +-h inside the FMAs of
    // the second-stage butterflies: no multiply of their own
    u[5] = pk_cmul_const(u[5], c1, -s1);                  // W16^1 = c1 - i s1
    const v2f q6 = pk_add_mi(u[6], u[6]);                 // (1 - i) u6,  u6 W16^2 = h q6
    u[7] = pk_cmul_const(u[7], s1, -c1);                  // W16^3 = s1 - i c1
    const v2f q9 = pk_add_mi(u[9], u[9]);                 // u9 W16^2 = h q9
    /* u[10] *= W16^4 = -i : folded into the third butterfly below */
    const v2f q11 = pk_sub_mi(u[11], u[11]);              // (1 + i) u11, u11 W16^6 = -h q11
    u[13] = pk_cmul_const(u[13], s1, -c1);                // W16^3
    const v2f q14 = pk_sub_mi(u[14], u[14]);              // u14 W16^6 = -h q14
    u[15] = pk_cmul_const(u[15], -c1, s1);                // W16^9 = -W16^1
    constexpr v2f ph{h, h}, mh{-h, -h};
    pk_dft4(u[0], u[1], u[2], u[3]);
    {   // dft4(u4, u5, h q6, u7)
        const v2f t0 = __builtin_elementwise_fma(q6, ph, u[4]), t1 = __builtin_elementwise_fma(q6, mh, u[4]);
        const v2f t2 = u[5] + u[7], d = u[5] - u[7];
        u[4] = t0 + t2;
        u[6] = t0 - t2;
        u[5] = pk_add_mi(t1, d);
        u[7] = pk_sub_mi(t1, d);
    }
    {   // dft4(u8, h q9, -i u10, -h q11):  x1 + x3 = h (q9 - q11),  x1 - x3 = h (q9 + q11)
        const v2f t0 = pk_add_mi(u[8], u[10]), t1 = pk_sub_mi(u[8], u[10]);
        const v2f sm = q9 - q11, sp = q9 + q11;
        u[8] = __builtin_elementwise_fma(sm, ph, t0);
        u[10] = __builtin_elementwise_fma(sm, mh, t0);
        u[9] = __builtin_elementwise_fma(pk_swap(sp), v2f{h, -h}, t1);    // t1 + (-i) h sp
        u[11] = __builtin_elementwise_fma(pk_swap(sp), v2f{-h, h}, t1);   // t1 - (-i) h sp
    }
    {   // dft4(u12, u13, -h q14, u15)
        const v2f t0 = __builtin_elementwise_fma(q14, mh, u[12]), t1 = __builtin_elementwise_fma(q14, ph, u[12]);
        const v2f t2 = u[13] + u[15], d = u[13] - u[15];
        u[12] = t0 + t2;
        u[14] = t0 - t2;
        u[13] = pk_add_mi(t1, d);
        u[15] = pk_sub_mi(t1, d);
    }
    // slot 4*k1 + k2 holds X[k1 + 4*k2]: transpose to natural order (pure renaming)
    v2f y[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) y[k1 + 4 * k2] = u[4 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = y[k];
}

// W_32^r = exp(-2 pi i r / 32), r = 0 .. 15
__device__ static constexpr float kW32[16][2] = {
    {1.00000000000000000000f, -0.00000000000000000000f}, {0.98078528040323043058f, -0.19509032201612824808f},
    {0.92387953251128673848f, -0.38268343236508978178f}, {0.83146961230254523567f, -0.55557023301960217765f},
    {0.70710678118654757274f, -0.70710678118654746172f}, {0.55557023301960228867f, -0.83146961230254523567f},
    {0.38268343236508983729f, -0.92387953251128673848f}, {0.19509032201612833135f, -0.98078528040323043058f},
    {0.00000000000000000000f, -1.00000000000000000000f}, {-0.19509032201612819257f, -0.98078528040323043058f},
    {-0.38268343236508972627f, -0.92387953251128673848f}, {-0.55557023301960195560f, -0.83146961230254545772f},
    {-0.70710678118654746172f, -0.70710678118654757274f}, {-0.83146961230254534669f, -0.55557023301960217765f},
    {-0.92387953251128673848f, -0.38268343236508989280f}, {-0.98078528040323043058f, -0.19509032201612860891f}};

// a * (-i)
__device__ __forceinline__ v2f pk_mul_mi(v2f a) { return pk_swap(a) * v2f{1.0f, -1.0f}; }

// a * W_32^R for a compile-time R
template <int R> __device__ __forceinline__ v2f pk_mul_w32(v2f a) {
    if constexpr (R == 0) return a;
    else if constexpr (R == 8) return pk_mul_mi(a);
    else return pk_cmul_const(a, kW32[R][0], kW32[R][1]);
}

// 32-point forward DFT in place, natural order out: even / odd halves (two 16-point DFTs), then
// X[k] = A0[k] + W_32^k A1[k],  X[k + 16] = A0[k] - W_32^k A1[k]
template <int K> __device__ __forceinline__ void pk_dft32_comb(v2f (&u)[32], const v2f (&a0)[16], const v2f (&a1)[16]) {
    if constexpr (K == 8) {  // W_32^8 = -i: folded into the additions
        u[K] = pk_add_mi(a0[K], a1[K]);
        u[K + 16] = pk_sub_mi(a0[K], a1[K]);
    } else {
        const v2f w = pk_mul_w32<K>(a1[K]);
        u[K] = a0[K] + w;
        u[K + 16] = a0[K] - w;
    }
}
template <int... K>
__device__ __forceinline__ void pk_dft32_comb_all(v2f (&u)[32], const v2f (&a0)[16], const v2f (&a1)[16],
                                                  std::integer_sequence<int, K...>) {
    (pk_dft32_comb<K>(u, a0, a1), ...);
}
__device__ __forceinline__ void pk_dft32(v2f (&u)[32]) {
    v2f a0[16], a1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { a0[j] = u[2 * j]; a1[j] = u[2 * j + 1]; }
    pk_dft16(a0);
    pk_dft16(a1);
    pk_dft32_comb_all(u, a0, a1, std::make_integer_sequence<int, 16>{});
}

template <int RADIX> __device__ __forceinline__ void pk_dft(v2f (&u)[RADIX]) {
    if constexpr (RADIX == 2) pk_dft2(u[0], u[1]);
    else if constexpr (RADIX == 4) pk_dft4(u[0], u[1], u[2], u[3]);
    else if constexpr (RADIX == 8) pk_dft8(u);
    else if constexpr (RADIX == 16) pk_dft16(u);
    else pk_dft32(u);
}

}  // namespace specgpu
