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
// The same algebra in fp64 (the strict-parity pipeline): a complex value is a 2-vector of doubles; there are
// no packed fp64 instructions, the compiler emits two scalar operations per vector operation.  Every
// function below is a template on the vector type V (v2f or v2d); the v2f instantiations are the packed code.
typedef double v2d __attribute__((ext_vector_type(2)));
template <typename V> struct pk_scalar;
template <> struct pk_scalar<v2f> { using type = float; };
template <> struct pk_scalar<v2d> { using type = double; };
template <typename V> using pk_scalar_t = typename pk_scalar<V>::type;
template <typename V> __device__ __forceinline__ constexpr V pk_make(double x, double y) {
    return V{(pk_scalar_t<V>)x, (pk_scalar_t<V>)y};
}

template <typename V> __device__ __forceinline__ V pk_swap(V a) { return __builtin_shufflevector(a, a, 1, 0); }

// b + a*(-i) and b - a*(-i).  Written as an FMA with the constant (1,-1) on the
// swapped operand: hipcc folds the swap into op_sel and keeps the constant in an
// SGPR pair, so each is ONE v_pk_fma_f32; written as b + (a.y, -a.x) it spends a
// v_xor and a v_mov on the per-lane negation first.
template <typename V> __device__ __forceinline__ V pk_add_mi(V b, V a) { return __builtin_elementwise_fma(pk_swap(a), pk_make<V>(1.0, -1.0), b); }
template <typename V> __device__ __forceinline__ V pk_sub_mi(V b, V a) { return __builtin_elementwise_fma(pk_swap(a), pk_make<V>(-1.0, 1.0), b); }

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
// fp64: four scalar operations
__device__ __forceinline__ v2d pk_cmul(v2d a, v2d w) {
    return v2d{__builtin_fma(a.x, w.x, -(a.y * w.y)), __builtin_fma(a.y, w.x, a.x * w.y)};
}
// a * (c + i d) for compile-time constants (given in double, rounded once to V's precision)
template <typename V> __device__ __forceinline__ V pk_cmul_const(V a, double c, double d) {
    return __builtin_elementwise_fma(pk_swap(a), pk_make<V>(-d, d), a * pk_make<V>(c, c));
}

template <typename V> __device__ __forceinline__ void pk_dft4(V &x0, V &x1, V &x2, V &x3) {
    const V t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3, d = x1 - x3;
    x0 = t0 + t2;
    x2 = t0 - t2;
    x1 = pk_add_mi(t1, d);   // t1 + (-i) d
    x3 = pk_sub_mi(t1, d);   // t1 - (-i) d
}
// |a|^2 = fma(x, x, y y) with scalar mul + fma: two instructions per bin.  Spelled as instructions since round 5: written as
// __builtin_fmaf(a.x, a.x, a.y * a.y) the SLP vectorizer of ROCm 7.2's hipcc pairs two BINS into one v_pk_mul_f32 + v_pk_fma_f32
// and spends four v_mov on gathering their x and y halves -- three instructions per bin instead of two (16 of the 420 vector
// instructions of a 4096-point line, 32 of a 32-point thread's).  Same two roundings, bit-identical results.
#ifndef SPEC_PK_NORM_ASM
#define SPEC_PK_NORM_ASM 1  // 0: the compiler's form (build.py --variant slpnorm)
#endif
__device__ __forceinline__ float pk_norm(v2f a) {
#if SPEC_PK_NORM_ASM
    float r;
    asm("v_mul_f32_e32 %0, %1, %1" : "=v"(r) : "v"(a.y));
    asm("v_fmac_f32_e32 %0, %1, %1" : "+v"(r) : "v"(a.x));
    return r;
#else
    return __builtin_fmaf(a.x, a.x, a.y * a.y);
#endif
}
// acc + |a|^2 = fma(x, x, fma(y, y, acc)): the Welch sums (two chained FMAs per point; the compiler's form costs the same v_mov pairs)
__device__ __forceinline__ float pk_norm_acc(v2f a, float acc) {
#if SPEC_PK_NORM_ASM
    asm("v_fmac_f32_e32 %0, %1, %1" : "+v"(acc) : "v"(a.y));
    asm("v_fmac_f32_e32 %0, %1, %1" : "+v"(acc) : "v"(a.x));
    return acc;
#else
    return __builtin_fmaf(a.x, a.x, __builtin_fmaf(a.y, a.y, acc));
#endif
}
__device__ __forceinline__ double pk_norm(v2d a) { return __builtin_fma(a.x, a.x, a.y * a.y); }

template <typename V> __device__ __forceinline__ void pk_dft2(V &a, V &b) {
    const V t = a - b;
    a = a + b;
    b = t;
}

// 8-point forward DFT in place, natural order out: n = 2 n1 + n2, k = k1 + 4 k2
template <typename V> __device__ __forceinline__ void pk_dft8(V (&u)[8]) {
    constexpr double h = 0.70710678118654752440;
    pk_dft4(u[0], u[2], u[4], u[6]);  // A0[k1] -> slots 0,2,4,6
    pk_dft4(u[1], u[3], u[5], u[7]);  // A1[k1] -> slots 1,3,5,7
    // A1[1] W8^1 = h(1 - i) A1[1] and A1[3] W8^3 = -h(1 + i) A1[3]: the factor h rides in the FMA of the
    // final additions instead of a multiply of its own
    const V q1 = pk_add_mi(u[3], u[3]);                  // (1 - i) A1[1]
    const V q3 = pk_sub_mi(u[7], u[7]);                  // (1 + i) A1[3]
    V y[8];
    y[0] = u[0] + u[1];
    y[4] = u[0] - u[1];
    y[1] = __builtin_elementwise_fma(q1, pk_make<V>(h, h), u[2]);
    y[5] = __builtin_elementwise_fma(q1, pk_make<V>(-h, -h), u[2]);
    y[2] = pk_add_mi(u[4], u[5]);                        // A1[2] W8^2 = -i folded in
    y[6] = pk_sub_mi(u[4], u[5]);
    y[3] = __builtin_elementwise_fma(q3, pk_make<V>(-h, -h), u[6]);
    y[7] = __builtin_elementwise_fma(q3, pk_make<V>(h, h), u[6]);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = y[k];
}

// 16-point forward DFT in place, natural order out (same index algebra as
// dft16 in spec_fft.h)
template <typename V> __device__ __forceinline__ void pk_dft16(V (&u)[16]) {
    constexpr double h = 0.70710678118654752440;
    constexpr double c1 = 0.92387953251128675613;  // cos(pi/8)
    constexpr double s1 = 0.38268343236508977173;  // sin(pi/8)
    pk_dft4(u[0], u[4], u[8], u[12]);
    pk_dft4(u[1], u[5], u[9], u[13]);
    pk_dft4(u[2], u[6], u[10], u[14]);
    pk_dft4(u[3], u[7], u[11], u[15]);
    // slot 4*k1 + n2 holds A[n2][k1]; multiply by W16^(n2*k1).  The four factors of modulus h,
    // W16^2 = h(1 - i) and W16^6 = -h(1 + i), are applied as (1 -+ i) here and as +-h inside the FMAs of
    // the second-stage butterflies: no multiply of their own
    u[5] = pk_cmul_const(u[5], c1, -s1);                  // W16^1 = c1 - i s1
    const V q6 = pk_add_mi(u[6], u[6]);                 // (1 - i) u6,  u6 W16^2 = h q6
    u[7] = pk_cmul_const(u[7], s1, -c1);                  // W16^3 = s1 - i c1
    const V q9 = pk_add_mi(u[9], u[9]);                 // u9 W16^2 = h q9
    /* u[10] *= W16^4 = -i : folded into the third butterfly below */
    const V q11 = pk_sub_mi(u[11], u[11]);              // (1 + i) u11, u11 W16^6 = -h q11
    u[13] = pk_cmul_const(u[13], s1, -c1);                // W16^3
    const V q14 = pk_sub_mi(u[14], u[14]);              // u14 W16^6 = -h q14
    u[15] = pk_cmul_const(u[15], -c1, s1);                // W16^9 = -W16^1
    const V ph = pk_make<V>(h, h), mh = pk_make<V>(-h, -h);
    pk_dft4(u[0], u[1], u[2], u[3]);
    {   // dft4(u4, u5, h q6, u7)
        const V t0 = __builtin_elementwise_fma(q6, ph, u[4]), t1 = __builtin_elementwise_fma(q6, mh, u[4]);
        const V t2 = u[5] + u[7], d = u[5] - u[7];
        u[4] = t0 + t2;
        u[6] = t0 - t2;
        u[5] = pk_add_mi(t1, d);
        u[7] = pk_sub_mi(t1, d);
    }
    {   // dft4(u8, h q9, -i u10, -h q11):  x1 + x3 = h (q9 - q11),  x1 - x3 = h (q9 + q11)
        const V t0 = pk_add_mi(u[8], u[10]), t1 = pk_sub_mi(u[8], u[10]);
        const V sm = q9 - q11, sp = q9 + q11;
        u[8] = __builtin_elementwise_fma(sm, ph, t0);
        u[10] = __builtin_elementwise_fma(sm, mh, t0);
        u[9] = __builtin_elementwise_fma(pk_swap(sp), pk_make<V>(h, -h), t1);    // t1 + (-i) h sp
        u[11] = __builtin_elementwise_fma(pk_swap(sp), pk_make<V>(-h, h), t1);   // t1 - (-i) h sp
    }
    {   // dft4(u12, u13, -h q14, u15)
        const V t0 = __builtin_elementwise_fma(q14, mh, u[12]), t1 = __builtin_elementwise_fma(q14, ph, u[12]);
        const V t2 = u[13] + u[15], d = u[13] - u[15];
        u[12] = t0 + t2;
        u[14] = t0 - t2;
        u[13] = pk_add_mi(t1, d);
        u[15] = pk_sub_mi(t1, d);
    }
    // slot 4*k1 + k2 holds X[k1 + 4*k2]: transpose to natural order (pure renaming)
    V y[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) y[k1 + 4 * k2] = u[4 * k1 + k2];
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = y[k];
}

// W_32^r = exp(-2 pi i r / 32), r = 0 .. 15
__device__ static constexpr double kW32[16][2] = {
    {1.00000000000000000000, -0.00000000000000000000}, {0.98078528040323043058, -0.19509032201612824808},
    {0.92387953251128673848, -0.38268343236508978178}, {0.83146961230254523567, -0.55557023301960217765},
    {0.70710678118654757274, -0.70710678118654746172}, {0.55557023301960228867, -0.83146961230254523567},
    {0.38268343236508983729, -0.92387953251128673848}, {0.19509032201612833135, -0.98078528040323043058},
    {0.00000000000000000000, -1.00000000000000000000}, {-0.19509032201612819257, -0.98078528040323043058},
    {-0.38268343236508972627, -0.92387953251128673848}, {-0.55557023301960195560, -0.83146961230254545772},
    {-0.70710678118654746172, -0.70710678118654757274}, {-0.83146961230254534669, -0.55557023301960217765},
    {-0.92387953251128673848, -0.38268343236508989280}, {-0.98078528040323043058, -0.19509032201612860891}};

// a * (-i)
template <typename V> __device__ __forceinline__ V pk_mul_mi(V a) { return pk_swap(a) * pk_make<V>(1.0, -1.0); }

// a * W_32^R for a compile-time R
template <int R, typename V> __device__ __forceinline__ V pk_mul_w32(V a) {
    if constexpr (R == 0) return a;
    else if constexpr (R == 8) return pk_mul_mi(a);
    else return pk_cmul_const(a, kW32[R][0], kW32[R][1]);
}

// 32-point forward DFT in place, natural order out: even / odd halves (two 16-point DFTs), then
// X[k] = A0[k] + W_32^k A1[k],  X[k + 16] = A0[k] - W_32^k A1[k]
template <int K, typename V> __device__ __forceinline__ void pk_dft32_comb(V (&u)[32], const V (&a0)[16], const V (&a1)[16]) {
    if constexpr (K == 8) {  // W_32^8 = -i: folded into the additions
        u[K] = pk_add_mi(a0[K], a1[K]);
        u[K + 16] = pk_sub_mi(a0[K], a1[K]);
    } else {
        const V w = pk_mul_w32<K>(a1[K]);
        u[K] = a0[K] + w;
        u[K + 16] = a0[K] - w;
    }
}
template <typename V, int... K>
__device__ __forceinline__ void pk_dft32_comb_all(V (&u)[32], const V (&a0)[16], const V (&a1)[16],
                                                  std::integer_sequence<int, K...>) {
    (pk_dft32_comb<K>(u, a0, a1), ...);
}
template <typename V> __device__ __forceinline__ void pk_dft32(V (&u)[32]) {
    V a0[16], a1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { a0[j] = u[2 * j]; a1[j] = u[2 * j + 1]; }
    pk_dft16(a0);
    pk_dft16(a1);
    pk_dft32_comb_all(u, a0, a1, std::make_integer_sequence<int, 16>{});
}

template <int RADIX, typename V> __device__ __forceinline__ void pk_dft(V (&u)[RADIX]) {
    if constexpr (RADIX == 2) pk_dft2(u[0], u[1]);
    else if constexpr (RADIX == 4) pk_dft4(u[0], u[1], u[2], u[3]);
    else if constexpr (RADIX == 8) pk_dft8(u);
    else if constexpr (RADIX == 16) pk_dft16(u);
    else pk_dft32(u);
}

}  // namespace specgpu
