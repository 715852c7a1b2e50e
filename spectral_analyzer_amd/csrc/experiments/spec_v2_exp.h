// experiments/spec_v2_exp.h -- experiments on the packed-fp32 family that the product library never compiles
// (included by spec_v2.h only under -DSPEC_V2_ROWS / -DSPEC_V2_STAMPS: build.py --variant v2rows, v2stamp).
//
//  * SPEC_V2_ROWS (round 5): the 16384-point Welch plan 16 x (32 x 32).  Built, parity-green
//    (42 Welch cases), measured SLOWER than 32 x 32 x 16: 6.24 against 5.79 ms per 1024 PSDs of cfg4.  Time line and reasons:
//    profiles/r05_rows.md.
//  * SPEC_V2_STAMPS: lane 0 of every wave notes the shader clock at the phase boundaries of three consecutive
//    segments of the Welch kernels (tools/v2_timeline.py).
// (Included in the middle of spec_v2.h, inside its namespaces, behind the Plan2 table.)
#pragma once

// Plan id 214 (round 5): 16384 points as 16 x (32 x 32) -- ONE workgroup-wide radix-16 step (decimation in frequency: the
// twiddles W_N^(i r) follow the butterfly), then sixteen 1024-point rows, each owned by the 32 lanes of half a wave, whose
// single exchange stays inside that half wave (no s_barrier; a wave's DS operations complete in order).  Two barriers
// per line instead of four, the same arithmetic in another order.  A row's thread ends up with the bins
// g + 16 l + 512 m (g = row, l = lane of the row): consecutive lanes are sixteen bins apart, so the plan serves the
// kernels that do not store a line per transform -- the Welch sums (MODE 1) -- and nothing else.
template <> struct Plan2<214> {
    static constexpr int E = 32, N = 16384, T = N / E, NPASS = 3;
    static constexpr int radix[4] = {16, 32, 32, 1};
    static constexpr int WG = T, LPW = 1;
    static constexpr bool WAVE_LOCAL = false;
    static constexpr int PADSH = 5;
    static constexpr int ROWS = 16, ROW = 1024 + 32;  // a row's region: 1024 elements + one pad per 32 (its own exchange)
    static constexpr int LINE = ROWS * ROW;
};
template <int L> constexpr bool p2_rows() { return L == 214; }

// Plan id 314 (round 5, DESIGN.md 8 item 2 "priced, not built" -> built): 16384 points on 1024 threads x 16 points = FOUR waves per
// SIMD at 128 registers, radix 4 x 16 x 16 x 16 (three exchanges instead of two).  The time line of the 512-thread kernels shows
// `vector + LDS` adding up because two waves per SIMD have nothing to fill each other's waits with; four might.  What makes 128
// registers possible at all: the last pass's fifteen twiddles W_N^(r t) are formed from two LDS tables (as spec_k_v3h.hip does
// in fp64), W_N^(r (t mod 32)) W_N^(32 r (t div 32)), instead of living in 30 registers.  Welch sums only (MODE 1).
template <> struct Plan2<314> {
    static constexpr int E = 16, N = 16384, T = N / E, NPASS = 4;
    static constexpr int radix[4] = {4, 16, 16, 16};
    static constexpr int WG = T, LPW = 1;
    static constexpr bool WAVE_LOCAL = false;
    static constexpr int PADSH = 4;
    static constexpr int LINE = N + (N >> PADSH);
};
template <int L> constexpr bool p2_lds_twl() { return L == 314; }
constexpr int P2_LDS_TWL_ENTRIES = 2 * 15 * 32;  // table A: (r - 1) 32 + j -> W_N^(r j); table B: (r - 1) 32 + h -> W_N^(32 r h)
// multiple of T in the bin index of register m at the end of a transform: the row plan's last butterfly leaves its
// outputs in split order (pk_dft32_split: even bins in the lower sixteen registers, odd bins in the upper)
template <int L> constexpr int p2_bin_reg(int m) { return p2_rows<L>() ? (m < 16 ? 2 * m : 2 * (m - 16) + 1) : m; }

// The same transform, decimation in frequency, outputs left in SPLIT order: u[k] = X[2k], u[16 + k] = X[2k + 1].
// The two 16-point halves are independent once the sixteen radix-2 steps are done -- a caller that consumes the bins
// where they stand (the Welch sums of spec_v2.h's row plan) is finished with the first half's 32 registers before the
// second half starts; the natural-order form above keeps all 64 and both halves' temporaries to the end.
template <int J, typename V> __device__ __forceinline__ void pk_dft32_dif_step(V (&u)[32]) {
    const V s = u[J] + u[J + 16];
    if constexpr (J == 8) {  // (a - b) W_32^8 = -i (a - b)
        u[J + 16] = pk_mul_mi(u[J] - u[J + 16]);
    } else {
        u[J + 16] = pk_mul_w32<J>(u[J] - u[J + 16]);
    }
    u[J] = s;
}
template <typename V, int... J> __device__ __forceinline__ void pk_dft32_dif_steps(V (&u)[32], std::integer_sequence<int, J...>) {
    (pk_dft32_dif_step<J>(u), ...);
}
template <typename V> __device__ __forceinline__ void pk_dft32_split(V (&u)[32]) {
    pk_dft32_dif_steps(u, std::make_integer_sequence<int, 16>{});
    V a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = u[j];
    pk_dft16(a);
#pragma unroll
    for (int j = 0; j < 16; ++j) u[j] = a[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = u[16 + j];
    pk_dft16(a);
#pragma unroll
    for (int j = 0; j < 16; ++j) u[16 + j] = a[j];
}

#ifdef SPEC_V2_STAMPS
#define V2_STAMP(sp, id) do { __builtin_amdgcn_sched_barrier(0); if (sp) { const uint32_t c__ = (uint32_t)__builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) (sp)[id] = c__; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define V2_STAMP(sp, id) do { (void)(sp); } while (0)
#endif

// Plan2<214>: see the plan.  twl[r] = W_N^(r t) as for every plan; tab = W_1024^(m l).
template <typename V>
__device__ __forceinline__ void v2_fft_rows(V (&v)[32], int t, V *lds, const V *tab, V (&twl)[16], uint32_t *sp) {
    using PL = Plan2<214>;
    constexpr int ROW = PL::ROW;
    // pass 0: butterfly s over the registers {s + 2 r} = x[i + 1024 r], i = t + 512 s; output r times W_N^(i r) =
    // W_N^(t r) W_32^(s r)
#ifndef SPEC_ABL_NOFFT
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        V u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = v[s + 2 * r];
        pk_dft16(u);
#pragma unroll
        for (int r = 1; r < 16; ++r) {
            u[r] = pk_cmul(u[r], twl[r]);
            if (s == 1) u[r] = r == 8 ? pk_mul_mi(u[r]) : pk_cmul_const(u[r], kW32[r][0], kW32[r][1]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[s + 2 * r] = u[r];
    }
#endif
    V2_STAMP(sp, 2);
#ifndef SPEC_ABL_NOLDS
#if !defined(SPEC_ABL_NOBAR) && SPEC_V2_LATE_WAR
    __syncthreads();  // every row of the previous line has been read
#endif
    V2_STAMP(sp, 3);
    {   // element i of row r at r ROW + i: a wave writes 64 consecutive elements per instruction
        V *base = lds + t;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) base[r * ROW + s * 512] = v[s + 2 * r];
    }
    V2_STAMP(sp, 4);
#ifndef SPEC_ABL_NOBAR
    __syncthreads();
#endif
    V2_STAMP(sp, 5);
    const int l = t & 31;
    V *row = lds + (t >> 5) * ROW;
    // (volatile: kept as ds_read_b64.  The row's reads are 256 bytes apart and hipcc pairs them into ds_read2_b64, which
    // takes twice the LDS cycles per byte -- MI355X_MICROARCH.md, LDS table -- and needs four consecutive registers)
    // (address space 3 spelled out: a volatile access through a generic pointer would become a flat load)
    typedef const volatile __attribute__((address_space(3))) V *lds_cvp;
    const lds_cvp rowv = (lds_cvp)row;
#pragma unroll
    for (int m = 0; m < 32; ++m) v[m] = rowv[l + 32 * m];
#endif
    // the row's 1024-point transform, 32 x 32 on its 32 lanes: first pass without twiddles
#ifndef SPEC_ABL_NOFFT
    pk_dft32(v);
#endif
    V2_STAMP(sp, 6);
#ifndef SPEC_ABL_NOLDS
    // the row's own exchange (padded: lane l writes 33-element rows): the region is read and written by the lanes of
    // this half wave only, and a wave's DS operations complete in issue order -- a scheduling fence is all it takes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        V *b = row + l * 33;
#pragma unroll
        for (int r = 0; r < 32; ++r) b[r] = v[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    V2_STAMP(sp, 7);
    // second pass: register m times W_1024^(m l), then the butterfly.  Values and twiddles are read SPEC_ROWS_CHUNK at a
    // time: left alone the scheduler requests all 63 first and the twiddles alone hold 62 registers (16 spilled)
#ifndef SPEC_ROWS_CHUNK
#define SPEC_ROWS_CHUNK 16
#endif
    {
        const lds_cvp tr = (lds_cvp)(tab + l);
#pragma unroll
        for (int c = 0; c < 32; c += SPEC_ROWS_CHUNK) {
#pragma unroll
            for (int m = c; m < c + SPEC_ROWS_CHUNK; ++m) {
                v[m] = rowv[l + 33 * m];
#ifndef SPEC_ABL_NOFFT
                if (m > 0) v[m] = pk_cmul(v[m], tr[32 * m]);
#endif
            }
            if (c + SPEC_ROWS_CHUNK < 32) __builtin_amdgcn_sched_barrier(0);
        }
    }
#if !defined(SPEC_ABL_NOBAR) && !SPEC_V2_LATE_WAR
    __syncthreads();  // every row of this line has been read: the next line's first exchange may be written (see v2_fft)
#endif
    V2_STAMP(sp, 8);
#endif
#ifndef SPEC_ABL_NOFFT
    pk_dft32_split(v);  // register p: bin 2 p (p < 16), 2 (p - 16) + 1 beyond -- p2_bin_reg
#endif
    V2_STAMP(sp, 9);
}
