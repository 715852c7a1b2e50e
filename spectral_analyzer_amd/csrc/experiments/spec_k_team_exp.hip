// experiments/spec_k_team_exp.hip -- the team kernel WITH its experiments, frozen at the end of round 4 (round 5 split it from
// csrc/spec_k_team.hip, which is this file with every experiment macro undefined and the dead branches removed; the two compile
// to the same instructions).  Built only into variant libraries (build.py: tWA*, tOLDa, tPROF; tools/r04_wa*.sh, tools/team_prof.py).
// spec_k_team.hip -- four-step FFT for lines longer than the LDS holds (fp32: nfft >= 32768, fp64:
// nfft >= 16384; BASELINE configs[4] is 65536-point cf64) with the intermediate kept in the XCD's L2.
//
// spec_k_large.hip runs the two halves of the decomposition N = N1 N2 (n = N2 n1 + n2, k = k1 + N1 k2) as two
// launches with the [n2][k1] intermediate in HBM: three bytes moved per algorithmic byte.  Here ONE persistent
// launch keeps that intermediate inside one XCD's 4 MiB L2:
//
//   * the launch is sized to be fully resident (512 threads per CU: one workgroup of 512 or two of 256).  Every workgroup reads
//     the id of the XCD it actually runs on (HW_REG_XCC_ID) and takes a ticket on that XCD; after a one-time
//     registration wait the workgroups of one XCD form TEAMS of 2 NT members (NT = N / (8 WG) tiles per line and
//     step), NT "column" workgroups and NT "row" workgroups.  Teams are built from the XCD ids the hardware
//     reports, never from blockIdx, so the result does not depend on how the dispatcher places workgroups --
//     a different placement only changes who is in which team; workgroups left over on an XCD exit.
//   * a team owns a contiguous range of lines and a ring of RING line-sized slots of intermediate (two by default:
//     the slots share the L2 with the streams, and a line that stays put is rewritten there before its write-back).
//     Column workgroup c: C-wide tile of columns n2 (C = 16 for 512 threads), N1-point FFTs over n1 (input rows N2 samples apart), times
//     W_N^(n2 k1), into slot[line % RING] as [n2][k1]; it keeps its registers across lines, so at 50 % overlap
//     half of the next tile is a register move, and the next line's rows are requested before the current FFT.
//     Row workgroup r: C-wide tile of rows k1, N2-point FFTs over n2, epilogue (SS:76-82), X[k1 + N1 k2] out.
//   * flow control, per team and ring slot: doneA counts column tiles stored, doneB row tiles read.  A row
//     workgroup starts line i when doneA == NT (i / RING + 1); a column workgroup may overwrite the slot for
//     line i when doneB == NT (i / RING).  No full barrier: the column side runs up to RING lines ahead.
//   * visibility inside the XCD: the column side's plain stores are write-through in the CU's L1 and land in
//     the XCD's L2; each storing wave waits for them (counted vmcnt), and the last wave to do so adds to doneA
//     (agent-scope atomic).  The row side polls doneA with sc1 loads and reads the slot with sc1 loads (L1 bypass, L2 served;
//     MI355X_MICROARCH "Workgroup dispatch, XCD placement & inter-workgroup visibility").  The slot is
//     rewritten every RING lines and stays dirty in L2: it costs L2 bandwidth, not HBM bandwidth.
//   * a line's three passes need two LDS exchanges per side.  Row side: the first one lives in the landing strips of
//     the LDS-DMA loads (after pass 0 every wave writes only its own strip), the second in the line buffers; column
//     side: the first in the line buffers, the second stays inside a wave (after the role change a column's threads
//     are neighbours).  Two workgroup barriers per line and side.
//   * every spin is bounded (wall clock): on a timeout -- the grid was not co-resident, e.g. the GPU is
//     shared -- the workgroup raises the abort word and every workgroup leaves; the host then runs the
//     two-launch path of spec_k_large.hip (guarded kernels that start only when the abort word is set).
#include <type_traits>

#include "spec_kernels.h"

namespace specgpu {

namespace {

constexpr int TE = 8;         // points per thread
constexpr int TEAM_RING_MAX = 4;
constexpr long long TEAM_SPIN_LIMIT = 200000000ll;  // wall_clock64 ticks (100 MHz): 2 s

// sub-transform of 2^L points by T = 2^L / 8 threads, radices 8 x 8 x (M / 64)
template <int L, int WG> struct TP {
    static_assert(L == 7 || L == 8, "sub-transforms of 128 or 256 points");  // T = 16 or 32 threads: never more than a wave
    static_assert(WG == 256 || WG == 512, "workgroups of 256 or 512 threads");
    static constexpr int M = 1 << L, T = M / TE, C = WG / T;  // C sub-transforms (columns / rows) per tile
    static constexpr int R2 = M / 64, S2 = TE / R2;
    static constexpr int SL = M + 1;  // LDS line stride in elements (odd: adjacent lines start in adjacent slots)
};

// words of the synchronisation block (uint32 each; the host zeroes it before every launch)
enum : uint32_t {
    TS_TOTAL = 0,       // workgroups registered
    TS_ABORT = 16,      // set by a workgroup whose wait timed out (own 64-byte line)
    TS_XCC = 32,        // [8] tickets per XCD, 16 words apart (paired roles: column tickets at +0, row tickets at +1)
    TS_RING = 32 + 8 * 16,  // per team and ring slot: doneA, doneB (16 words apart)
};
constexpr uint32_t TEAM_MAX_TEAMS = 64;
constexpr uint32_t TS_CU = TS_RING + TEAM_MAX_TEAMS * TEAM_RING_MAX * 2 * 16;  // [8 XCDs][256 CU keys] arrivals per CU
constexpr uint32_t TEAM_SYNC_WORDS = TS_CU + 8 * 256;

struct TeamArgs {
    const uint8_t *iq;  // first byte of line 0
    uint32_t n_lines;
    uint32_t hop, bps;
    int kind, be;
    const void *tw1, *tw2;  // W_N1, W_N2 tables (cx<R>)
    const void *twn;        // W_N table, always fp64: inter-step twiddles
    const void *win;        // R[N] or nullptr
    void *scratch;          // cx<R>[teams][ring][N]
    void *out;
    int out_fmt;
    uint32_t ring;
    uint32_t block;         // lines per block of the block-cyclic deal of lines to teams (0: one contiguous range each)
    uint32_t *sync;
};

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup-wide wait until *ctr >= target (wrap-safe).  One lane polls; false (for every thread) when the
// wait timed out or another workgroup has raised the abort word.
__device__ __forceinline__ bool team_wait(const uint32_t *ctr, uint32_t target, uint32_t *sync, int *flag) {
    if (threadIdx.x == 0) {
        int ok = 1;
        if ((int32_t)(ld_sc1(ctr) - target) < 0) {
            const long long t0 = wall_clock64();
            uint32_t spins = 0;
            while ((int32_t)(ld_sc1(ctr) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 31u) == 0 && (ld_sc1(sync + TS_ABORT) != 0 || wall_clock64() - t0 > TEAM_SPIN_LIMIT)) {
                    ok = 0;
                    break;
                }
            }
            if (!ok) __hip_atomic_store(sync + TS_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *flag = ok;
    }
    __syncthreads();
    const int ok = *flag;
    __syncthreads();
    return ok != 0;
}

// The same wait for ONE wave (wave-autonomous sides, SPEC_TEAM_WA): lane 0 polls, the verdict is broadcast; no barrier.
__device__ __forceinline__ bool wave_wait(const uint32_t *ctr, uint32_t target, uint32_t *sync) {
    int ok = 1;
    if ((threadIdx.x & 63) == 0) {
        if ((int32_t)(ld_sc1(ctr) - target) < 0) {
            const long long t0 = wall_clock64();
            uint32_t spins = 0;
            while ((int32_t)(ld_sc1(ctr) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 31u) == 0 && (ld_sc1(sync + TS_ABORT) != 0 || wall_clock64() - t0 > TEAM_SPIN_LIMIT)) {
                    ok = 0;
                    break;
                }
            }
            if (!ok) __hip_atomic_store(sync + TS_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

template <typename R> __device__ __forceinline__ void ctw(cx<R> &u, const cx<R> w) { u = cmul(u, w); }

// An LDS exchange whose writes and reads stay inside one wave needs no s_barrier: a wave's DS operations complete in
// issue order; this only keeps the compiler from moving them across each other (as v2_sync of spec_v2.h).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// passes 1 and 2 of the 8 x 8 x R2 plan on the registers of butterfly index t (pass 0 is a bare dft8)
template <typename R, int L, int WG> __device__ __forceinline__ void pass1(cx<R> (&v)[TE], int t, const cx<R> *tab) {
    const int k = t & 7;
#pragma unroll
    for (int r = 1; r < 8; ++r) ctw(v[r], tab[r * k * (TP<L, WG>::M / 64)]);
    dft8(v);
}
template <typename R, int L, int WG> __device__ __forceinline__ void pass2(cx<R> (&v)[TE], int t, const cx<R> *tab) {
    using P = TP<L, WG>;
#pragma unroll
    for (int s = 0; s < P::S2; ++s) {
        const int k = (t + s * P::T) & 63;
        cx<R> u[P::R2];
#pragma unroll
        for (int r = 0; r < P::R2; ++r) u[r] = v[s + r * P::S2];
#pragma unroll
        for (int r = 1; r < P::R2; ++r) ctw(u[r], tab[r * k]);
        dft<R, P::R2>(u);
#pragma unroll
        for (int r = 0; r < P::R2; ++r) v[s + r * P::S2] = u[r];
    }
}
// The same passes with their twiddles held in registers across lines (13 complex values per thread: the table reads
// of a line were a quarter of its LDS traffic, and the LDS pipe is as busy as the fp64 ALUs in these kernels)
template <typename R, int L, int WG> struct PassTw {
    using P = TP<L, WG>;
    cx<R> w1[7], w2[P::S2][P::R2 - 1];
    __device__ __forceinline__ void load(int t, const cx<R> *tab) {
        const int k = t & 7;
#pragma unroll
        for (int r = 1; r < 8; ++r) w1[r - 1] = tab[r * k * (P::M / 64)];
#pragma unroll
        for (int s = 0; s < P::S2; ++s) {
            const int k2 = (t + s * P::T) & 63;
#pragma unroll
            for (int r = 1; r < P::R2; ++r) w2[s][r - 1] = tab[r * k2];
        }
    }
    __device__ __forceinline__ void pass1(cx<R> (&v)[TE]) const {
#pragma unroll
        for (int r = 1; r < 8; ++r) ctw(v[r], w1[r - 1]);
        dft8(v);
    }
    __device__ __forceinline__ void pass2(cx<R> (&v)[TE]) const {
#pragma unroll
        for (int s = 0; s < P::S2; ++s) {
            cx<R> u[P::R2];
#pragma unroll
            for (int r = 0; r < P::R2; ++r) u[r] = v[s + r * P::S2];
#pragma unroll
            for (int r = 1; r < P::R2; ++r) ctw(u[r], w2[s][r - 1]);
            dft<R, P::R2>(u);
#pragma unroll
            for (int r = 0; r < P::R2; ++r) v[s + r * P::S2] = u[r];
        }
    }
};

// exchanges: after pass 0 butterfly t writes 8 t + r; after pass 1, (t - k) 8 + k + 8 r; reads are t + m T
template <typename R> __device__ __forceinline__ void xstore0(const cx<R> (&v)[TE], int t, cx<R> *line) {
#pragma unroll
    for (int r = 0; r < 8; ++r) line[8 * t + r] = v[r];
}
template <typename R> __device__ __forceinline__ void xstore1(const cx<R> (&v)[TE], int t, cx<R> *line) {
    const int k = t & 7, j = (t - k) * 8 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) line[j + 8 * r] = v[r];
}
template <typename R, int L, int WG> __device__ __forceinline__ void xload(cx<R> (&v)[TE], int t, const cx<R> *line) {
#pragma unroll
    for (int m = 0; m < TE; ++m) v[m] = line[t + m * TP<L, WG>::T];
}

// ---- loads the compiler does not track ------------------------------------------------------------------
// Both sides keep loads in flight ACROSS loop iterations and behind younger stores.  hipcc inserts its own
// s_waitcnt for every load it knows; for a result that is used one iteration later it cannot count the
// instructions in between and waits for vmcnt(0) -- which, vmcnt being one in-order counter for loads and
// stores, also waits for every store issued since (the row side then sat out the HBM write latency of its own
// output once per line).  Loads written as inline assembly are invisible to that pass, but a load with a VGPR
// destination is not safe either: the compiler believes the value is there at ";;#ASMEND" and resolved the
// loops' phi nodes by copying those registers right behind the load -- copying what had not arrived (wrong
// lines; tying the operands "+v" did not stop it).  So the pipelined loads are LDS-DMA (global_load_lds: no
// register destination, nothing for the compiler to copy): every wave lands its loads in a private 1 KiB-per-
// instruction strip of LDS (lane l at byte 16 l), waits with vm_wait<N> -- "all but the N youngest vector-memory
// instructions of this wave are done", N a LOWER bound of the instructions issued since -- and reads its own
// lanes back with ds_read_b128.  The waits carry a "memory" clobber: no LDS read moves above them.  LDS-DMA moves
// 4 or 16 bytes per lane: a cx<double> each, or two neighbouring cx<float> of one row (dma_coords in the kernel).
#ifdef SPEC_TEAM_STRICT_WAITS  // debugging aid: every counted wait becomes vmcnt(0)
#define VM_N(N) 0
#else
#define VM_N(N) (N)
#endif
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_N(N)) : "memory"); }

// development aid (variant builds with -DSPEC_TEAM_PROF): lane 0 of every workgroup adds up shader-clock cycles
// spent in its waits; eight 64-bit words per workgroup behind the synchronisation block
#ifdef SPEC_TEAM_PROF
#define PROF_DECL unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pf_t = 0; (void)pf_t
#define PROF_T0() (pf_t = __builtin_readcyclecounter())
#define PROF_ADD(k) (pf[k] += __builtin_readcyclecounter() - pf_t)
#define PROF_INC(k) (pf[k] += 1)
#define PROF_PH(k) do { const unsigned long long n__ = __builtin_readcyclecounter(); ph[k] += n__ - ph_t; ph_t = n__; } while (0)
#define PROF_OUT(role)                                                                                          \
    do {                                                                                                        \
        if (threadIdx.x == 0) {                                                                                 \
            unsigned long long *o = reinterpret_cast<unsigned long long *>(a.sync + TEAM_SYNC_WORDS) + 16ull * blockIdx.x; \
            pf[7] = (unsigned long long)(role) | ((unsigned long long)team << 8) | ((unsigned long long)my_lines << 32);   \
            {                                                                                                   \
                uint32_t hw__, xc__;                                                                            \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__));                             \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc__));                            \
                pf[6] = hw__ | ((unsigned long long)xc__ << 32);                                                \
            }                                                                                                   \
            for (int k = 0; k < 8; ++k) o[k] = pf[k];                                                           \
            for (int k = 0; k < 8; ++k) o[8 + k] = ph[k];                                                       \
        }                                                                                                       \
    } while (0)
// event trace of ONE team (team 0), lines TRACE_L0 .. TRACE_L0 + 31, lane 0 of every member workgroup: wall-clock ticks
// (100 MHz, the same clock on every CU) behind the per-workgroup words -- tools/team_trace.py prints the hand-offs
#define TRACE_L0 1000u
#define TRACE(ev, i)                                                                                                  \
    do {                                                                                                              \
        if (threadIdx.x == 0 && team == 0 && (i) >= TRACE_L0 && (i) < TRACE_L0 + 32u)                                   \
            (reinterpret_cast<unsigned long long *>(a.sync + TEAM_SYNC_WORDS) + 16384ull)[((size_t)member * 32u + ((i) - TRACE_L0)) * 8u + (ev)] = \
                (unsigned long long)wall_clock64();                                                                   \
    } while (0)
#else
#define TRACE(ev, i) (void)0
#define PROF_DECL (void)0
#define PROF_T0() (void)0
#define PROF_ADD(k) (void)0
#define PROF_INC(k) (void)0
#define PROF_PH(k) (void)0
#define PROF_OUT(role) (void)0
#endif

__device__ __forceinline__ uint32_t lds_addr(const void *p) {  // LDS byte address of a pointer into shared memory
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}
// 16 bytes per lane from gsrc (per lane) to LDS lds_dst + 16 * lane (lds_dst wave-uniform).  M0 carries the LDS
// base; it is the compiler's register, so it is saved and restored inside the statement.  The leading
// lgkmcnt(0): this wave's earlier reads of the strip have left the LDS queue before the strip is rewritten.
// POLICY 0: nt (the recording, read once); 1: sc1 (the ring slot: L2, never this CU's L1)
template <int POLICY> __device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    if constexpr (POLICY == 0)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else if constexpr (POLICY == 1)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else  // 2: plain (cached): a row of the recording is requested by several waves in pieces, one after the other
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// one counter word (lane 0 of the calling branch) to LDS lds_dst, sc1
__device__ __forceinline__ void glds4_sc1(const uint32_t *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// the same load as the compiler knows it (plain forms)
template <typename R> __device__ __forceinline__ cx<R> ld_stream(const uint8_t *p) {
    if constexpr (sizeof(R) == 8) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 u = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p));
        return cx<R>{u.x, u.y};
    } else {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 u = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(p));
        return cx<R>{u.x, u.y};
    }
}
typedef uint32_t tu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t tu32x2 __attribute__((ext_vector_type(2)));
// one cx<R> from the ring slot with an sc1 load the compiler knows: served by the XCD's L2, never by this CU's L1
template <typename R> __device__ __forceinline__ cx<R> ld_slot(__amdgpu_buffer_rsrc_t rs, int voff) {
    constexpr int SC1 = 16;
    if constexpr (sizeof(R) == 8) {
        const tu32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, SC1);
        return cx<R>{__longlong_as_double((long long)(((uint64_t)u.y << 32) | u.x)),
                     __longlong_as_double((long long)(((uint64_t)u.w << 32) | u.z))};
    } else {
        const tu32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, SC1);
        return cx<R>{__uint_as_float(u.x), __uint_as_float(u.y)};
    }
}

template <typename R> __device__ __forceinline__ void st_slot(cx<R> *p, cx<R> v) {  // plain store: stays in L2
#ifdef SPEC_ABL_TEAM_NOSLOT  // ablation: the value stays alive, nothing is stored
    if (v.x != (R)1.2345e-30) return;
#endif
    if constexpr (sizeof(R) == 8) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<d2 *>(p) = d2{v.x, v.y};
    } else {
        typedef float f2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f2 *>(p) = f2{v.x, v.y};
    }
}

#ifdef SPEC_ABL_TEAM_NOSTORE
constexpr int TEAM_NST = 0;
#else
#ifdef SPEC_TEAM_SINGLE_STORES
constexpr int TEAM_NST = TE;      // output stores per thread and line: one instruction per bin, every format
#else
constexpr int TEAM_NST = TE / 2;  // output stores per thread and line: one instruction per pair of bins, every format
#endif
#endif

// value of one bin (SS:76-82) in the arithmetic of the pipeline
template <typename R, int FMT> __device__ __forceinline__ R bin_value(cx<R> z, const double *dbt) {
#ifdef SPEC_ABL_TEAM_NOEPI
    return z.x;
#else
    constexpr bool dbf = FMT == OUT_DB20_F32 || FMT == OUT_DB20_F64;
    if constexpr (sizeof(R) == 4) return dbf ? db20(z) : z.x * z.x + z.y * z.y;
    else return dbf ? db20_tab(z, dbt) : __builtin_fma(z.x, z.x, z.y * z.y);
#endif
}
// the value of the neighbouring lane (lane ^ 1): DPP quad permutation, no LDS
__device__ __forceinline__ float lane_swap1(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ double lane_swap1(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// two adjacent bins in one store (non-temporal: written once, never read by this launch)
template <typename T> __device__ __forceinline__ void st_pair(T *p, T lo, T hi) {
    typedef T t2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(t2{lo, hi}, reinterpret_cast<t2 *>(p));
}

// one bin of the result (SS:76-82), non-temporal: written once, never read by this launch
template <typename R, int FMT> __device__ __forceinline__ void emit_bin(void *out, uint64_t idx, cx<R> z, const double *dbt) {
#ifdef SPEC_ABL_TEAM_NOSTORE
    if (z.x == (R)1.2345e-30) static_cast<float *>(out)[idx & 1023] = 0.0f;  // never true: keeps z alive
#elif defined(SPEC_ABL_TEAM_NOEPI)
    if constexpr (FMT >= OUT_DB20_F64) __builtin_nontemporal_store((double)z.x, static_cast<double *>(out) + idx);
    else __builtin_nontemporal_store((float)z.x, static_cast<float *>(out) + idx);
#else
    if constexpr (sizeof(R) == 4) {
        const float r = FMT == OUT_DB20_F32 ? db20(z) : z.x * z.x + z.y * z.y;
        __builtin_nontemporal_store(r, static_cast<float *>(out) + idx);
    } else {
        const bool dbf = FMT == OUT_DB20_F32 || FMT == OUT_DB20_F64;
        const double r = dbf ? db20_tab(z, dbt) : __builtin_fma(z.x, z.x, z.y * z.y);
        if constexpr (FMT >= OUT_DB20_F64) __builtin_nontemporal_store(r, static_cast<double *>(out) + idx);
        else __builtin_nontemporal_store((float)r, static_cast<float *>(out) + idx);
    }
#endif
}

// Dynamic LDS of one workgroup.  Pipelined loads: 256 bytes for the polled counter word, then the landing
// strips -- WG / 64 waves x TE x 64 elements (one 1 KiB instruction lands 64 cx<double> or 128 cx<float>) -- FIRST,
// so that every strip's base address fits the 16 bits of M0 that are certain to carry it; then line buffers +
// sub-transform table (MAIN, in cx<R> elements).
// DENSE: two workgroups per CU (four waves per SIMD, 128 registers each) that hide latency by occupancy: no
// landing strips, the plain forms of both sides.
template <typename R, int L1, int L2, int WG, bool DENSE> struct TeamLds {
#ifdef SPEC_TEAM_WA
    // wave-autonomous column side: line buffers padded by one element per eight (WSL = M + M / 8 elements per column)
    static constexpr size_t A = (size_t)TP<L1, WG>::C * (TP<L1, WG>::M + TP<L1, WG>::M / 8) + TP<L1, WG>::M;
#else
    static constexpr size_t A = (size_t)TP<L1, WG>::C * TP<L1, WG>::SL + TP<L1, WG>::M;
#endif
    static constexpr size_t B = (size_t)TP<L2, WG>::C * TP<L2, WG>::SL + TP<L2, WG>::M;
    static constexpr size_t MAIN = A > B ? A : B;
    static constexpr bool PIPE = !DENSE;
    static constexpr size_t LAND_BYTES = PIPE ? 256 + (size_t)(WG / 64) * TE * 64 * sizeof(cx<R>) : 0;
    static constexpr size_t DBT_OFF = LAND_BYTES + MAIN * sizeof(cx<R>);  // table of the fp64 dB epilogue (spec_fft.h)
    static constexpr size_t BYTES = DBT_OFF + DB20_TAB_DOUBLES * sizeof(double);
    // the last strip starts at LAND_BYTES - 1024 behind the kernel's static __shared__ words (< 768 bytes)
    static_assert(LAND_BYTES == 0 || LAND_BYTES - 1024 + 767 <= 65535, "strip bases must stay below 64 KiB");
};

// HALF: hop == N / 2 (BASELINE's 50 % overlap): the lower half of line i + 1 is the upper half of line i and stays
// in registers.  A template parameter because the pipelined column loop's waits count its load instructions.
template <typename R, int L1, int L2, bool DIRECT, bool HALF, int WG, bool DENSE>
__global__ __launch_bounds__(WG, DENSE ? 4 : 2) void large_team_kernel(const TeamArgs a) {
    using PA = TP<L1, WG>;
    using PB = TP<L2, WG>;
    using LD = TeamLds<R, L1, L2, WG, DENSE>;
    constexpr int N1 = PA::M, N2 = PB::M, N = N1 * N2;
    constexpr uint32_t NT = N / (WG * TE);  // tiles per line and step
    static_assert(N2 / PA::C == (int)NT && N1 / PB::C == (int)NT, "tile counts of the two steps match");
    constexpr uint32_t TEAM = 2 * NT;
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    constexpr int NEWH = TE / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_flag, s_next;
    __shared__ uint32_t s_info[4];
    __shared__ uint32_t s_arrive;  // column side: waves whose stores of the pending line are in L2 (monotonic)
    __shared__ uint32_t s_arrive_slot[TEAM_RING_MAX];  // wave-autonomous column side: the same count per ring slot
    constexpr uint32_t WAVES = WG / 64;
#ifndef SPEC_TEAM_WA_DEEP
#define SPEC_TEAM_WA_DEEP 1  // wave-autonomous column side at 50 % overlap: requests two lines ahead
#endif
#ifndef SPEC_TEAM_WA_ANN
#define SPEC_TEAM_WA_ANN 0  // where a wave of the wave-autonomous column side waits for its previous line's stores and
#endif                      // announces: 0 behind pass 0's exchange write, 1 behind pass 1, 2 behind the whole transform
#ifndef SPEC_TEAM_WA_LD
#define SPEC_TEAM_WA_LD 2  // cache policy of the wave-autonomous column side's requests (glds16): 0 nt, 2 plain
#endif
#ifdef SPEC_TEAM_WA
    // wave-autonomous column side (fp64 lines, samples as they are in memory): every WAVE announces its own two columns
    constexpr bool WA_COL = sizeof(R) == 8 && DIRECT && !DENSE && WG == 512;
#else
    constexpr bool WA_COL = false;
#endif
    constexpr uint32_t APT = 1u;  // announcements (adds to doneA) per column tile and line (one per WAVE was tried: 128 adds per
                                  // line and team on one L2 word -- 35 ms per cfg5 step against 13.3)
    double *s_dbt = reinterpret_cast<double *>(smem + LD::DBT_OFF);
#ifdef SPEC_TEAM_PROF
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = 0;  // phase stamps of lane 0 (tools/team_prof.py)
#endif
    cx<R> *lds = reinterpret_cast<cx<R> *>(smem + LD::LAND_BYTES);  // line buffers + table
    const int tid = threadIdx.x;
    uint32_t *sync = a.sync;

    // ---- registration: which XCD am I on, which ticket do I hold there --------------------------------
    // PAIRED (two 256-thread workgroups per CU): the first workgroup to arrive on a CU takes a column role, the
    // second a row role, so that every CU carries one of each -- the row side has half again as much arithmetic per
    // line, and two workgroups with barriers of their own fill each other's LDS and barrier phases.  Which CU:
    // HW_REG_HW_ID bits 8..15 (CU, shader array, shader engine); were that key ever shared by two CUs the roles
    // would still come out in equal numbers, only the pairing would be imperfect.
    constexpr bool PAIRED = WG == 256;
    if (tid == 0) s_arrive = 0;
    if (tid < TEAM_RING_MAX) s_arrive_slot[tid] = 0;
    if (tid == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        uint32_t ticket;
        if constexpr (PAIRED) {
            uint32_t hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            const uint32_t key = (hwid >> 8) & 0xFFu;
            const uint32_t role = __hip_atomic_fetch_add(sync + TS_CU + 256 * xcc + key, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
            ticket = __hip_atomic_fetch_add(sync + TS_XCC + 16 * xcc + role, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_info[2] = role;
        } else {
            ticket = __hip_atomic_fetch_add(sync + TS_XCC + 16 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_fetch_add(sync + TS_TOTAL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_info[0] = xcc;
        s_info[1] = ticket;
    }
    if (tid < DB20_TAB_DOUBLES) s_dbt[tid] = DB20_TAB[tid];
    __syncthreads();
    if (!team_wait(sync + TS_TOTAL, gridDim.x, sync, &s_flag)) return;
    // everybody has registered: the tickets per XCD are final
    if (tid == 0) {
        const uint32_t xcc = s_info[0], ticket = s_info[1];
        uint32_t teams_before = 0, teams_total = 0, mine = 0;
        for (uint32_t x = 0; x < 8; ++x) {
            uint32_t t;
            if constexpr (PAIRED) {  // a team needs NT members of either role
                const uint32_t nc = ld_sc1(sync + TS_XCC + 16 * x), nr = ld_sc1(sync + TS_XCC + 16 * x + 1);
                t = (nc < nr ? nc : nr) / NT;
            } else {
                t = ld_sc1(sync + TS_XCC + 16 * x) / TEAM;
            }
            if (x < xcc) teams_before += t;
            if (x == xcc) mine = t;
            teams_total += t;
        }
        uint32_t local_team, member;
        if constexpr (PAIRED) {
            local_team = ticket / NT;
            member = s_info[2] * NT + ticket % NT;
        } else {
            local_team = ticket / TEAM;
            member = ticket % TEAM;
        }
        s_info[1] = member;
        s_info[2] = local_team < mine ? teams_before + local_team : NONE;  // left over on this XCD: no team
        s_info[3] = teams_total;
        if (teams_total == 0 || teams_total > TEAM_MAX_TEAMS)  // nobody could form a team: the host falls back
            __hip_atomic_store(sync + TS_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const uint32_t team = s_info[2], n_teams = s_info[3], member = s_info[1];
    if (team == NONE || n_teams == 0 || n_teams > TEAM_MAX_TEAMS) return;
    // Which lines are this team's.  block == 0: one contiguous range per team.  Otherwise blocks of `block`
    // consecutive lines are dealt to the teams in turn: at any moment the teams then work within a few MiB of
    // each other instead of at equal offsets in regions 2^k bytes apart (the same HBM channels and banks for
    // every team), and a team still walks consecutive lines inside a block (the register reuse of 50 % overlap).
    const uint32_t blk = a.block;
    uint32_t line_first = 0, my_lines;
    if (blk == 0) {
        line_first = (uint32_t)((uint64_t)a.n_lines * team / n_teams);
        my_lines = (uint32_t)((uint64_t)a.n_lines * (team + 1) / n_teams) - line_first;
    } else {
        const uint32_t nb = (a.n_lines + blk - 1) / blk;
        const uint32_t mine = team < nb ? (nb - team + n_teams - 1) / n_teams : 0;
        my_lines = mine * blk;
        if (mine && team + (mine - 1) * n_teams == nb - 1) my_lines -= nb * blk - a.n_lines;  // the last block may be short
    }
    auto line_of = [&](uint32_t i) -> uint32_t {  // the i-th line of this team
        return blk ? ((i / blk) * n_teams + team) * blk + i % blk : line_first + i;
    };
    auto follows = [&](uint32_t i) -> bool {  // is the team's i-th line the recording's next line after its (i-1)-th?
        return blk == 0 || i % blk != 0;
    };
    cx<R> *slots = static_cast<cx<R> *>(a.scratch) + (uint64_t)team * a.ring * N;
    uint32_t *ring = sync + TS_RING + team * (TEAM_RING_MAX * 32);  // slot s: doneA at 32 s, doneB at 32 s + 16
    const cx<double> *__restrict__ twn = static_cast<const cx<double> *>(a.twn);
    if (my_lines == 0) return;
    // landing strips of the pipelined loads: element m of this wave's lane l at land[64 m + l].  One LDS-DMA
    // instruction moves 16 bytes per lane = EPL elements: for cx<double> the lane's own element m = j; for cx<float>
    // two neighbouring columns of ONE row, and the 64 lanes of instruction j cover the rows m = 2 j and 2 j + 1 of the
    // wave's threads in exactly the [m][lane] order above (dma_coords below), so that only the requests differ
    // between the two precisions, not what is read back.
    const uint32_t wave = (uint32_t)tid >> 6, lane = (uint32_t)tid & 63;
    constexpr int EPL = 16 / (int)sizeof(cx<R>);  // elements per lane and LDS-DMA instruction
    constexpr int NI = TE / EPL;                   // instructions per tile and wave, 1 KiB of strip each
    uint32_t *pland = reinterpret_cast<uint32_t *>(smem);
    cx<R> *land = reinterpret_cast<cx<R> *>(smem + 256 + (size_t)wave * TE * 64 * sizeof(cx<R>));
    // which (row within the tile's thread grid, column, m offset) a lane REQUESTS for a tile C columns wide
    struct DmaCoords { uint32_t t, q, mm; };
    auto dma_coords = [&](uint32_t C) -> DmaCoords {
        const uint32_t half = C / EPL, s = lane / half;         // EPL == 1: half = C, s = lane / C
        const uint32_t per_wave = 64u / C;                       // thread rows t of one wave
        return DmaCoords{wave * per_wave + s % per_wave, EPL * (lane % half), s / per_wave};
    };
    const uint32_t land_addr = __builtin_amdgcn_readfirstlane(lds_addr(land));
    const uint32_t pland_addr = __builtin_amdgcn_readfirstlane(lds_addr(pland));
    (void)land_addr; (void)pland_addr; (void)lane;

    if (member < NT) {
        // ================= column side: tile of C columns, N1-point transforms over n1 =====================
#ifdef SPEC_ABL_TEAM_NOA
        return;
#endif
        const uint32_t c0 = member * PA::C;
        const int q0 = tid % PA::C, t0 = tid / PA::C;  // loads: columns fastest (contiguous samples)
        const int t1 = tid % PA::T, q1 = tid / PA::T;  // stores: k1 fastest (contiguous intermediate)
        constexpr int WSL = PA::M + PA::M / 8;  // wave-autonomous form: padded column buffers (below)
        cx<R> *tab = lds + (size_t)PA::C * (WA_COL ? WSL : PA::SL);
        for (int e = tid; e < N1; e += WG) tab[e] = static_cast<const cx<R> *>(a.tw1)[e];
        const R *__restrict__ win = static_cast<const R *>(a.win);
        // inter-step twiddle W_N^(n2 k1), k1 = t1 + m T: W^(n2 t1) (W^(n2 T))^m, recurrence in fp64
        const uint32_t n2 = c0 + q1;
        const cx<double> w0 = twn[n2 * (uint32_t)t1], wstep = twn[n2 * (uint32_t)PA::T];
        // fp32 lines: the eight twiddles W_N^(n2 (t1 + m T)) of a thread do not depend on the line -- straight from the
        // table (n2 k1 < N), rounded to fp32 once, held in 16 registers the fp32 kernel has to spare: eight packed
        // complex multiplications per line instead of sixteen fp64 ones and the conversions either way (round 3; the
        // fp64 kernel keeps the recurrence: it has no 32 registers left)
        constexpr bool TW_LINE_INV = sizeof(R) == 4 && !DENSE;
        cx<R> wtw[TW_LINE_INV ? TE : 1];
        if constexpr (TW_LINE_INV) {
#pragma unroll
            for (int m = 0; m < TE; ++m) {
                const cx<double> w = twn[n2 * (uint32_t)(t1 + m * PA::T)];
                wtw[m] = cx<R>{(R)w.x, (R)w.y};
            }
        }
        (void)wtw;
        // rest of one line behind the first exchange: pass 2 and the inter-step twiddle
        PassTw<R, L1, WG> twr;
        constexpr bool TWREG = !DENSE;  // twiddles in registers where there are 256 of them
        if constexpr (TWREG) {
            __syncthreads();  // table visible
            twr.load(t1, tab);
        }
        auto finish = [&](cx<R> (&v)[TE]) {
#ifndef SPEC_ABL_TEAM_NOFFT
            if constexpr (TWREG) twr.pass2(v);
            else pass2<R, L1, WG>(v, t1, tab);
#endif
            if constexpr (TW_LINE_INV) {
#pragma unroll
                for (int m = 0; m < TE; ++m) v[m] = cmul(v[m], wtw[m]);
            } else {
                cx<double> w = w0;
#pragma unroll
                for (int m = 0; m < TE; ++m) {
                    const cx<double> z = cmul(cx<double>{(double)v[m].x, (double)v[m].y}, w);
                    v[m] = cx<R>{(R)z.x, (R)z.y};
                    w = cmul(w, wstep);
                }
            }
        };
        if constexpr (WA_COL) {
            if (win == nullptr) {
                // ---- wave-autonomous form (round 4) ----------------------------------------------------------------------
                // The workgroup's 16 columns are dealt to its 8 waves, two each; a column's 32 threads are one half of a
                // wave, so ALL of a line's exchanges stay inside the wave and nothing in the line loop is a workgroup
                // barrier: the eight waves of a CU drift apart and fill each other's LDS, vector-memory and arithmetic
                // phases instead of walking them in lock step (one workgroup's waves in step leave the vector
                // ALU 0.36 busy).  Every wave requests its own two columns (32- or 64-byte pieces: CPW adjacent lanes = one row of its
                // columns), keeps line i + 1's upper half in flight behind line i's transform, stores and announces its own
                // columns, polls the ring for itself (a word of its own), and the last wave to see its stores done announces the tile.
                constexpr uint32_t CPW = 64u / PA::T;                      // columns per wave: 2 (256-point columns) or 4 (128-point)
                const uint32_t wq = lane / PA::T, wt = lane % PA::T;       // transform role: column wq of the wave's, butterfly index wt
                const uint32_t ncol = c0 + CPW * wave + wq;                // n2 of this lane's column
                cx<R> *colbuf = lds + (size_t)(CPW * wave + wq) * WSL;     // that column's line buffer: this wave's alone
                // Exchanges in the (column, butterfly) lane order: a column's threads are NEIGHBOURS, so pass 0's writes
                // (element 8 t + r: lanes 128 bytes apart) would all hit the same four banks -- measured 35 ms per cfg5
                // step.  One pad element per eight, index a -> a + (a >> 3): eight neighbouring lanes then cover all 32
                // banks once on every write and read of both exchanges.
                auto pad8 = [](int a) { return a + (a >> 3); };
                auto wstore0 = [&](const cx<R> (&x)[TE]) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) colbuf[9 * (int)wt + r] = x[r];            // pad8(8 t + r) = 9 t + r
                };
                auto wstore1 = [&](const cx<R> (&x)[TE]) {
                    const int k = (int)wt & 7, j = ((int)wt - k) * 8 + k;
#pragma unroll
                    for (int r = 0; r < 8; ++r) colbuf[pad8(j) + 9 * r] = x[r];            // pad8(j + 8 r): (j + 8 r) >> 3 = (j >> 3) + r
                };
                auto wload = [&](cx<R> (&x)[TE]) {
#pragma unroll
                    for (int m = 0; m < TE; ++m) x[m] = colbuf[pad8((int)wt + m * PA::T)];
                };
#ifdef SPEC_ABL_WA_COALESCED  // ablation (results WRONG by construction): the same request volume as 256-byte rows -- 16 neighbouring
                                   // lanes = the workgroup's 16 columns of one row -- to price the 32-byte pieces of the real thing
                const uint32_t dk = 4u * wave + lane / 16u, dcol = c0 + lane % 16u;
#else
                const uint32_t dk = lane / CPW, dcol = c0 + CPW * wave + lane % CPW;  // request role: row dk (+ T m) of column dcol
#endif
                constexpr int NLD = HALF ? NEWH : TE;
                auto issue_rows = [&](uint32_t line, auto first_tag, auto end_tag) {
                    constexpr int FIRST = decltype(first_tag)::value, END = decltype(end_tag)::value;
                    const uint8_t *src = a.iq + (uint64_t)line * a.hop * sizeof(cx<R>);
#pragma unroll
                    for (int m = FIRST; m < END; ++m)
                        glds16<SPEC_TEAM_WA_LD>(src + (uint64_t)((dk + (uint32_t)PA::T * m) * N2 + dcol) * sizeof(cx<R>), land_addr + 1024u * m);
                };
                auto issue_next = [&](uint32_t i_next) {
                    const uint32_t ic = i_next < my_lines ? i_next : my_lines - 1;  // tail: a valid line again, unused
                    const uint32_t ln = line_of(ic);
                    if constexpr (HALF) {
                        if (!follows(ic)) issue_rows(ln, std::integral_constant<int, 0>{}, std::integral_constant<int, NEWH>{});
                        issue_rows(ln, std::integral_constant<int, NEWH>{}, std::integral_constant<int, TE>{});
                    } else {
                        issue_rows(ln, std::integral_constant<int, 0>{}, std::integral_constant<int, TE>{});
                    }
                };
                const cx<R> *mine = land + CPW * wt + wq;                  // element (row wt + T m) of my column at mine[64 m]
                __syncthreads();  // table visible (set-up: the only workgroup barrier of this side)
                PassTw<R, L1, WG> wtw;
                wtw.load((int)wt, tab);
                const cx<double> ww0 = twn[ncol * wt], wwstep = twn[ncol * (uint32_t)PA::T];
                cx<R> cur[TE];
                issue_rows(line_of(0), std::integral_constant<int, 0>{}, std::integral_constant<int, TE>{});
                vm_wait<0>();
#pragma unroll
                for (int m = 0; m < TE; ++m) cur[m] = mine[64 * m];
                wave_sync();
                uint32_t pending = NONE;
#if SPEC_TEAM_WA_DEEP
                if constexpr (HALF) {
                    if (blk == 0) {
                        // ---- 50 % overlap, contiguous lines: the new upper half of a line is requested TWO lines ahead.  One line
                        // ahead, a wave's loop time is bounded by the memory latency (its stores and the next requests sit in one
                        // in-order queue, and the next line cannot start before they are back: measured 1.9 us per line with the
                        // slot stores removed, 3.9 with them, against 1.3 us of arithmetic).  The strip is half empty at 50 %
                        // overlap: lines alternate between its two halves (instruction slots 4 p .. 4 p + 3, p = line & 1).
                        auto issue_upper = [&](uint32_t i_line) {
                            const uint32_t ic = i_line < my_lines ? i_line : my_lines - 1;  // tail: a valid line again, unused
                            const uint8_t *src = a.iq + (uint64_t)line_of(ic) * a.hop * sizeof(cx<R>);
                            const uint32_t base = land_addr + 1024u * NEWH * (i_line & 1u);
#pragma unroll
                            for (int m = 0; m < NEWH; ++m)
                                glds16<SPEC_TEAM_WA_LD>(src + (uint64_t)((dk + (uint32_t)PA::T * (m + NEWH)) * N2 + dcol) * sizeof(cx<R>), base + 1024u * m);
                        };
                        issue_upper(1);
                        issue_upper(2);
#ifdef SPEC_ABL_TEAM_NOSLOT
                        constexpr int YOUNGER = NEWH;           // ablation build: no slot stores in the queue
#else
                        constexpr int YOUNGER = TE + NEWH;      // behind line i + 1's requests: the stores of line i - 1, line i + 2's requests
#endif
                        for (uint32_t i = 0; i < my_lines; ++i) {
                            const uint32_t slot = i % a.ring, round = i / a.ring;
                            cx<R> v[TE];
#pragma unroll
                            for (int m = 0; m < TE; ++m) v[m] = cur[m];
#ifndef SPEC_ABL_TEAM_NOFFT
                            dft8(v);
#endif
                            wstore0(v);
                            wave_sync();
                            wload(v);
                            wave_sync();
#ifndef SPEC_ABL_TEAM_NOFFT
                            wtw.pass1(v);
#endif
                            wstore1(v);
                            wave_sync();
                            wload(v);
                            wave_sync();
#ifndef SPEC_ABL_TEAM_NOFFT
                            wtw.pass2(v);
#endif
                            {   // inter-step twiddle W_N^(n2 k1), k1 = wt + T m: recurrence in fp64
                                cx<double> w = ww0;
#pragma unroll
                                for (int m = 0; m < TE; ++m) {
                                    v[m] = cmul(v[m], w);
                                    w = cmul(w, wwstep);
                                }
                            }
                            // line i + 1's upper half was requested two lines ago; what is younger may stay in flight
                            if (i == 0) vm_wait<NEWH>();
                            else vm_wait<YOUNGER>();
                            {
                                const cx<R> *up = mine + 64 * NEWH * ((i + 1) & 1u);
#pragma unroll
                                for (int m = 0; m < NEWH; ++m) { cur[m] = cur[m + NEWH]; cur[m + NEWH] = up[64 * m]; }
                            }
                            wave_sync();
                            // the stores of line i - 1 (older than line i + 2's requests) have had a whole line: announce it
                            vm_wait<NEWH>();
                            if (lane == 0 && pending != NONE && atomicAdd(&s_arrive_slot[pending], 1u) + 1 == WAVES * ((i - 1) / a.ring + 1))
                                __hip_atomic_fetch_add(ring + 32 * pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef SPEC_ABL_TEAM_NOWAIT
                            if (round != 0 && (pending == NONE || (int32_t)(pland[2u * wave + (i & 1u)] - NT * round) < 0) &&
                                !wave_wait(ring + 32 * slot + 16, NT * round, sync)) return;
#endif
                            {
                                const uint32_t word = __builtin_amdgcn_readfirstlane(pland_addr + 8u * wave + 4u * ((i + 1) & 1u));
                                if (lane == 0) glds4_sc1(ring + 32 * ((i + 1) % a.ring) + 16, word);  // before the stores
                            }
                            cx<R> *dst = slots + (uint64_t)slot * N + (uint64_t)ncol * N1;
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int m = 0; m < TE; ++m) st_slot<R>(dst + wt + m * PA::T, v[m]);
                            asm volatile("" ::: "memory");  // the loads below stay behind the stores above
                            pending = slot;
                            issue_upper(i + 3);             // into the half of the strip just read
                        }
                        vm_wait<0>();
                        if (lane == 0 && atomicAdd(&s_arrive_slot[pending], 1u) + 1 == WAVES * ((my_lines - 1) / a.ring + 1))
                            __hip_atomic_fetch_add(ring + 32 * pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                }
#endif
                issue_next(1);
                for (uint32_t i = 0; i < my_lines; ++i) {
                    const uint32_t slot = i % a.ring, round = i / a.ring;
                    cx<R> v[TE];
#pragma unroll
                    for (int m = 0; m < TE; ++m) v[m] = cur[m];
#ifndef SPEC_ABL_TEAM_NOFFT
                    dft8(v);
#endif
                    wstore0(v);
                    wave_sync();
                    auto announce = [&]() {
                        // the LAST wave of the workgroup to see its stores of the pending line (i - 1) done announces the tile.
                        // Counted per ring slot: the waves are not in step, but nobody arrives for line j + ring before line j
                        // has been announced (its slot is not free before that)
                        if (lane == 0 && pending != NONE && atomicAdd(&s_arrive_slot[pending], 1u) + 1 == WAVES * ((i - 1) / a.ring + 1))
                            __hip_atomic_fetch_add(ring + 32 * pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    };
#if SPEC_TEAM_WA_ANN == 0
                    // the previous line's stores have had this long: wait for them, not for the NLD loads behind them
                    vm_wait<NLD>();
                    announce();
#endif
                    wload(v);
                    wave_sync();
#ifndef SPEC_ABL_TEAM_NOFFT
                    wtw.pass1(v);
#endif
#if SPEC_TEAM_WA_ANN == 1
                    vm_wait<NLD>();
                    announce();
#endif
                    wstore1(v);
                    wave_sync();
                    wload(v);
                    wave_sync();
#ifndef SPEC_ABL_TEAM_NOFFT
                    wtw.pass2(v);
#endif
                    {   // inter-step twiddle W_N^(n2 k1), k1 = wt + T m: recurrence in fp64
                        cx<double> w = ww0;
#pragma unroll
                        for (int m = 0; m < TE; ++m) {
                            v[m] = cmul(v[m], w);
                            w = cmul(w, wwstep);
                        }
                    }
                    vm_wait<0>();  // line i + 1's rows, requested a whole line ago
#if SPEC_TEAM_WA_ANN == 2
                    announce();
#endif
                    if (HALF && follows(i + 1 < my_lines ? i + 1 : my_lines - 1)) {
#pragma unroll
                        for (int m = 0; m < NEWH; ++m) { cur[m] = cur[m + NEWH]; cur[m + NEWH] = mine[64 * (m + NEWH)]; }
                    } else {
#pragma unroll
                        for (int m = 0; m < TE; ++m) cur[m] = mine[64 * m];
                    }
                    wave_sync();
#ifndef SPEC_ABL_TEAM_NOWAIT
                    // is the slot free?  The poll was taken one line earlier (this wave's own word, parity i & 1; it landed
                    // before the vm_wait<0> above); "not yet" is a blocking wait of this wave alone
                    if (round != 0 && (pending == NONE || (int32_t)(pland[2u * wave + (i & 1u)] - NT * round) < 0) &&
                        !wave_wait(ring + 32 * slot + 16, NT * round, sync)) return;
#endif
                    {
                        const uint32_t word = __builtin_amdgcn_readfirstlane(pland_addr + 8u * wave + 4u * ((i + 1) & 1u));
                        if (lane == 0) glds4_sc1(ring + 32 * ((i + 1) % a.ring) + 16, word);  // before the stores
                    }
                    cx<R> *dst = slots + (uint64_t)slot * N + (uint64_t)ncol * N1;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int m = 0; m < TE; ++m) st_slot<R>(dst + wt + m * PA::T, v[m]);
                    asm volatile("" ::: "memory");  // the loads below stay behind the stores above
                    pending = slot;
                    issue_next(i + 2);
                }
                vm_wait<0>();
                if (lane == 0 && atomicAdd(&s_arrive_slot[pending], 1u) + 1 == WAVES * ((my_lines - 1) / a.ring + 1))
                    __hip_atomic_fetch_add(ring + 32 * pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        if constexpr (DIRECT && LD::PIPE) {
            if (win == nullptr) {
                // ---- pipelined form: samples are cx<double> in memory, no window ----------------------------------
                // `cur` (registers) holds line i.  The rows of line i + 1 that `cur` lacks (at 50 % overlap the upper
                // half: the lower half of line i + 1 is the upper half of line i) land in the strips, requested one
                // whole line earlier and issued BEHIND line i - 1's stores, so that the counted wait for those stores
                // leaves them in flight.  Per line: [poll slot of line i + 1] [stores of line i] [loads of line i + 2].
                constexpr int NLD = (HALF ? NEWH : TE) / EPL;  // LOWER bound of the loads issued behind a line's stores
                const DmaCoords dc = dma_coords(PA::C);
                auto issue_rows = [&](uint32_t line, auto first_tag, auto end_tag) {  // rows [FIRST, END) of the line's tile
                    constexpr int FIRST = decltype(first_tag)::value, END = decltype(end_tag)::value;
                    static_assert(FIRST % EPL == 0 && END % EPL == 0, "whole instructions");
                    const uint8_t *src = a.iq + (uint64_t)line * a.hop * sizeof(cx<R>);
#pragma unroll
                    for (int j = FIRST / EPL; j < END / EPL; ++j)
                        glds16<0>(src + (uint64_t)((dc.t + (EPL * j + dc.mm) * PA::T) * N2 + c0 + dc.q) * sizeof(cx<R>), land_addr + 1024u * j);
                };
                auto issue = [&](uint32_t line, auto first_tag) { issue_rows(line, first_tag, std::integral_constant<int, TE>{}); };
                // at the start of a block of lines all eight rows are requested (the counted wait then also waits for
                // the first four of them, once per block).  The NLD loads the counted wait relies on -- the upper half
                // at 50 % overlap, all rows otherwise -- are issued UNCONDITIONALLY, the lower half in front of them
                // where it is needed: every control-flow path from a line's stores to the wait carries at least NLD
                // younger loads (tools/check_team_handoff.py R1 walks the paths of the ISA)
                auto issue_next = [&](uint32_t i_next) {
                    const uint32_t ic = i_next < my_lines ? i_next : my_lines - 1;  // tail: a valid line again, unused
                    const uint32_t ln = line_of(ic);
                    if constexpr (HALF) {
                        if (!follows(ic)) issue_rows(ln, std::integral_constant<int, 0>{}, std::integral_constant<int, NEWH>{});
                        issue_rows(ln, std::integral_constant<int, NEWH>{}, std::integral_constant<int, TE>{});
                    } else {
                        issue(ln, std::integral_constant<int, 0>{});
                    }
                };
                cx<R> cur[TE];
                issue(line_of(0), std::integral_constant<int, 0>{});
                vm_wait<0>();  // first line: nothing to overlap with yet
#pragma unroll
                for (int m = 0; m < TE; ++m) cur[m] = land[64 * m + lane];
                issue_next(1);
                __syncthreads();  // table visible
                uint32_t pending = NONE;  // ring slot whose stores are issued but not yet announced
                PROF_DECL;
#ifdef SPEC_TEAM_PROF
                const unsigned long long pf_begin = __builtin_readcyclecounter();
#endif
                for (uint32_t i = 0; i < my_lines; ++i) {
                    const uint32_t slot = i % a.ring, round = i / a.ring;
                    TRACE(0, i);
                    cx<R> v[TE];
#pragma unroll
                    for (int m = 0; m < TE; ++m) v[m] = cur[m];
#ifndef SPEC_ABL_TEAM_NOFFT
                    dft8(v);
#endif
                    xstore0<R>(v, t0, lds + (size_t)q0 * PA::SL);
                    // the previous line's stores (and the poll before them) have had this long: wait for them, not
                    // for the NLD loads issued behind them; the last wave to see its stores done announces the line
                    PROF_T0();
                    vm_wait<NLD>();
                    if (lane == 0 && pending != NONE && atomicAdd(&s_arrive, 1u) + 1 == WAVES * i)
                        __hip_atomic_fetch_add(ring + 32 * pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    PROF_ADD(1);
                    TRACE(1, i);
                    PROF_T0();
                    __syncthreads();
                    PROF_ADD(4);
                    TRACE(2, i);
                    // was the slot free when the poll was taken?  (first lines: nobody has used it yet.)  The poll of
                    // line i landed in word i & 1 before wave 0 came to the barrier above; that word is rewritten two
                    // lines on
                    const bool slot_free = round == 0 || (pending != NONE && (int32_t)(pland[32 * (i & 1u)] - NT * round) >= 0);
                    // Thread roles change here: from now on the T threads of a column are neighbours in one wave, and a
                    // wave reads and writes only the rows of its own columns -- the second exchange needs no barrier
                    xload<R, L1, WG>(v, t1, lds + (size_t)q1 * PA::SL);
                    wave_sync();
#ifndef SPEC_ABL_TEAM_NOFFT
                    if constexpr (TWREG) twr.pass1(v);
                    else pass1<R, L1, WG>(v, t1, tab);
#endif
                    xstore1<R>(v, t1, lds + (size_t)q1 * PA::SL);
                    wave_sync();
                    xload<R, L1, WG>(v, t1, lds + (size_t)q1 * PA::SL);
                    finish(v);
                    TRACE(3, i);
                    // cur <- line i + 1: its rows were requested a whole line ago (behind the last line: unused)
                    PROF_T0();
                    vm_wait<0>();
                    PROF_ADD(2);
                    if (HALF && follows(i + 1 < my_lines ? i + 1 : my_lines - 1)) {
#pragma unroll
                        for (int m = 0; m < NEWH; ++m) { cur[m] = cur[m + NEWH]; cur[m + NEWH] = land[64 * (m + NEWH) + lane]; }
                    } else {
#pragma unroll
                        for (int m = 0; m < TE; ++m) cur[m] = land[64 * m + lane];
                    }
#ifndef SPEC_ABL_TEAM_NOWAIT
                    PROF_T0();
                    if (!slot_free) PROF_INC(5);
                    if (!slot_free && !team_wait(ring + 32 * slot + 16, NT * round, sync, &s_flag)) return;
                    PROF_ADD(3);
#endif
                    TRACE(4, i);
                    if (tid == 0) glds4_sc1(ring + 32 * ((i + 1) % a.ring) + 16, pland_addr + 128u * ((i + 1) & 1u));  // before the stores
                    cx<R> *dst = slots + (uint64_t)slot * N + (uint64_t)n2 * N1;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int m = 0; m < TE; ++m) st_slot<R>(dst + t1 + m * PA::T, v[m]);
                    asm volatile("" ::: "memory");  // the loads below stay behind the stores above
                    pending = slot;
                    issue_next(i + 2);
                    TRACE(5, i);
                    __syncthreads();  // this line's last LDS reads | the next line's first LDS writes
                }
                vm_wait<0>();
                if (lane == 0 && atomicAdd(&s_arrive, 1u) + 1 == WAVES * my_lines)
                    __hip_atomic_fetch_add(ring + 32 * pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SPEC_TEAM_PROF
                pf[0] = __builtin_readcyclecounter() - pf_begin;
#endif
                PROF_OUT(1);
                return;
            }
        }
        // ---- plain form (fp32, decoded formats, windows): one line of input ahead, every line announced at once ----
        {
            constexpr bool half = HALF;
            auto load_rows = [&](uint32_t line, cx<R> (&x)[TE], auto first_tag) {
                constexpr int FIRST = decltype(first_tag)::value;
                const uint8_t *src = a.iq + (uint64_t)line * a.hop * a.bps;
#pragma unroll
                for (int m = FIRST; m < TE; ++m) {
                    const uint32_t n = (uint32_t)(t0 + m * PA::T) * N2 + c0 + q0;
                    if constexpr (DIRECT) x[m] = ld_stream<R>(src + (uint64_t)n * sizeof(cx<R>));
                    else x[m] = decode_sample<R>(src + (uint64_t)n * a.bps, a.kind, a.be != 0);
                }
            };
            cx<R> nxt[TE];
            load_rows(line_of(0), nxt, std::integral_constant<int, 0>{});
            __syncthreads();  // table visible
            for (uint32_t i = 0; i < my_lines; ++i) {
                const uint32_t slot = i % a.ring, round = i / a.ring;
                cx<R> v[TE];
#pragma unroll
                for (int m = 0; m < TE; ++m) v[m] = nxt[m];
                if (i + 1 < my_lines) {  // the next line's rows stay in flight behind this FFT
                    if (half && follows(i + 1)) {
#pragma unroll
                        for (int m = 0; m < NEWH; ++m) nxt[m] = nxt[m + NEWH];
                        load_rows(line_of(i + 1), nxt, std::integral_constant<int, NEWH>{});
                    } else {
                        load_rows(line_of(i + 1), nxt, std::integral_constant<int, 0>{});
                    }
                }
                if (win) {
#pragma unroll
                    for (int m = 0; m < TE; ++m) {
                        const R w = win[(uint32_t)(t0 + m * PA::T) * N2 + c0 + q0];
                        v[m].x *= w;
                        v[m].y *= w;
                    }
                }
#ifndef SPEC_ABL_TEAM_NOFFT
                dft8(v);
#endif
                xstore0<R>(v, t0, lds + (size_t)q0 * PA::SL);
                __syncthreads();
                xload<R, L1, WG>(v, t1, lds + (size_t)q1 * PA::SL);  // thread roles change here: a column's threads in one wave
                wave_sync();
#ifndef SPEC_ABL_TEAM_NOFFT
                if constexpr (TWREG) twr.pass1(v);
                else pass1<R, L1, WG>(v, t1, tab);
#endif
                xstore1<R>(v, t1, lds + (size_t)q1 * PA::SL);
                wave_sync();
                xload<R, L1, WG>(v, t1, lds + (size_t)q1 * PA::SL);
                finish(v);
#ifndef SPEC_ABL_TEAM_NOWAIT
                if (!team_wait(ring + 32 * slot + 16, NT * round, sync, &s_flag)) return;
#else
                __syncthreads();
#endif
                cx<R> *dst = slots + (uint64_t)slot * N + (uint64_t)n2 * N1;
#pragma unroll
                for (int m = 0; m < TE; ++m) st_slot<R>(dst + t1 + m * PA::T, v[m]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are in L2
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(ring + 32 * slot, APT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        // ================= row side: tile of C rows k1, N2-point transforms over n2, epilogue ===============
#ifdef SPEC_ABL_TEAM_NOB
        return;
#endif
        const uint32_t r0 = (member - NT) * PB::C;
        const int q0 = tid % PB::C, t0 = tid / PB::C;  // rows k1 fastest: the slot reads and the final stores
        cx<R> *tab = lds + (size_t)PB::C * PB::SL;
        for (int e = tid; e < N2; e += WG) tab[e] = static_cast<const cx<R> *>(a.tw2)[e];
        __syncthreads();
        cx<R> *line_lds = lds + (size_t)q0 * PB::SL;
        PassTw<R, L2, WG> twr;
        constexpr bool TWREG = !DENSE;
        if constexpr (TWREG) twr.load(t0, tab);
        // The whole walk over this workgroup's lines is specialised on the output format: the format is chosen ONCE,
        // below, and the line loop's stores are a straight-line sequence -- the counted waits of the pipelined form
        // count exactly those stores, and tools/check_team_handoff.py can follow every path of the loop in the ISA
        // (with the format switch inside the loop "no format at all" was a path of the control-flow graph).
        auto row_side = [&](auto out_fmt_tag) {
        constexpr int OUTFMT = decltype(out_fmt_tag)::value;
        // the epilogue of one line (SS:76-82) and its stores; between(4) sits in front of the first store
        auto epilogue = [&](cx<R> (&v)[TE], uint32_t line, auto between) {
            const uint64_t base = (uint64_t)line * N;
            auto emit = [&](auto fmt_tag) {  // one format per call: the branch on the format is outside the bins
                constexpr int FMT = decltype(fmt_tag)::value;
                using TO = std::conditional_t<(FMT >= OUT_DB20_F64), double, float>;
#if defined(SPEC_ABL_TEAM_NOSTORE)
                between(std::integral_constant<int, 4>{});
#pragma unroll
                for (int m = 0; m < TE; ++m) {
                    const uint32_t k = (r0 + q0) + (uint32_t)N1 * (t0 + m * PB::T);
                    emit_bin<R, FMT>(a.out, base + ((k + N / 2) & (N - 1)), v[m], s_dbt);  // SS:78
                }
#else
                TO *out = static_cast<TO *>(a.out);
                // the epilogue in two halves: the first half's stores go out while the second half is computed (the
                // ring poll in front of the first store: every store of the line stays younger than it); 1-2 % over
                // all eight bins first
                constexpr int NH = 2;
                constexpr int HB = TE / NH;  // bins per part
                const bool odd = (q0 & 1) != 0;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    TO d[HB];
                    cx<R> z[HB];
#pragma unroll
                    for (int m = 0; m < HB; ++m) z[m] = v[h * HB + m];
#ifndef SPEC_ABL_TEAM_NOEPI
                    if constexpr (FMT == OUT_DB20_F64 && sizeof(R) == 8) {
                        db20_tab_n<HB>(z, s_dbt, d);
                    } else
#endif
                    if constexpr (FMT == OUT_DB20_F32 && sizeof(R) == 4) {
                        // fp32 lines: ONE range test per thread and half (as the packed family's epilogue, spec_v2.h): while
                        // every |X|^2 is in [1e-4, 1e37), |X| + 1e-10 == |X| in fp32 and the value is 10 log10(p) -- one
                        // v_log_f32 per bin; otherwise the exact form bin by bin (a bin inside the range gets the same value
                        // either way).  Per bin the exact form alone is seven compares, a square root and the selects.
                        float p[HB];
#pragma unroll
                        for (int m = 0; m < HB; ++m) p[m] = __builtin_fmaf(z[m].x, z[m].x, z[m].y * z[m].y);
                        float lo = p[0], hi = p[0];
#pragma unroll
                        for (int m = 1; m < HB; ++m) { lo = fminf(lo, p[m]); hi = fmaxf(hi, p[m]); }
                        constexpr float k10 = 3.01029995663981195f;  // 10 log10(2)
                        if (lo > 1e-4f && hi < 1e37f) {  // a NaN fails the first test
#pragma unroll
                            for (int m = 0; m < HB; ++m) d[m] = k10 * __log2f(p[m]);
                        } else {
#pragma unroll
                            for (int m = 0; m < HB; ++m) d[m] = db20(z[m]);
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < HB; ++m) d[m] = (TO)bin_value<R, FMT>(z[m], s_dbt);
                    }
#ifdef SPEC_TEAM_SINGLE_STORES
                    // one store per bin: a thread holds bin k1 = r0 + q0 of eight rows k2 (the C lanes of a row k2 write one run)
                    asm volatile("" ::: "memory");
                    if (h == 0) between(std::integral_constant<int, 4>{});
#pragma unroll
                    for (int m = 0; m < HB; ++m) {
                        const uint32_t k = (r0 + q0) + (uint32_t)N1 * (t0 + (h * HB + m) * PB::T);
                        __builtin_nontemporal_store(d[m], out + base + ((k + N / 2) & (N - 1)));  // SS:78
                    }
#else
                    // A thread holds bin k1 = r0 + q0 of eight rows k2; its neighbour (lane ^ 1) holds k1 ^ 1 of the same
                    // rows.  Even lanes store {k1, k1 + 1} of the rows m = 0, 2, 4, 6, odd lanes {k1 - 1, k1} of the rows
                    // m = 1, 3, 5, 7: TE / 2 stores of two bins each instead of TE stores of one.
                    TO o[HB];
#pragma unroll
                    for (int m = 0; m < HB; ++m) o[m] = lane_swap1(d[m]);
                    asm volatile("" ::: "memory");
                    if (h == 0) between(std::integral_constant<int, 4>{});
#pragma unroll
                    for (int pr = 0; pr < HB / 2; ++pr) {
                        const int me = 2 * pr, mo = 2 * pr + 1;
                        const uint32_t k2 = (uint32_t)t0 + (uint32_t)(h * HB + (odd ? mo : me)) * PB::T;
                        const uint32_t k = (r0 + ((uint32_t)q0 & ~1u)) + (uint32_t)N1 * k2;  // the even bin of the pair
                        st_pair<TO>(out + base + ((k + N / 2) & (N - 1)), odd ? o[mo] : d[me], odd ? d[mo] : o[me]);  // SS:78
                    }
#endif
                }
#endif
            };
            asm volatile("" ::: "memory");
            emit(std::integral_constant<int, OUTFMT>{});
            asm volatile("" ::: "memory");
        };
        // The rest of one line behind its first pass and exchange.
        // `between(stage)`: the caller's requests for the next tile, a part of them in front of each stretch of arithmetic
        // instead of all at once (eight 1 KiB requests per wave keep the texture addresser busy for ~130 cycles per wave,
        // ~1000 per workgroup, during which nobody computed); stage 4: between the epilogue's arithmetic and its stores
        // (the caller's poll of the ring: as late as the order of the counted waits allows).
        // Plain form: both exchanges in the line buffers, four barriers per line, stages 0 .. 3.
        auto rest_of_line = [&](cx<R> (&v)[TE], uint32_t line, auto between) {
            between(std::integral_constant<int, 0>{});
            PROF_PH(2);  // hand-back, first requests for the next tile
            xload<R, L2, WG>(v, t0, line_lds);
            __syncthreads();
            PROF_PH(3);
            between(std::integral_constant<int, 1>{});
#ifndef SPEC_ABL_TEAM_NOFFT
            if constexpr (TWREG) twr.pass1(v);
            else pass1<R, L2, WG>(v, t0, tab);
#endif
            between(std::integral_constant<int, 2>{});
            xstore1<R>(v, t0, line_lds);
            __syncthreads();
            PROF_PH(4);
            between(std::integral_constant<int, 3>{});
            xload<R, L2, WG>(v, t0, line_lds);
#ifndef SPEC_ABL_TEAM_NOFFT
            if constexpr (TWREG) twr.pass2(v);
            else pass2<R, L2, WG>(v, t0, tab);
#endif
            PROF_PH(5);
            epilogue(v, line, between);
            PROF_PH(6);
            __syncthreads();  // the line buffers are rewritten by the next line's first exchange
            PROF_PH(7);
        };
        // Pipelined form: the FIRST exchange lives in the landing strips.  After pass 0 the threads of wave w hold,
        // of every row q of the tile, the G = 512 / C consecutive elements [w G, (w + 1) G): C rows x G elements =
        // 512 elements, exactly the wave's own strip -- so every wave WRITES only its own strip (which it has just
        // read: no barrier in front), and after the one barrier everybody reads across the strips.  The second
        // exchange keeps the line buffers.  Two barriers per line instead of four (none between a line's last LDS read
        // and the next line's first LDS write: those are different buffers now); the price is that the next tile may
        // be requested only behind the second barrier (everybody has read the strips), stages 3, 5, 6.
        // Element g of row q sits at [q G + (g ^ (q & 15))]: 16 lanes of one row-major access hit 16 different
        // 16-byte slots on the write (g fixed per instruction) and on the read side alike.
        constexpr uint32_t SXG = 8u * 64u / PB::C, SXR = SXG / PB::T;
        static_assert(SXG == PB::T * SXR && (SXR == 1 || SXR == 2) && SXG >= 16, "strip exchange: a row's share per wave");
        const uint32_t sx_qs = (uint32_t)q0 & 15u;
        cx<R> *sx_wr = land + (size_t)q0 * SXG;                                       // own strip, row q0
        const uint32_t sx_b8 = (8u * ((uint32_t)t0 % (64u / PB::C))) ^ sx_qs;          // (8 tl + r) ^ qs == sx_b8 ^ r
        const cx<R> *sx_rd = reinterpret_cast<const cx<R> *>(smem + 256) + (size_t)q0 * SXG + ((uint32_t)t0 ^ sx_qs);
        auto xstore0s = [&](const cx<R> (&v)[TE]) {
#pragma unroll
            for (int r = 0; r < 8; ++r) sx_wr[sx_b8 ^ (uint32_t)r] = v[r];
        };
        auto xload0s = [&](cx<R> (&v)[TE]) {  // element t0 + m T of row q0: written by wave m / SXR
#pragma unroll
            for (int m = 0; m < TE; ++m) v[m] = sx_rd[(m / (int)SXR) * (TE * 64) + (int)PB::T * (m % (int)SXR)];
        };
        auto rest_of_line_sx = [&](cx<R> (&v)[TE], uint32_t line, auto between) {
            PROF_PH(2);
            xload0s(v);
            PROF_PH(3);
#ifndef SPEC_ABL_TEAM_NOFFT
            if constexpr (TWREG) twr.pass1(v);
            else pass1<R, L2, WG>(v, t0, tab);
#endif
            xstore1<R>(v, t0, line_lds);
            __syncthreads();  // everybody has read the strips and written the line buffers
            PROF_PH(4);
            between(std::integral_constant<int, 3>{});
            xload<R, L2, WG>(v, t0, line_lds);
            between(std::integral_constant<int, 5>{});
#ifndef SPEC_ABL_TEAM_NOFFT
            if constexpr (TWREG) twr.pass2(v);
            else pass2<R, L2, WG>(v, t0, tab);
#endif
            between(std::integral_constant<int, 6>{});
            PROF_PH(5);
            epilogue(v, line, between);
            PROF_PH(6);
        };
        (void)xstore0s; (void)rest_of_line_sx; (void)rest_of_line;
        auto ready = [&](uint32_t j, uint32_t w) { return j < my_lines && (int32_t)(w - NT * APT * (j / a.ring + 1)) >= 0; };
        if constexpr (LD::PIPE) {
            // ---- pipelined form -----------------------------------------------------------------------------------------
            // Per line: [tile of line i + 1 into the strips, if the column side has it (poll taken one line earlier)]
            // [poll for line i + 2] [the 8 output stores of line i].  The tile is read at the top of the next
            // iteration behind vm_wait<8>: the loads and the poll are older than the stores, the stores stay in flight.
            const DmaCoords dc = dma_coords(PB::C);
            auto issue_part = [&](uint32_t i, auto from_tag, auto to_tag) {
                static_assert(decltype(from_tag)::value % EPL == 0 && decltype(to_tag)::value % EPL == 0, "whole instructions");
                const cx<R> *src = slots + (uint64_t)(i % a.ring) * N + r0 + dc.q;
#pragma unroll
                for (int j = decltype(from_tag)::value / EPL; j < decltype(to_tag)::value / EPL; ++j)
                    glds16<1>(src + (uint64_t)(dc.t + (EPL * j + dc.mm) * PB::T) * N1, land_addr + 1024u * j);  // [n2][k1]
            };
            auto issue = [&](uint32_t i) { issue_part(i, std::integral_constant<int, 0>{}, std::integral_constant<int, TE>{}); };
            auto issue_poll = [&](uint32_t j) {  // doneA of line j's slot (lane 0)
                if (tid == 0 && j < my_lines) glds4_sc1(ring + 32 * (j % a.ring), pland_addr);
            };
#ifndef SPEC_ABL_TEAM_NOWAIT
            if (!team_wait(ring, NT * APT, sync, &s_flag)) return;
#endif
            issue(0);
            issue_poll(1);
            vm_wait<0>();  // the first tile: nothing to overlap with yet
            PROF_DECL;
#ifdef SPEC_TEAM_PROF
            const unsigned long long pf_begin = __builtin_readcyclecounter();
#endif
            for (uint32_t i = 0; i < my_lines; ++i) {
                const uint32_t slot = i % a.ring;
#ifdef SPEC_TEAM_PROF
                ph_t = __builtin_readcyclecounter();
#endif
                TRACE(0, i);
                PROF_T0();
                vm_wait<TEAM_NST>();  // tile i and the poll have landed; the stores of line i - 1 fly on
                PROF_ADD(1);
                TRACE(1, i);
                cx<R> v[TE];
#pragma unroll
                for (int m = 0; m < TE; ++m) v[m] = land[64 * m + lane];
#ifndef SPEC_ABL_TEAM_NOFFT
                dft8(v);
#endif
#ifdef SPEC_TEAM_NO_STRIPX
                xstore0<R>(v, t0, line_lds);
#else
                xstore0s(v);  // into this wave's own strip (just read): no barrier in front
#endif
                if (tid == 0) s_next = ready(i + 1, pland[0]);
                PROF_PH(0);
                __syncthreads();
                PROF_PH(1);
                // every thread has consumed its slot reads: hand the slot back before the rest of the transform
                if (tid == 0) __hip_atomic_fetch_add(ring + 32 * slot + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SPEC_ABL_TEAM_NOWAIT
                const bool ahead = i + 1 < my_lines;
#else
                const bool ahead = s_next != 0;
#endif
                TRACE(2, i);
                if (ahead) { TRACE(7, i); }
                auto requests = [&](auto stage) {  // every request is older than the line's output stores
                    constexpr int ST = decltype(stage)::value, Q = TE / 4;
                    if constexpr (ST == 4) {
                        // has the column side stored line i + 2?  Asked as late as possible: the answer is read at the
                        // top of the next line, and a "not yet" costs that line a blocking wait and an exposed tile read
                        issue_poll(i + 2);
                    } else if constexpr (ST < 4) {
#ifdef SPEC_TEAM_NO_STRIPX
                        if (ahead) issue_part(i + 1, std::integral_constant<int, ST * Q>{}, std::integral_constant<int, ST * Q + Q>{});
#else
                        // strip exchange: nothing may land before everybody has read the strips (stage 3 is the first one
                        // behind that barrier): half of the tile there, a quarter each behind the next two stretches
                        if constexpr (ST == 3) { if (ahead) issue_part(i + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * Q>{}); }
#endif
                    } else if constexpr (ST == 5) {
                        if (ahead) issue_part(i + 1, std::integral_constant<int, 2 * Q>{}, std::integral_constant<int, 3 * Q>{});
                    } else if constexpr (ST == 6) {
                        if (ahead) issue_part(i + 1, std::integral_constant<int, 3 * Q>{}, std::integral_constant<int, 4 * Q>{});
                    }
                };
#ifdef SPEC_TEAM_NO_STRIPX
                rest_of_line(v, line_of(i), requests);
#else
                rest_of_line_sx(v, line_of(i), requests);
#endif
                TRACE(4, i);
                if (!ahead && i + 1 < my_lines) {  // the column side is not ahead: wait for it here
                    PROF_T0();
                    PROF_INC(5);
#ifndef SPEC_ABL_TEAM_NOWAIT
                    if (!team_wait(ring + 32 * ((i + 1) % a.ring), NT * APT * ((i + 1) / a.ring + 1), sync, &s_flag)) return;
#endif
                    PROF_ADD(3);
                    TRACE(5, i);
                    PROF_T0();
                    issue(i + 1);
                    vm_wait<0>();  // nothing younger to leave in flight on this path
                    PROF_ADD(2);
                    TRACE(6, i);
                }
            }
#ifdef SPEC_TEAM_PROF
            pf[0] = __builtin_readcyclecounter() - pf_begin;
#endif
            PROF_OUT(2);
        } else {
            // ---- plain form (fp32): the next tile requested as soon as the column side has it, loads the compiler knows
            auto load_tile = [&](uint32_t i, cx<R> (&x)[TE]) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    slots + (uint64_t)(i % a.ring) * N, 0, (uint32_t)(N * sizeof(cx<R>)), 0x00020000);
#pragma unroll
                for (int m = 0; m < TE; ++m)  // [n2][k1]
                    x[m] = ld_slot<R>(rs, (int)(((uint32_t)(t0 + m * PB::T) * N1 + r0 + q0) * sizeof(cx<R>)));
            };
            cx<R> nxt[TE];
#ifndef SPEC_ABL_TEAM_NOWAIT
            if (!team_wait(ring, NT * APT, sync, &s_flag)) return;
#endif
            load_tile(0, nxt);
            for (uint32_t i = 0; i < my_lines; ++i) {
                const uint32_t slot = i % a.ring;
                cx<R> v[TE];
#pragma unroll
                for (int m = 0; m < TE; ++m) v[m] = nxt[m];
#ifndef SPEC_ABL_TEAM_NOFFT
                dft8(v);
#endif
                xstore0<R>(v, t0, line_lds);
                if (tid == 0) s_next = ready(i + 1, ld_sc1(ring + 32 * ((i + 1) % a.ring)));
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(ring + 32 * slot + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ahead = s_next != 0;
                if (ahead) load_tile(i + 1, nxt);
                rest_of_line(v, line_of(i), [](auto) {});
                if (!ahead && i + 1 < my_lines) {
#ifndef SPEC_ABL_TEAM_NOWAIT
                    if (!team_wait(ring + 32 * ((i + 1) % a.ring), NT * APT * ((i + 1) / a.ring + 1), sync, &s_flag)) return;
#endif
                    load_tile(i + 1, nxt);
                }
            }
        }
        };  // row_side
        if constexpr (sizeof(R) == 4) {  // an fp32 pipeline only ever stores floats (spec_capi.hip run_lines)
            if (a.out_fmt == OUT_DB20_F32) row_side(std::integral_constant<int, OUT_DB20_F32>{});
            else row_side(std::integral_constant<int, OUT_POW_F32>{});
        } else {
            switch (a.out_fmt) {
            case OUT_DB20_F32: row_side(std::integral_constant<int, OUT_DB20_F32>{}); break;
            case OUT_POW_F32: row_side(std::integral_constant<int, OUT_POW_F32>{}); break;
            case OUT_DB20_F64: row_side(std::integral_constant<int, OUT_DB20_F64>{}); break;
            default: row_side(std::integral_constant<int, OUT_POW_F64>{}); break;
            }
        }
    }
}

template <typename R, int L1, int L2, int WG, bool DENSE> constexpr size_t team_lds_bytes() { return TeamLds<R, L1, L2, WG, DENSE>::BYTES; }

template <typename R, int L1, int L2, int WG, bool DENSE>
hipError_t launch_team_wg(const TeamArgs &a, int n_cu, uint32_t *teams_max, hipStream_t s, bool query_only) {
    constexpr size_t lds = team_lds_bytes<R, L1, L2, WG, DENSE>();
    const bool direct = !a.be && a.kind == (sizeof(R) == 8 ? K_CF64 : K_CF32);
    const bool half = (uint64_t)a.hop * 2 == ((uint64_t)1 << (L1 + L2));
    auto fn = direct ? (half ? &large_team_kernel<R, L1, L2, true, true, WG, DENSE> : &large_team_kernel<R, L1, L2, true, false, WG, DENSE>)
                     : (half ? &large_team_kernel<R, L1, L2, false, true, WG, DENSE> : &large_team_kernel<R, L1, L2, false, false, WG, DENSE>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, WG, lds);
    if (e != hipSuccess) return e;
    // lines in flight per XCD are bounded by what its L2 holds: one 512-thread workgroup per CU, two when DENSE
    constexpr int WANT = (DENSE ? 1024 : 512) / WG;
    if (per_cu > WANT) per_cu = WANT;
    if (per_cu < 1) return hipErrorLaunchOutOfResources;
    const uint32_t grid = (uint32_t)per_cu * (uint32_t)n_cu;
    constexpr uint32_t TEAM = 2 * ((1u << (L1 + L2)) / (WG * TE));
    *teams_max = grid / TEAM;
    if (query_only) return hipSuccess;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(WG), lds, s, a);
    return hipGetLastError();
}
template <typename R, int L1, int L2>
hipError_t launch_team(const TeamArgs &a, int wg, int n_cu, uint32_t *teams_max, hipStream_t s, bool query_only) {
    // 512: one workgroup per CU, 16-wide tiles -- the only geometry the default dispatch reaches, and the only one the
    // product library carries.  The two experiment geometries (measured slower) are compiled only with
    // -DSPEC_TEAM_VARIANTS (python -m spectral_analyzer_amd.build --variant teamvar; tests/test_gpu_large.py runs them
    // from that library): 256 = two workgroups per CU, 8-wide tiles, one of either role on every CU; 1024 = the dense
    // form, two 512-thread workgroups per CU at 128 registers.
#ifdef SPEC_TEAM_VARIANTS
    if (wg == 1024) return launch_team_wg<R, L1, L2, 512, true>(a, n_cu, teams_max, s, query_only);
    if (wg == 256) return launch_team_wg<R, L1, L2, 256, false>(a, n_cu, teams_max, s, query_only);
#else
    if (wg != 512) return hipErrorNotSupported;
#endif
    return launch_team_wg<R, L1, L2, 512, false>(a, n_cu, teams_max, s, query_only);
}

}  // namespace

#ifdef SPEC_TEAM_PROF
size_t large_team_sync_bytes() { return (size_t)TEAM_SYNC_WORDS * sizeof(uint32_t) + 1024 * 128 + 64 * 32 * 8 * 8; }  // + 16 words x 1024 workgroups + the event trace
#else
size_t large_team_sync_bytes() { return (size_t)TEAM_SYNC_WORDS * sizeof(uint32_t); }
#endif
uint32_t large_team_prof_offset_bytes() { return TEAM_SYNC_WORDS * (uint32_t)sizeof(uint32_t); }
uint32_t large_team_abort_word() { return TS_ABORT; }

// Launches the team kernel over all n_lines (one launch).  `sync` must be zeroed (large_team_sync_bytes()) on
// the same stream before the call; sync[16] != 0 afterwards means a wait timed out and the output is not
// complete.  `scratch` holds teams_max * ring * N complex values (query with query_only first).
hipError_t launch_spectro_team(const WfArgs &w, int log2n, bool f64, const void *tw1, const void *tw2, void *scratch,
                               uint32_t ring, uint32_t *sync, int n_cu, uint32_t *teams_max, bool query_only,
                               hipStream_t s, int wg, uint32_t block) {
    if (wg != 256 && wg != 512 && wg != 1024) return hipErrorInvalidValue;
    TeamArgs a{};
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.bps = w.bps; a.kind = w.kind; a.be = w.be;
    a.tw1 = tw1; a.tw2 = tw2; a.twn = w.tw; a.win = w.win; a.scratch = scratch; a.out = w.out; a.out_fmt = w.out_fmt;
    a.ring = ring < 1 ? 1 : (ring > (uint32_t)TEAM_RING_MAX ? (uint32_t)TEAM_RING_MAX : ring);
    a.block = block;
    a.sync = sync;
    if (f64) {
        switch (log2n) {
        case 14: return launch_team<double, 7, 7>(a, wg, n_cu, teams_max, s, query_only);
        case 15: return launch_team<double, 7, 8>(a, wg, n_cu, teams_max, s, query_only);
        case 16: return launch_team<double, 8, 8>(a, wg, n_cu, teams_max, s, query_only);
        }
    } else {
        switch (log2n) {
        case 15: return launch_team<float, 7, 8>(a, wg, n_cu, teams_max, s, query_only);
        case 16: return launch_team<float, 8, 8>(a, wg, n_cu, teams_max, s, query_only);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace specgpu
