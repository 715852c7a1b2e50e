// spec_k_team.hip -- four-step FFT for lines longer than the LDS holds (fp32: nfft >= 32768, fp64:
// nfft >= 16384; BASELINE configs[4] is 65536-point cf64) with the intermediate kept in the XCD's L2.
//
// spec_k_large.hip runs the two halves of the decomposition N = N1 N2 (n = N2 n1 + n2, k = k1 + N1 k2) as two
// launches with the [n2][k1] intermediate in HBM: three bytes moved per algorithmic byte.  Here ONE persistent
// launch keeps that intermediate inside one XCD's 4 MiB L2:
//
//   * the launch is sized to be fully resident (<= 2 workgroups of 256 threads per CU).  Every workgroup reads
//     the id of the XCD it actually runs on (HW_REG_XCC_ID) and takes a ticket on that XCD; after a one-time
//     registration wait the workgroups of one XCD form TEAMS of 2 NT members (NT = N / 2048 tiles per line and
//     step), NT "column" workgroups and NT "row" workgroups.  Teams are built from the XCD ids the hardware
//     reports, never from blockIdx, so the result does not depend on how the dispatcher places workgroups --
//     a different placement only changes who is in which team; workgroups left over on an XCD exit.
//   * a team owns a contiguous range of lines and a ring of RING line-sized slots of intermediate.  Column
//     workgroup c: 8-wide tile of columns n2, N1-point FFTs over n1 (input rows N2 samples apart), times
//     W_N^(n2 k1), into slot[line % RING] as [n2][k1]; it keeps its registers across lines, so at 50 % overlap
//     half of the next tile is a register move, and the next line's rows are requested before the current FFT.
//     Row workgroup r: 8-wide tile of rows k1, N2-point FFTs over n2, epilogue (SS:76-82), X[k1 + N1 k2] out.
//   * flow control, per team and ring slot: doneA counts column tiles stored, doneB row tiles read.  A row
//     workgroup starts line i when doneA == NT (i / RING + 1); a column workgroup may overwrite the slot for
//     line i when doneB == NT (i / RING).  No full barrier: the column side runs up to RING lines ahead.
//   * visibility inside the XCD: the column side's plain stores are write-through in the CU's L1 and land in
//     the XCD's L2; each storing wave waits vmcnt(0), the workgroup barriers, one lane adds to doneA (agent-scope
//     atomic).  The row side polls doneA with sc1 loads and reads the slot with sc1 loads (L1 bypass, L2 served;
//     MI355X_MICROARCH "Workgroup dispatch, XCD placement & inter-workgroup visibility").  The slot is
//     rewritten every RING lines and stays dirty in L2: it costs L2 bandwidth, not HBM bandwidth.
//   * every spin is bounded (wall clock): on a timeout -- the grid was not co-resident, e.g. the GPU is
//     shared -- the workgroup raises the abort word and every workgroup leaves; the host then runs the
//     two-launch path of spec_k_large.hip (guarded kernels that start only when the abort word is set).
#include <type_traits>

#include "spec_kernels.h"

namespace specgpu {

namespace {

constexpr int TEAM_WG = 256;  // threads per workgroup
constexpr int TE = 8;         // points per thread
constexpr int TEAM_RING_MAX = 4;
constexpr long long TEAM_SPIN_LIMIT = 200000000ll;  // wall_clock64 ticks (100 MHz): 2 s

// sub-transform of 2^L points by T = 2^L / 8 threads, radices 8 x 8 x (M / 64)
template <int L> struct TP {
    static_assert(L == 7 || L == 8, "sub-transforms of 128 or 256 points");
    static constexpr int M = 1 << L, T = M / TE, C = TEAM_WG / T;  // C sub-transforms (columns / rows) per tile
    static constexpr int R2 = M / 64, S2 = TE / R2;
    static constexpr int SL = M + 1;  // LDS line stride in elements (odd: adjacent lines start in adjacent slots)
};

// words of the synchronisation block (uint32 each; the host zeroes it before every launch)
enum : uint32_t {
    TS_TOTAL = 0,       // workgroups registered
    TS_ABORT = 16,      // set by a workgroup whose wait timed out (own 64-byte line)
    TS_XCC = 32,        // [8] tickets per XCD, 16 words apart
    TS_RING = 32 + 8 * 16,  // per team and ring slot: doneA, doneB (16 words apart)
};
constexpr uint32_t TEAM_MAX_TEAMS = 64;
constexpr uint32_t TEAM_SYNC_WORDS = TS_RING + TEAM_MAX_TEAMS * TEAM_RING_MAX * 2 * 16;

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

template <typename R> __device__ __forceinline__ void ctw(cx<R> &u, const cx<R> w) { u = cmul(u, w); }

// passes 1 and 2 of the 8 x 8 x R2 plan on the registers of butterfly index t (pass 0 is a bare dft8)
template <typename R, int L> __device__ __forceinline__ void pass1(cx<R> (&v)[TE], int t, const cx<R> *tab) {
    const int k = t & 7;
#pragma unroll
    for (int r = 1; r < 8; ++r) ctw(v[r], tab[r * k * (TP<L>::M / 64)]);
    dft8(v);
}
template <typename R, int L> __device__ __forceinline__ void pass2(cx<R> (&v)[TE], int t, const cx<R> *tab) {
    using P = TP<L>;
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
template <typename R, int L> __device__ __forceinline__ void xload(cx<R> (&v)[TE], int t, const cx<R> *line) {
#pragma unroll
    for (int m = 0; m < TE; ++m) v[m] = line[t + m * TP<L>::T];
}

typedef uint32_t tu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t tu32x2 __attribute__((ext_vector_type(2)));
// one cx<R> from the ring slot with an sc1 load: served by the XCD's L2, never by this CU's L1
template <typename R> __device__ __forceinline__ cx<R> ld_slot(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    constexpr int SC1 = 16;
    if constexpr (sizeof(R) == 8) {
        const tu32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, SC1);
        return cx<R>{__longlong_as_double((long long)(((uint64_t)u.y << 32) | u.x)),
                     __longlong_as_double((long long)(((uint64_t)u.w << 32) | u.z))};
    } else {
        const tu32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, SC1);
        return cx<R>{__uint_as_float(u.x), __uint_as_float(u.y)};
    }
}

// one cx<R> of the recording, read once: non-temporal
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

template <typename R, int L1, int L2, bool DIRECT>
__global__ __launch_bounds__(TEAM_WG, 2) void large_team_kernel(const TeamArgs a) {
    using PA = TP<L1>;
    using PB = TP<L2>;
    constexpr int N1 = PA::M, N2 = PB::M, N = N1 * N2;
    constexpr uint32_t NT = N / (TEAM_WG * TE);  // tiles per line and step
    static_assert(N2 / PA::C == (int)NT && N1 / PB::C == (int)NT, "tile counts of the two steps match");
    constexpr uint32_t TEAM = 2 * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_flag;
    __shared__ uint32_t s_info[4];
    cx<R> *lds = reinterpret_cast<cx<R> *>(smem);
    const int tid = threadIdx.x;
    uint32_t *sync = a.sync;

    // ---- registration: which XCD am I on, which ticket do I hold there --------------------------------
    if (tid == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const uint32_t ticket = __hip_atomic_fetch_add(sync + TS_XCC + 16 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(sync + TS_TOTAL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_info[0] = xcc;
        s_info[1] = ticket;
    }
    __syncthreads();
    if (!team_wait(sync + TS_TOTAL, gridDim.x, sync, &s_flag)) return;
    // everybody has registered: the tickets per XCD are final
    if (tid == 0) {
        const uint32_t xcc = s_info[0], ticket = s_info[1];
        uint32_t teams_before = 0, teams_total = 0, mine = 0;
        for (uint32_t x = 0; x < 8; ++x) {
            const uint32_t t = ld_sc1(sync + TS_XCC + 16 * x) / TEAM;
            if (x < xcc) teams_before += t;
            if (x == xcc) mine = t;
            teams_total += t;
        }
        const uint32_t local_team = ticket / TEAM;
        s_info[2] = local_team < mine ? teams_before + local_team : 0xFFFFFFFFu;  // left over on this XCD: no team
        s_info[3] = teams_total;
        if (teams_total == 0 || teams_total > TEAM_MAX_TEAMS)  // nobody could form a team: the host falls back
            __hip_atomic_store(sync + TS_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const uint32_t team = s_info[2], n_teams = s_info[3], member = s_info[1] % TEAM;
    if (team == 0xFFFFFFFFu || n_teams == 0 || n_teams > TEAM_MAX_TEAMS) return;
    const uint32_t line_first = (uint32_t)((uint64_t)a.n_lines * team / n_teams);
    const uint32_t line_end = (uint32_t)((uint64_t)a.n_lines * (team + 1) / n_teams);
    const uint32_t my_lines = line_end - line_first;
    cx<R> *slots = static_cast<cx<R> *>(a.scratch) + (uint64_t)team * a.ring * N;
    uint32_t *ring = sync + TS_RING + team * (TEAM_RING_MAX * 32);  // slot s: doneA at 32 s, doneB at 32 s + 16
    const cx<double> *__restrict__ twn = static_cast<const cx<double> *>(a.twn);

    if (member < NT) {
        // ================= column side: tile of C columns, N1-point transforms over n1 =====================
        const uint32_t c0 = member * PA::C;
        const int q0 = tid % PA::C, t0 = tid / PA::C;  // loads: columns fastest (contiguous samples)
        const int t1 = tid % PA::T, q1 = tid / PA::T;  // stores: k1 fastest (contiguous intermediate)
        cx<R> *tab = lds + (size_t)PA::C * PA::SL;
        for (int e = tid; e < N1; e += TEAM_WG) tab[e] = static_cast<const cx<R> *>(a.tw1)[e];
        const bool half = (uint64_t)a.hop * 2 == (uint64_t)N;  // uniform: 50 % overlap
        const R *__restrict__ win = static_cast<const R *>(a.win);
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
        if (my_lines) load_rows(line_first, nxt, std::integral_constant<int, 0>{});
        __syncthreads();  // table visible
        // inter-step twiddle W_N^(n2 k1), k1 = t1 + m T: W^(n2 t1) (W^(n2 T))^m, recurrence in fp64
        const uint32_t n2 = c0 + q1;
        const cx<double> w0 = twn[n2 * (uint32_t)t1], wstep = twn[n2 * (uint32_t)PA::T];
        for (uint32_t i = 0; i < my_lines; ++i) {
            const uint32_t line = line_first + i, slot = i % a.ring, round = i / a.ring;
            cx<R> v[TE];
#pragma unroll
            for (int m = 0; m < TE; ++m) v[m] = nxt[m];
            if (i + 1 < my_lines) {  // the next line's rows stay in flight behind this FFT
                if (half) {
#pragma unroll
                    for (int m = 0; m < TE / 2; ++m) nxt[m] = nxt[m + TE / 2];
                    load_rows(line + 1, nxt, std::integral_constant<int, TE / 2>{});
                } else {
                    load_rows(line + 1, nxt, std::integral_constant<int, 0>{});
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
            dft8(v);
            xstore0<R>(v, t0, lds + (size_t)q0 * PA::SL);
            __syncthreads();
            xload<R, L1>(v, t1, lds + (size_t)q1 * PA::SL);  // thread roles change here
            __syncthreads();
            pass1<R, L1>(v, t1, tab);
            xstore1<R>(v, t1, lds + (size_t)q1 * PA::SL);
            __syncthreads();
            xload<R, L1>(v, t1, lds + (size_t)q1 * PA::SL);
            pass2<R, L1>(v, t1, tab);
            cx<double> w = w0;
#pragma unroll
            for (int m = 0; m < TE; ++m) {
                const cx<double> z = cmul(cx<double>{(double)v[m].x, (double)v[m].y}, w);
                v[m] = cx<R>{(R)z.x, (R)z.y};
                w = cmul(w, wstep);
            }
            // the slot is free once the row side has read its previous line (the barriers inside also
            // separate this line's last LDS reads from the next line's first LDS writes)
            if (!team_wait(ring + 32 * slot + 16, NT * round, sync, &s_flag)) return;
            cx<R> *dst = slots + (uint64_t)slot * N + (uint64_t)n2 * N1;
#pragma unroll
            for (int m = 0; m < TE; ++m) dst[t1 + m * PA::T] = v[m];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are in L2
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(ring + 32 * slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // ================= row side: tile of C rows k1, N2-point transforms over n2, epilogue ===============
        const uint32_t r0 = (member - NT) * PB::C;
        const int q0 = tid % PB::C, t0 = tid / PB::C;  // rows k1 fastest: the slot reads and the final stores
        cx<R> *tab = lds + (size_t)PB::C * PB::SL;
        for (int e = tid; e < N2; e += TEAM_WG) tab[e] = static_cast<const cx<R> *>(a.tw2)[e];
        __syncthreads();
        cx<R> *line_lds = lds + (size_t)q0 * PB::SL;
        for (uint32_t i = 0; i < my_lines; ++i) {
            const uint32_t line = line_first + i, slot = i % a.ring, round = i / a.ring;
            if (!team_wait(ring + 32 * slot, NT * (round + 1), sync, &s_flag)) return;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                slots + (uint64_t)slot * N, 0, (uint32_t)(N * sizeof(cx<R>)), 0x00020000);
            cx<R> v[TE];
#pragma unroll
            for (int m = 0; m < TE; ++m)  // [n2][k1]
                v[m] = ld_slot<R>(rs, (int)(((uint32_t)(t0 + m * PB::T) * N1 + r0 + q0) * sizeof(cx<R>)), 0);
            dft8(v);
            xstore0<R>(v, t0, line_lds);
            __syncthreads();
            // every thread has consumed its slot reads: hand the slot back before the rest of the transform
            if (tid == 0) __hip_atomic_fetch_add(ring + 32 * slot + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xload<R, L2>(v, t0, line_lds);
            __syncthreads();
            pass1<R, L2>(v, t0, tab);
            xstore1<R>(v, t0, line_lds);
            __syncthreads();
            xload<R, L2>(v, t0, line_lds);
            pass2<R, L2>(v, t0, tab);
            const uint64_t base = (uint64_t)line * N;
#pragma unroll
            for (int m = 0; m < TE; ++m) {
                const uint32_t k = (r0 + q0) + (uint32_t)N1 * (t0 + m * PB::T);
                store_bin<R>(a.out, base + ((k + N / 2) & (N - 1)), v[m], a.out_fmt);  // SS:78
                if constexpr (sizeof(R) == 8) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();  // the line buffers are rewritten by the next line's first exchange
        }
    }
}

template <typename R, int L1, int L2> constexpr size_t team_lds_bytes() {
    constexpr size_t a = ((size_t)TP<L1>::C * TP<L1>::SL + TP<L1>::M) * sizeof(cx<R>);
    constexpr size_t b = ((size_t)TP<L2>::C * TP<L2>::SL + TP<L2>::M) * sizeof(cx<R>);
    return a > b ? a : b;
}

template <typename R, int L1, int L2>
hipError_t launch_team(const TeamArgs &a, int n_cu, uint32_t *teams_max, hipStream_t s, bool query_only) {
    constexpr size_t lds = team_lds_bytes<R, L1, L2>();
    const bool direct = !a.be && a.kind == (sizeof(R) == 8 ? K_CF64 : K_CF32);
    auto fn = direct ? &large_team_kernel<R, L1, L2, true> : &large_team_kernel<R, L1, L2, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, TEAM_WG, lds);
    if (e != hipSuccess) return e;
    if (per_cu > 2) per_cu = 2;  // one column and one row workgroup per CU; more teams would not fit the L2
    if (per_cu < 1) return hipErrorLaunchOutOfResources;
    const uint32_t grid = (uint32_t)per_cu * (uint32_t)n_cu;
    constexpr uint32_t TEAM = 2 * ((1u << (L1 + L2)) / (TEAM_WG * TE));
    *teams_max = grid / TEAM;
    if (query_only) return hipSuccess;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(TEAM_WG), lds, s, a);
    return hipGetLastError();
}

}  // namespace

size_t large_team_sync_bytes() { return (size_t)TEAM_SYNC_WORDS * sizeof(uint32_t); }
uint32_t large_team_abort_word() { return TS_ABORT; }

// Launches the team kernel over all n_lines (one launch).  `sync` must be zeroed (large_team_sync_bytes()) on
// the same stream before the call; sync[16] != 0 afterwards means a wait timed out and the output is not
// complete.  `scratch` holds teams_max * ring * N complex values (query with query_only first).
hipError_t launch_spectro_team(const WfArgs &w, int log2n, bool f64, const void *tw1, const void *tw2, void *scratch,
                               uint32_t ring, uint32_t *sync, int n_cu, uint32_t *teams_max, bool query_only,
                               hipStream_t s) {
    TeamArgs a{};
    a.iq = w.iq; a.n_lines = (uint32_t)w.n_lines; a.hop = w.hop; a.bps = w.bps; a.kind = w.kind; a.be = w.be;
    a.tw1 = tw1; a.tw2 = tw2; a.twn = w.tw; a.win = w.win; a.scratch = scratch; a.out = w.out; a.out_fmt = w.out_fmt;
    a.ring = ring < 1 ? 1 : (ring > (uint32_t)TEAM_RING_MAX ? (uint32_t)TEAM_RING_MAX : ring);
    a.sync = sync;
    if (f64) {
        switch (log2n) {
        case 14: return launch_team<double, 7, 7>(a, n_cu, teams_max, s, query_only);
        case 15: return launch_team<double, 7, 8>(a, n_cu, teams_max, s, query_only);
        case 16: return launch_team<double, 8, 8>(a, n_cu, teams_max, s, query_only);
        }
    } else {
        switch (log2n) {
        case 15: return launch_team<float, 7, 8>(a, n_cu, teams_max, s, query_only);
        case 16: return launch_team<float, 8, 8>(a, n_cu, teams_max, s, query_only);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace specgpu
