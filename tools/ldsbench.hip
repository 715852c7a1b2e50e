// tools/ldsbench.hip -- development aid (round 5): what the LDS STORE path delivers for the exchange shapes of spec_v2.h
// at the residency those kernels run at (one 512-thread workgroup per CU, two waves per SIMD, all eight waves storing at
// once), per instruction form.  Every case moves the same 128 KiB per workgroup and repetition (32 complex values per
// thread); the time is the shader clock (s_memtime) between two workgroup barriers, lane 0 of wave 0 of each workgroup.
//   hipcc -O3 --offload-arch=gfx950 tools/ldsbench.hip -o /tmp/ldsbench && /tmp/ldsbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_p;

constexpr int REPS = 64;

template <int CASE> __global__ __launch_bounds__(512, 2) void k(unsigned long long *out, float seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    v2f v[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) v[m] = v2f{seed * t + m, seed - m};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < REPS; ++rep) {
        if constexpr (CASE == 0) {  // ds_write_b64, lane-consecutive (the row plan's first exchange)
            volatile __attribute__((address_space(3))) v2f *p = (volatile __attribute__((address_space(3))) v2f *)(smem) + t;
#pragma unroll
            for (int m = 0; m < 32; ++m) p[m * 512] = v[m];
        } else if constexpr (CASE == 1) {  // ds_write_b64, lane stride 33 elements (padded first-pass store), single
            volatile __attribute__((address_space(3))) v2f *p = (volatile __attribute__((address_space(3))) v2f *)(smem) + t * 33;
#pragma unroll
            for (int m = 0; m < 32; ++m) p[m] = v[m];
        } else if constexpr (CASE == 2) {  // the same addresses as ds_write2_b64 pairs (what hipcc makes of it)
            v2f *p = reinterpret_cast<v2f *>(smem) + t * 33;
#pragma unroll
            for (int m = 0; m < 32; ++m) p[m] = v[m];
        } else if constexpr (CASE == 3) {  // ds_write_b128: lane stride 34 elements (16-byte aligned rows)
            v4f *p = reinterpret_cast<v4f *>(smem) + t * 17;
#pragma unroll
            for (int m = 0; m < 16; ++m) p[m] = v4f{v[2 * m].x, v[2 * m].y, v[2 * m + 1].x, v[2 * m + 1].y};
        } else if constexpr (CASE == 4) {  // ds_write_b32, lane-consecutive, real and imaginary planes
            volatile __attribute__((address_space(3))) float *p = (volatile __attribute__((address_space(3))) float *)(smem) + t;
#pragma unroll
            for (int m = 0; m < 32; ++m) { p[m * 512] = v[m].x; p[16384 + m * 512] = v[m].y; }
        } else if constexpr (CASE == 5) {  // ds_write_addtid_b32, planes: address = M0 + offset + 4 lane
            const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((t & ~63) * 4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // 16-bit offsets: two M0 bases per plane pair
                asm volatile("s_mov_b32 m0, %0" ::"s"(base + (unsigned)h * 1024u) : "memory");  // (M0[15:0]: the form reaches the first 128 KiB only)
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v[h * 16 + m].x), "n"(m * 2048) : "memory");
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v[h * 16 + m].y), "n"(32768 + m * 2048) : "memory");
                }
            }
        } else if constexpr (CASE == 6) {  // ds_write_b64, the wide-stride store of a later pass: base[r * 32], lanes in runs of 32
            volatile __attribute__((address_space(3))) v2f *p = (volatile __attribute__((address_space(3))) v2f *)(smem) + ((t & ~31) * 32 + (t & 31));
#pragma unroll
            for (int m = 0; m < 32; ++m) p[m * 32] = v[m];
        } else if constexpr (CASE == 7) {  // the same as pairs chosen by hipcc
            v2f *p = reinterpret_cast<v2f *>(smem) + ((t & ~31) * 32 + (t & 31));
#pragma unroll
            for (int m = 0; m < 32; ++m) p[m * 32] = v[m];
        } else if constexpr (CASE == 8) {  // reads for comparison: ds_read_b64 lane-consecutive
            volatile __attribute__((address_space(3))) v2f *p = (volatile __attribute__((address_space(3))) v2f *)(smem) + t;
#pragma unroll
            for (int m = 0; m < 32; ++m) v[m] += p[m * 512];
        }
#pragma unroll
        for (int m = 0; m < 32; ++m) asm volatile("" : "+v"(v[m]));
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (t == 0) out[blockIdx.x] = t1 - t0;
    if (v[3].x == 123.0f) out[0] = 0;
}

template <int CASE> void run(const char *name, unsigned long long *d_out, int grid) {
    const size_t lds = 160 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<CASE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<unsigned long long> h(grid);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<CASE>, dim3(grid), dim3(512), lds, 0, d_out, 1.0f);
    (void)hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[grid / 2] / REPS;
    printf("%-72s %7.0f cycles per 128 KiB and workgroup = %5.1f B/clk/CU\n", name, cyc, 131072.0 / cyc);
}

int main() {
    unsigned long long *d_out;
    const int grid = 256;
    (void)hipMalloc(&d_out, grid * 8);
    run<0>("ds_write_b64, lanes consecutive", d_out, grid);
    run<1>("ds_write_b64, lane stride 33 elements (single)", d_out, grid);
    run<2>("the same left to hipcc (ds_write2_b64 pairs)", d_out, grid);
    run<3>("ds_write_b128, lane stride 34 elements", d_out, grid);
    run<4>("ds_write_b32, real / imaginary planes, lanes consecutive", d_out, grid);
    run<5>("ds_write_addtid_b32, real / imaginary planes", d_out, grid);
    run<6>("ds_write_b64, base[r * 32], lanes in runs of 32 (single)", d_out, grid);
    run<7>("the same left to hipcc", d_out, grid);
    run<8>("ds_read_b64, lanes consecutive (reads, for scale)", d_out, grid);
    return 0;
}
