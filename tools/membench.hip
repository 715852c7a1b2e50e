// membench.hip -- development tool: what does the memory system deliver for the spectrogram's
// traffic shape (read B bytes, write B bytes, both streaming) without any FFT in between?
//   mode 0: float4 grid-stride copy (the guide's "copy ceiling" shape)
//   mode 1: the v2 kernel's shape -- one 256-thread workgroup per run of `run` lines; per line
//           NLD 8-byte loads per lane at stride 2 KiB (next line prefetched) and 16 4-byte
//           stores per lane at stride 1 KiB; LDS reservation sets the workgroups per CU
//   mode 2: as 1, prefetch two lines ahead
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_f4(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

// lines of 2048 new cf32 samples in (16 KiB), 4096 floats out (16 KiB)
template <int DEPTH, int AUX>
__global__ __launch_bounds__(256) void stream_lines(const uint8_t *in, float *out, uint32_t n_lines, uint32_t run) {
    extern __shared__ unsigned char smem[];
    const int t = threadIdx.x;
    const uint32_t line0 = blockIdx.x * run;
    uint32_t lines = n_lines - line0 < run ? n_lines - line0 : run;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(in) + (uint64_t)line0 * 16384, 0, lines * 16384u, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out + (uint64_t)line0 * 4096, 0, lines * 16384u, 0x00020000);
    u32x2 raw[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int m = 0; m < 8; ++m) raw[d][m] = __builtin_amdgcn_raw_buffer_load_b64(src, t * 8, d * 16384 + m * 2048, AUX);
    for (uint32_t line = 0; line < lines; line += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            float v[16];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                v[2 * m] = __uint_as_float(raw[d][m].x) * 1.5f;
                v[2 * m + 1] = __uint_as_float(raw[d][m].y) * 1.5f;
            }
#pragma unroll
            for (int m = 0; m < 8; ++m)
                raw[d][m] = __builtin_amdgcn_raw_buffer_load_b64(src, t * 8, (int)((line + DEPTH + d) * 16384u) + m * 2048, AUX);
#pragma unroll
            for (int m = 0; m < 16; ++m)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[m]), dst, t * 4, (int)((line + d) * 16384u) + m * 1024, AUX);
        }
    }
    if (smem[0] == 77 && t == 9999) out[0] = 0;  // keep the LDS reservation
}

int main(int argc, char **argv) {
    const size_t B = (size_t)1 << 33;  // 8 GiB in, 8 GiB out
    uint8_t *in; float *out;
    CK(hipMalloc(&in, B)); CK(hipMalloc(&out, B));
    CK(hipMemset(in, 1, B)); CK(hipMemset(out, 0, B));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 8; ++i) launch();
        std::vector<float> ms;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-64s %7.3f ms  %6.0f GB/s (read+write)\n", name, ms[ms.size() / 2], 2.0 * B / ms[ms.size() / 2] / 1e6);
        fflush(stdout);
    };
    for (int wgs : {2048, 8192, 65536})
        for (int lds : {0, 40000}) {
            char nm[128]; snprintf(nm, sizeof nm, "float4 grid-stride copy, %d workgroups, lds %d", wgs, lds);
            timeit(nm, [&] { hipLaunchKernelGGL(copy_f4, dim3(wgs), dim3(256), lds, 0, (const f4 *)in, (f4 *)out, B / 16); });
        }
    const uint32_t n_lines = (uint32_t)(B / 16384);
    for (int lds : {1000, 30000, 39000, 52000}) {  // 8 (wave-limited), 5, 4, 3 workgroups per CU
        for (uint32_t run : {8u, 32u, 128u}) {
            char nm[128];
            const uint32_t wgs = (n_lines + run - 1) / run;
            snprintf(nm, sizeof nm, "line stream depth 1 nt, lds %d, run %u", lds, run);
            timeit(nm, [&] { hipLaunchKernelGGL((stream_lines<1, 2>), dim3(wgs), dim3(256), lds, 0, in, out, n_lines, run); });
            snprintf(nm, sizeof nm, "line stream depth 2 nt, lds %d, run %u", lds, run);
            timeit(nm, [&] { hipLaunchKernelGGL((stream_lines<2, 2>), dim3(wgs), dim3(256), lds, 0, in, out, n_lines, run); });
            if (run == 32) {
                snprintf(nm, sizeof nm, "line stream depth 1 default policy, lds %d, run %u", lds, run);
                timeit(nm, [&] { hipLaunchKernelGGL((stream_lines<1, 0>), dim3(wgs), dim3(256), lds, 0, in, out, n_lines, run); });
            }
        }
    }
    return 0;
}
