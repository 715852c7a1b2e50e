// membench_rw.hip -- development tool: what the memory system delivers per DIRECTION.  The spectrogram kernels with 2- and 4-byte
// input formats write twice to four times the bytes they read; this measures a write-only stream, a read-only stream and mixed
// streams of r bytes read per w bytes written, all with 16-byte accesses, grid-stride, non-temporal where it is a hint.
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/membench_rw.hip -o /tmp/membench_rw && /tmp/membench_rw
// (spectral_analyzer_amd/build.py build_tools() compiles it to lib/membench_rw; `membench_rw R W` is what bench.py runs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// every thread: R loads and W stores of 16 bytes per trip, grid-stride over `trips`
template <int R, int W>
__global__ __launch_bounds__(256) void rw_kernel(const f4 *__restrict__ in, f4 *__restrict__ out, size_t trips) {
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < trips; i += stride) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc += __builtin_nontemporal_load(in + (size_t)r * trips + i);   // R coalesced streams
#pragma unroll
        for (int w = 0; w < W; ++w) __builtin_nontemporal_store(acc + (float)w, out + (size_t)w * trips + i);  // W coalesced streams
    }
    if (W == 0 && acc.x == 123.456f) out[0] = acc;  // keep the loads
}

template <int R, int W> void run(const char *name, const f4 *in, f4 *out, size_t bytes_total, int wgs) {
    const size_t per_trip = (size_t)(R + W) * 16, trips = bytes_total / per_trip;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((rw_kernel<R, W>), dim3(wgs), dim3(256), 0, 0, in, out, trips);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep >= 2 && ms < best) best = ms;
    }
    const double rd = (double)trips * R * 16, wr = (double)trips * W * 16;
    printf("%-34s %6d workgroups  %7.3f ms   read %6.0f GB/s   write %6.0f GB/s   total %6.0f GB/s\n", name, wgs, best,
           rd / best / 1e6, wr / best / 1e6, (rd + wr) / best / 1e6);
}

template <int R, int W> double best_total(const f4 *in, f4 *out, size_t bytes_total, double *rd_out, double *wr_out) {
    const size_t trips = bytes_total / ((size_t)(R + W) * 16);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int wgs : {16384, 65536})
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((rw_kernel<R, W>), dim3(wgs), dim3(256), 0, 0, in, out, trips);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 2 && ms < best) best = ms;
        }
    *rd_out = (double)trips * R * 16 / best / 1e6;
    *wr_out = (double)trips * W * 16 / best / 1e6;
    return *rd_out + *wr_out;
}

// `membench_rw R W`: ONE mix, one JSON line (bench.py's roofline.stream: the box's own ceiling for the workload's read : write mix)
int probe(int R, int W) {
    const size_t B = (size_t)1 << 32;  // 4 GiB each side
    f4 *in, *out;
    CK(hipMalloc(&in, B)); CK(hipMalloc(&out, B));
    CK(hipMemset(in, 1, B)); CK(hipMemset(out, 0, B));
    double rd = 0, wr = 0, tot = -1;
    if (R == 2 && W == 2) tot = best_total<2, 2>(in, out, B, &rd, &wr);
    else if (R == 1 && W == 2) tot = best_total<1, 2>(in, out, B, &rd, &wr);
    else if (R == 1 && W == 4) tot = best_total<1, 4>(in, out, B, &rd, &wr);
    else if (R == 2 && W == 1) tot = best_total<2, 1>(in, out, B, &rd, &wr);
    else if (R == 4 && W == 0) tot = best_total<4, 0>(in, out, B, &rd, &wr);
    else if (R == 0 && W == 4) tot = best_total<0, 4>(in, out, B, &rd, &wr);
    if (tot < 0) { printf("{\"error\": \"unsupported mix\"}\n"); return 2; }
    printf("{\"read_parts\": %d, \"write_parts\": %d, \"GBps\": %.1f, \"read_GBps\": %.1f, \"write_GBps\": %.1f}\n", R, W, tot, rd, wr);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 3) return probe(atoi(argv[1]), atoi(argv[2]));
    const size_t B = (size_t)1 << 33;  // 8 GiB each side
    f4 *in, *out;
    CK(hipMalloc(&in, B)); CK(hipMalloc(&out, B));
    CK(hipMemset(in, 1, B)); CK(hipMemset(out, 0, B));
    for (int wgs : {2048, 16384, 65536}) {
        run<0, 4>("write only", in, out, B, wgs);
        run<4, 0>("read only", in, out, B, wgs);
        run<2, 2>("read 1 : write 1 (cf32 lines)", in, out, B, wgs);
        run<1, 2>("read 1 : write 2 (ci16 lines)", in, out, B, wgs);
        run<1, 4>("read 1 : write 4 (cu8 lines)", in, out, B, wgs);
        run<2, 1>("read 2 : write 1", in, out, B, wgs);
    }
    return 0;
}
