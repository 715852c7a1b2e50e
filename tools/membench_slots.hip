// membench_slots.hip -- development tool (round 4): does the memory system move more when the whole chip alternates between
// READ-only and WRITE-only time slots?  Plain streams reach 7.0 TB/s reading and 6.8 TB/s writing on this part, a 1:1 mix
// 5.5-5.9 TB/s (tools/membench.hip) -- the ceiling of every spectrogram kernel here.  If the loss is the DRAM bus turning
// around, slots aligned on the chip-wide 100 MHz wall clock (s_memrealtime: no communication needed) should recover some.
//   every workgroup (256 threads): wait for a read slot, load CH x 16 bytes per thread, wait for them; wait for a write slot,
//   store them, wait.  SLOT = slot length in ticks of 10 ns; 0 = no slots (plain chunked copy with the same code).
// hipcc -O3 --offload-arch=gfx950 tools/membench_slots.hip -o /tmp/membench_slots && /tmp/membench_slots
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256) void slots(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n_chunks, unsigned slot_ticks) {
    // chunk c: CH * 256 consecutive f4; workgroup b takes chunks b, b + grid, ...
    const int t = threadIdx.x;
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const f4 *src = in + c * (size_t)(CH * 256) + t;
        f4 *dst = out + c * (size_t)(CH * 256) + t;
        if (slot_ticks) while ((wall_clock64() / slot_ticks) & 1) __builtin_amdgcn_s_sleep(2);   // read slots: even
        f4 v[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) v[k] = __builtin_nontemporal_load(src + k * 256);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (slot_ticks) while (!((wall_clock64() / slot_ticks) & 1)) __builtin_amdgcn_s_sleep(2);  // write slots: odd
#pragma unroll
        for (int k = 0; k < CH; ++k) __builtin_nontemporal_store(v[k], dst + k * 256);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

int main() {
    const size_t B = (size_t)1 << 32;  // 4 GiB in, 4 GiB out
    f4 *in, *out;
    CK(hipMalloc(&in, B)); CK(hipMalloc(&out, B));
    CK(hipMemset(in, 1, B)); CK(hipMemset(out, 0, B));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        std::vector<float> ms;
        for (int i = 0; i < 7; ++i) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-72s %7.3f ms  %6.0f GB/s (read+write)\n", name, ms[ms.size() / 2], 2.0 * B / ms[ms.size() / 2] / 1e6);
        fflush(stdout);
    };
    for (int grid : {768, 1536, 2048}) {
        for (unsigned slot : {0u, 100u, 200u, 400u, 800u, 1600u}) {
            char nm[160];
            snprintf(nm, sizeof nm, "chunk 32 KiB/WG (8 x 16 B per thread), %4d WGs, slot %5.1f us", grid, slot / 100.0);
            timeit(nm, [&] { hipLaunchKernelGGL((slots<8>), dim3(grid), dim3(256), 0, 0, in, out, B / (8 * 256 * 16), slot); });
            snprintf(nm, sizeof nm, "chunk 64 KiB/WG (16 x 16 B per thread), %4d WGs, slot %5.1f us", grid, slot / 100.0);
            timeit(nm, [&] { hipLaunchKernelGGL((slots<16>), dim3(grid), dim3(256), 0, 0, in, out, B / (16 * 256 * 16), slot); });
        }
    }
    return 0;
}
