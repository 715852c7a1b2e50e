#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned char* p, float* o, unsigned nbytes, unsigned soff, int flags_sel) {
  __amdgpu_buffer_rsrc_t r = flags_sel ? __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000)
                                       : __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00027000);
  u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, threadIdx.x * 8, soff, 0);
  __amdgpu_buffer_rsrc_t w = flags_sel ? __builtin_amdgcn_make_buffer_rsrc((void*)o, 0, 64 * 8, 0x00020000)
                                       : __builtin_amdgcn_make_buffer_rsrc((void*)o, 0, 64 * 8, 0x00027000);
  __builtin_amdgcn_raw_buffer_store_b32(__uint_as_float(a.x), w, threadIdx.x * 4, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(__uint_as_float(a.y), w, threadIdx.x * 4, 256, 0);
}
int main() {
  std::vector<float> h(64 * 2 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  unsigned char* d; float* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int sel = 0; sel < 2; ++sel) for (unsigned soff : {0u, 512u, 1536u}) for (unsigned nb : {2048u, 1024u}) {
    hipMemset(o, 0xFF, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, nb, soff, sel);
    std::vector<float> r(128);
    hipMemcpy(r.data(), o, 512, hipMemcpyDeviceToHost);
    printf("flags_sel %d soff %u num_records %u: out[0..3] = %g %g %g %g  out[64]=%g out[127]=%g  err=%s\n", sel, soff, nb, r[0], r[1], r[2], r[3], r[64], r[127], hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
