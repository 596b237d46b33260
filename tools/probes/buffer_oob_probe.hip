#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint32_t* in, uint32_t* out, int n) {
  __amdgpu_buffer_rsrc_t si = __builtin_amdgcn_make_buffer_rsrc(in, 0, n * 16, 0x00020000);
  __amdgpu_buffer_rsrc_t so = __builtin_amdgcn_make_buffer_rsrc(out, 0, n * 16, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(si, threadIdx.x * 16, 0, 0);
  v[0] += 1;
  __builtin_amdgcn_raw_buffer_store_b128(v, so, threadIdx.x * 16, 0, 0);
}
int main() {
  uint32_t *a, *b; hipMalloc(&a, 64 * 16); hipMalloc(&b, 64 * 16);
  hipMemset(a, 0, 64 * 16); hipMemset(b, 0xff, 64 * 16);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, 40);
  uint32_t h[256]; hipMemcpy(h, b, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane0 %x lane39 %x lane40 %x (expect 1 1 ffffffff)\n", h[0], h[39 * 4], h[40 * 4]);
  return 0;
}
