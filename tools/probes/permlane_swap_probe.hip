// probe: semantics of v_permlane32_swap on gfx950 (which halves are exchanged)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t* out) {
  uint32_t a = threadIdx.x, b = 100 + threadIdx.x;
  u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[threadIdx.x + 64] = r[1];
}
int main() {
  uint32_t* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint32_t h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
  printf("r1: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
