// Hardware probe (development aid, not product): dumps the lane->element mapping of
// ds_read_b64_tr_b16 so that LDS transpose-read layouts can be designed against facts.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void tr(short* out) {
  __shared__ short lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (short)i;
  __syncthreads();
  // each lane passes the address of its own 8-byte chunk, chunks laid out linearly
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 512); hipLaunchKernelGGL(tr, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
