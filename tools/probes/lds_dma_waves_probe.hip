// Does the global -> LDS (LDS-DMA) fetch rate of a CU scale with the number of waves issuing the pieces?
// NW waves per workgroup (one workgroup per CU), each wave-instruction moves 1 KiB (16 rows x 64 B of a 2 KB-pitch, L2-resident
// 786 KB matrix -- the weight slice of the 1024 -> 128 k = 3 convolution), DEPTH pieces per wave between waits.
//   hipcc --offload-arch=gfx950 -O3 lds_dma_waves_probe.hip -o p
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NW, int DEPTH, int REG>
__global__ __launch_bounds__(NW * 64) void k(const char* __restrict__ w, long pitch, int rows, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) char lds[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* src = w + (long)(lane >> 2) * pitch + (lane & 3) * 16;
  int r = (wave * 16) % rows, col = 0;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 accv = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (REG) { u32x4 v = *(const u32x4*)(src + (long)r * pitch + col); accv ^= v; }
      else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)r * pitch + col),
                                            (__attribute__((address_space(3))) void*)(lds + ((wave * DEPTH + d) & 127) * 1024), 16, 0, 0);
      r += NW * 16;
      if (r >= rows) { r -= rows; col += 64; if (col >= pitch) col = 0; }
    }
    if (!REG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH));
  }
  asm volatile("s_waitcnt vmcnt(0)");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (REG && accv[0] == 0x12345 && accv[1] == 77) cyc[300] = 1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = (long long)(t1 - t0);
}
template <int NW, int DEPTH, int REG = 0> void run(const char* w, long pitch, int rows, long long* cyc) {
  const int iters = 4096 / DEPTH / NW * 4;
  hipLaunchKernelGGL((k<NW, DEPTH, REG>), dim3(256), dim3(NW * 64), 0, 0, w, pitch, rows, iters, cyc);
  hipLaunchKernelGGL((k<NW, DEPTH, REG>), dim3(256), dim3(NW * 64), 0, 0, w, pitch, rows, iters, cyc);
  hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double t = 0; for (int i = 0; i < 256; ++i) t += h[i]; t = t / 256 * 10e-9;   // mean seconds per workgroup
  const double bytes = (double)iters * DEPTH * NW * 1024;
  printf("%s%2d waves, %2d pieces per wave in flight: %6.1f GB/s per CU = %5.2f TB/s chip-wide (%.0f ns per 1 KiB piece per CU)\n", REG ? "to registers, " : "LDS-DMA,      ", NW, DEPTH,
         bytes / t / 1e9, bytes / t * 256 / 1e12, t / (iters * DEPTH * NW) * 1e9);
}
int main() {
  const long pitch = 2048; const int rows = 384;
  char* w; long long* cyc; hipMalloc(&w, pitch * rows); hipMalloc(&cyc, 512 * 8); hipMemset(w, 1, pitch * rows);
  run<1, 8>(w, pitch, rows, cyc); run<2, 8>(w, pitch, rows, cyc); run<4, 2>(w, pitch, rows, cyc); run<4, 8>(w, pitch, rows, cyc); run<4, 16>(w, pitch, rows, cyc);
  run<8, 4>(w, pitch, rows, cyc); run<8, 8>(w, pitch, rows, cyc); run<12, 8>(w, pitch, rows, cyc); run<16, 8>(w, pitch, rows, cyc);
  run<4, 8, 1>(w, pitch, rows, cyc); run<8, 8, 1>(w, pitch, rows, cyc); run<16, 8, 1>(w, pitch, rows, cyc);
  return 0;
}
