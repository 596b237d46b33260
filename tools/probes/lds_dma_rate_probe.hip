// Per-CU global -> LDS fetch rate from an L2-resident matrix, as a function of the contiguous bytes fetched per row
// (development probe: decides the K-chunk width of the conv GEMM ring).  hipcc --offload-arch=gfx950 -O3 -o p p.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
// 4 waves; every wave-instruction moves 1 KiB: ROWB = 64 -> 16 rows x 64 B, 128 -> 8 rows x 128 B, 256 -> 4 rows x 256 B
template <int ROWB, int DEPTH, int STAGGER, int REG>
__global__ __launch_bounds__(256) void k(const char* __restrict__ w, long pitch, int rows, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) char lds[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = ROWB / 16, RPI = 64 / LPR;             // lanes per row, rows per instruction
  const char* src = w + (long)(wave * RPI + lane / LPR) * pitch + (lane % LPR) * 16;
  long long t0 = clock64();
  int r = STAGGER ? (int)((blockIdx.x * 5u) % (rows / (4 * RPI))) * 4 * RPI : 0, col = STAGGER ? (int)(blockIdx.x % 16u) * 128 : 0;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 accv = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (REG) { u32x4 v = __builtin_nontemporal_load((const u32x4*)(src + (long)r * pitch + col)); accv ^= v; }
      else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)r * pitch + col),
                                       (__attribute__((address_space(3))) void*)(lds + ((it * DEPTH + d) & 31) * 4096 + wave * 1024), 16, 0, 0);
      r += 4 * RPI;
      if (r >= rows) { r = 0; col += ROWB; if (col >= pitch) col = 0; }
    }
    if (!REG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * 2 > 60 ? 60 : DEPTH * 2));
  }
  asm volatile("s_waitcnt vmcnt(0)");
  long long t1 = clock64();
  if (REG && accv[0] == 0x12345 && accv[1] == 77) cyc[300] = 1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ROWB, int DEPTH, int STAGGER = 0, int REG = 0>
void run(const char* w, long pitch, int rows, long long* cyc) {
  const int iters = 2000 / DEPTH;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<ROWB, DEPTH, STAGGER, REG>), dim3(256), dim3(256), 0, 0, w, pitch, rows, iters, cyc);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<ROWB, DEPTH, STAGGER, REG>), dim3(256), dim3(256), 0, 0, w, pitch, rows, iters, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double bytes = (double)iters * DEPTH * 4096;
  printf("%s%srow %3d B, %2d pieces/wave in flight: %6.1f B/clk/CU (s_memtime-like clock64), %6.2f TB/s chip, %.1f us\n", STAGGER ? "staggered " : "", REG ? "to-registers " : "", ROWB, DEPTH * 3,
         bytes / (double)h[0], bytes * 256 / (ms * 1e-3) / 1e12, ms * 1e3);
}
int main() {
  const long pitch = 2048; const int rows = 384;          // a 786 KB weight slice: L2 resident
  char* w; long long* cyc; hipMalloc(&w, pitch * rows); hipMalloc(&cyc, 512 * 8); hipMemset(w, 1, pitch * rows);
  run<64, 4>(w, pitch, rows, cyc); run<64, 9>(w, pitch, rows, cyc); run<64, 18>(w, pitch, rows, cyc);
  run<128, 4>(w, pitch, rows, cyc); run<128, 9>(w, pitch, rows, cyc); run<128, 18>(w, pitch, rows, cyc);
  run<256, 4>(w, pitch, rows, cyc); run<256, 9>(w, pitch, rows, cyc); run<256, 18>(w, pitch, rows, cyc);
  run<64, 9, 1>(w, pitch, rows, cyc); run<128, 9, 1>(w, pitch, rows, cyc); run<128, 18, 1>(w, pitch, rows, cyc);
  run<64, 8, 0, 1>(w, pitch, rows, cyc); run<128, 8, 0, 1>(w, pitch, rows, cyc); run<128, 8, 1, 1>(w, pitch, rows, cyc);
  return 0;
}
