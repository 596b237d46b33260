// Do VALU instructions of the SAME wave (and of a second wave on the SIMD) issue while the matrix pipe works on an MFMA?
// Loop body: NM independent v_mfma_f32_32x32x16_bf16 + NV independent VALU instructions (v_add_f32 / v_cvt_pk_bf16_f32 / v_max_f32),
// one workgroup of 4 waves per CU x W.  Prints cycles per loop iteration per SIMD next to the MFMA-only and VALU-only figures.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_probe.hip -o mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NM, int NV, int KIND> __global__ __launch_bounds__(256) void k(float* out, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + threadIdx.x * 1e-3f); b[e] = (__bf16)(seed * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float& x = v[(m * NV + j) & 7];
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
        if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
        if (KIND == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
      }
    }
    if (NM == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(seed));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static int G = 256;
template <typename F> double run(F launch) {
  float* out; hipMalloc(&out, 8192 * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(out); hipDeviceSynchronize();
  hipEventRecord(a); launch(out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipFree(out);
  return ms * 1e-3 * 2.4e9 / ITERS / (G / 256.0);   // cycles per iteration per wave-slot... per SIMD: (G/256) waves share a SIMD
}
int main() {
  for (G = 256; G <= 1024; G *= 2) {
    printf("---- %d wave(s) per SIMD: cycles per loop iteration of ONE wave's share of the SIMD (x waves = SIMD time)\n", G / 256);
    printf("4 MFMA only            %7.1f\n", run([](float* o) { k<4, 0, 0><<<G, 256>>>(o, 1.0001f); }));
    printf("16 v_add only          %7.1f\n", run([](float* o) { k<0, 16, 0><<<G, 256>>>(o, 1.0001f); }));
    printf("4 MFMA + 4x1 v_add     %7.1f\n", run([](float* o) { k<4, 1, 0><<<G, 256>>>(o, 1.0001f); }));
    printf("4 MFMA + 4x2 v_add     %7.1f\n", run([](float* o) { k<4, 2, 0><<<G, 256>>>(o, 1.0001f); }));
    printf("4 MFMA + 4x4 v_add     %7.1f\n", run([](float* o) { k<4, 4, 0><<<G, 256>>>(o, 1.0001f); }));
    printf("4 MFMA + 4x6 v_add     %7.1f\n", run([](float* o) { k<4, 6, 0><<<G, 256>>>(o, 1.0001f); }));
    printf("4 MFMA + 4x8 v_add     %7.1f\n", run([](float* o) { k<4, 8, 0><<<G, 256>>>(o, 1.0001f); }));
    printf("4 MFMA + 4x4 v_cvt_pk  %7.1f\n", run([](float* o) { k<4, 4, 1><<<G, 256>>>(o, 1.0001f); }));
    printf("4 MFMA + 4x4 v_max     %7.1f\n", run([](float* o) { k<4, 4, 2><<<G, 256>>>(o, 1.0001f); }));
  }
  return 0;
}
