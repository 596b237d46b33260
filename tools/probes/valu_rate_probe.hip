// Issue rate of a few VALU instructions on gfx950, one workgroup of 4 waves per CU x 2 (2 waves / SIMD), 8 independent chains
// per lane.  Prints cycles per wave-instruction per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
template <int OP> __global__ __launch_bounds__(256) void k(float* out, float seed) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
      if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
      if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
      if (OP == 5) asm volatile("v_cmp_ge_u32_sdwa vcc, %0, %1 src0_sel:BYTE_1 src1_sel:DWORD" : : "v"(v[i]), "v"(seed) : "vcc");
      if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(v[i]) : "v"(seed));
      if (OP == 7) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(seed));
      if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> __global__ __launch_bounds__(256) void k2(float* out, float seed) {   // packed: two floats per instruction
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f2{seed + threadIdx.x * 1e-3f + i, seed};
  f2 c = {seed, seed};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// one dependent chain per lane (ILP = 1): the issue interval of back-to-back dependent instructions
template <int OP> __global__ __launch_bounds__(256) void kd(float* out, float seed) {
  float v = seed + threadIdx.x * 1e-3f;
  unsigned u = threadIdx.x * 2654435761u + 12345u, c = 0x9E3779u;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(seed));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
      if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u) : "v"(c));
      if (OP == 3) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(u));
      if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1\n\tv_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(u) : "v"(c));
      if (OP == 5) asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %0, %0, %1" : "+v"(v) : "v"(seed));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = v + u;
}
static int G = 512;
template <typename F> void run_dep(const char* name, int per_iter, F launch) {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(out); hipDeviceSynchronize();
  hipEventRecord(a); launch(out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double inst_per_wave = (double)ITERS * 8 * per_iter;
  printf("%-34s %8.3f ms  -> %.2f cycles per instruction PER WAVE (dependent chain) at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / inst_per_wave / (G / 256.0) * (G / 256.0));
  hipFree(out);
}
template <typename F> void run(const char* name, F launch) {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(out); hipDeviceSynchronize();
  hipEventRecord(a); launch(out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // G workgroups x 4 waves over 256 CUs x 4 SIMDs, each wave issuing ITERS * 8 instructions
  double inst_per_simd = (G / 256.0) * ITERS * 8;
  printf("%-22s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
  hipFree(out);
}
int main() {
  for (G = 256; G <= 2048; G *= 2) {
  printf("---- %d waves per SIMD\n", G / 256);
  run("v_fma_f32", [](float* o) { k<0><<<G, 256>>>(o, 1.0001f); });
  run("v_pk_fma_f32", [](float* o) { k2<0><<<G, 256>>>(o, 1.0001f); });
  run("v_exp_f32", [](float* o) { k<1><<<G, 256>>>(o, 1.0001f); });
  run("v_rcp_f32", [](float* o) { k<8><<<G, 256>>>(o, 1.0001f); });
  run("v_mul_u32_u24", [](float* o) { k<2><<<G, 256>>>(o, 1.0001f); });
  run("v_mul_lo_u32", [](float* o) { k<3><<<G, 256>>>(o, 1.0001f); });
  run("v_cvt_pk_bf16_f32", [](float* o) { k<4><<<G, 256>>>(o, 1.0001f); });
  run("v_cmp_ge_u32_sdwa", [](float* o) { k<5><<<G, 256>>>(o, 1.0001f); });
  run("v_alignbit_b32", [](float* o) { k<6><<<G, 256>>>(o, 1.0001f); });
  run("v_max3_f32", [](float* o) { k<7><<<G, 256>>>(o, 1.0001f); });
  run_dep("dep v_fma_f32", 1, [](float* o) { kd<0><<<G, 256>>>(o, 1.0001f); });
  run_dep("dep v_exp_f32", 1, [](float* o) { kd<1><<<G, 256>>>(o, 1.0001f); });
  run_dep("dep v_mul_u32_u24", 1, [](float* o) { kd<2><<<G, 256>>>(o, 1.0001f); });
  run_dep("dep v_xor_b32_sdwa", 1, [](float* o) { kd<3><<<G, 256>>>(o, 1.0001f); });
  run_dep("dep mul24 -> xor_sdwa (pair)", 2, [](float* o) { kd<4><<<G, 256>>>(o, 1.0001f); });
  run_dep("dep exp -> add (pair)", 2, [](float* o) { kd<5><<<G, 256>>>(o, 1.0001f); });
  }
  return 0;
}
