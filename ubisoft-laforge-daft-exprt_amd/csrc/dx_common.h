// Shared device/host helpers for the Daft-Exprt gfx950 kernels.
// Everything here targets CDNA4 (wave64, MFMA 32x32x16 bf16 / 32x32x2 f32) directly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/daft_exprt_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DX_WAVE 64

// ---- error plumbing (host) ------------------------------------------------------------
void dx_set_error(const char* fmt, ...);
#define DX_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      dx_set_error(__VA_ARGS__);               \
      return (code);                           \
    }                                          \
  } while (0)
#define DX_LAUNCH_CHECK()                                                   \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      dx_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return DX_ERR_LAUNCH;                                                 \
    }                                                                       \
  } while (0)

__host__ __device__ static inline int dx_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------------------
template <typename T>
struct Vec8;
template <>
struct Vec8<bf16_t> { typedef bf16x8 type; };
template <>
struct Vec8<float> { typedef f32x8 type; };

// load 8 consecutive elements of TA from global and convert to TC
template <typename TA, typename TC>
__device__ __forceinline__ typename Vec8<TC>::type dx_load8(const TA* p);

template <>
__device__ __forceinline__ bf16x8 dx_load8<bf16_t, bf16_t>(const bf16_t* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
template <>
__device__ __forceinline__ bf16x8 dx_load8<float, bf16_t>(const float* p) {
  f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  bf16x8 r;
  r[0] = (bf16_t)lo[0]; r[1] = (bf16_t)lo[1]; r[2] = (bf16_t)lo[2]; r[3] = (bf16_t)lo[3];
  r[4] = (bf16_t)hi[0]; r[5] = (bf16_t)hi[1]; r[6] = (bf16_t)hi[2]; r[7] = (bf16_t)hi[3];
  return r;
}
template <>
__device__ __forceinline__ f32x8 dx_load8<float, float>(const float* p) {
  f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return r;
}
template <>
__device__ __forceinline__ f32x8 dx_load8<bf16_t, float>(const bf16_t* p) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
  return r;
}

// One K=16 step of a 32x32 output tile.  Lane l holds row/col (l & 31) and the 8 k-values
// 8*(l>>5)..+7 of the step for both operands.  bf16: one v_mfma_f32_32x32x16_bf16.
// fp32: eight v_mfma_f32_32x32x2_f32 (exact fp32; instruction j contracts k = j and 8+j).
__device__ __forceinline__ void dx_mma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void dx_mma(f32x16& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}

// row index inside a 32x32 accumulator tile for register r of lane-group g = lane >> 5
__device__ __forceinline__ int dx_acc_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

template <typename T>
__device__ __forceinline__ float dx_to_f32(T v) { return (float)v; }

__device__ __forceinline__ float dx_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float dx_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// k-major gather from a row-major tile: lane gets column c = col0 + (lane & 31) % WRAP and the 8 rows
// {kA..kA+3, kB..kB+3}.
template <typename TC, int WRAP>
__device__ __forceinline__ typename Vec8<TC>::type gather8(const TC* tile, int ld, int kA, int kB, int col0, int lane);

template <>
__device__ __forceinline__ bf16x8 gather8<bf16_t, 32>(const bf16_t* tile, int ld, int kA, int kB, int col0, int lane) {
  const int i = lane & 15, half = (lane >> 4) & 1, j = i >> 2, q = i & 3;
  const bf16_t* pa = tile + (kA + j) * ld + col0 + 16 * half + 4 * q;
  const bf16_t* pb = tile + (kB + j) * ld + col0 + 16 * half + 4 * q;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}
template <>
__device__ __forceinline__ bf16x8 gather8<bf16_t, 16>(const bf16_t* tile, int ld, int kA, int kB, int col0, int lane) {
  const int i = lane & 15, j = i >> 2, q = i & 3;
  const bf16_t* pa = tile + (kA + j) * ld + col0 + 4 * q;
  const bf16_t* pb = tile + (kB + j) * ld + col0 + 4 * q;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}
template <>
__device__ __forceinline__ f32x8 gather8<float, 32>(const float* tile, int ld, int kA, int kB, int col0, int lane) {
  const int c = col0 + (lane & 31);
  f32x8 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) { r[t] = tile[(kA + t) * ld + c]; r[4 + t] = tile[(kB + t) * ld + c]; }
  return r;
}
template <>
__device__ __forceinline__ f32x8 gather8<float, 16>(const float* tile, int ld, int kA, int kB, int col0, int lane) {
  const int c = col0 + (lane & 15);
  f32x8 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) { r[t] = tile[(kA + t) * ld + c]; r[4 + t] = tile[(kB + t) * ld + c]; }
  return r;
}

template <typename TC>
__device__ __forceinline__ typename Vec8<TC>::type zero8() {
  typename Vec8<TC>::type v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (TC)0.f;
  return v;
}
template <typename TC>
__device__ __forceinline__ typename Vec8<TC>::type pack8(const float* p) {
  typename Vec8<TC>::type v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (TC)p[e];
  return v;
}

// Counter-based dropout RNG: one 32-bit hash (murmur3 finaliser) per element.  keep iff hash >= p * 2^32.
// `key` is a 32-bit stream key folded on the fly from the 64-bit call seed (+ a per-(batch, head) salt in attention);
// `idx` is the element index inside the tensor (fits 32 bits for every tensor of this model, B*N*C < 2^32).
// Bit-parity with torch's Philox stream is not a goal; the forward and backward kernels
// regenerate the same mask from (seed, element index).  64-bit multiplies are avoided on purpose: they cost 4
// quarter-rate v_mul each and made the d_head = 16 attention kernels VALU-bound.
__host__ __device__ __forceinline__ uint32_t dx_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t dx_key32(uint64_t seed, uint32_t salt) {
  return dx_mix32((uint32_t)seed ^ dx_mix32((uint32_t)(seed >> 32) + 0x9E3779B9u * (salt + 1u)));
}
// ---- attention-weight dropout: counter based, built from full-rate integer ops only (v_mul_u32_u24; the 32-bit
// v_mul_lo_u32 of dx_mix32 is quarter rate on CDNA and used to dominate the attention kernels, which are VALU bound).
//   counter(q, key) = (q * N + (key & ~1)) * DX_CTR_MUL + stream_key                       (wraps mod 2^32)
//   keep(q, key)    = mul24(prefix(counter), key odd ? DX_M24_ODD : DX_M24_EVEN) >= p * 2^32
// An (even, odd) key pair shares the 4-op prefix; measured on 2M counters: keep rates within 5e-4 of 1 - p, pair /
// neighbour / row / column correlations < 2e-3, chi^2 of both fields ~1.0 per degree of freedom.
constexpr uint32_t DX_CTR_MUL = 0x9E3779B1u, DX_M24_PRE = 0x9E3779u, DX_M24_EVEN = 0xC2B2AFu, DX_M24_ODD = 0x85EBCBu;
__device__ __forceinline__ uint32_t dx_drop_prefix(uint32_t ctr) {
  ctr ^= ctr >> 16;
  ctr = __umul24(ctr, DX_M24_PRE);
  ctr ^= ctr >> 13;
  return ctr;   // the field multiply reads its low 24 bits
}
__device__ __forceinline__ uint32_t dx_drop_field(uint32_t prefix, uint32_t mult24) { return __umul24(prefix, mult24); }

__device__ __forceinline__ bool dx_keep(uint32_t key, uint32_t idx, uint32_t thresh) {
  return dx_mix32(idx * 0x9E3779B1u + key) >= thresh;
}
