// helpers shared by the conv GEMM (conv_gemm.hip), weight-gradient (conv_wgrad.hip) and weight-packing (conv_pack.hip) units;
// included INSIDE each unit's anonymous namespace
#pragma once

// zeros for the LDS-DMA lanes whose row lies outside the utterance / the weight matrix (read at offsets < 2 * Cin bytes)
constexpr int DX_ZERO_PAGE_EL = 4096;
__device__ __attribute__((aligned(16))) unsigned short dx_zero_page[DX_ZERO_PAGE_EL + 32];

template <typename TC> struct Pad;
template <> struct Pad<bf16_t> { static constexpr int value = 8; };
template <> struct Pad<float> { static constexpr int value = 4; };

template <typename T, int V> struct VecN;
template <> struct VecN<float, 8> { typedef f32x8 type; };
template <> struct VecN<bf16_t, 8> { typedef bf16x8 type; };

template <typename T>
__device__ __forceinline__ typename VecN<T, 8>::type raw_load8(const T* p);
template <>
__device__ __forceinline__ f32x8 raw_load8<float>(const float* p) {
  f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
  f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return r;
}
template <>
__device__ __forceinline__ bf16x8 raw_load8<bf16_t>(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <typename TS, typename TD>
__device__ __forceinline__ typename Vec8<TD>::type cvt8(const typename VecN<TS, 8>::type& v) {
  typename Vec8<TD>::type r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (TD)v[e];
  return r;
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float* v);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float* v) {
  f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
  *reinterpret_cast<f32x4*>(p) = lo;
  *reinterpret_cast<f32x4*>(p + 4) = hi;
}
template <>
__device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* v) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[e];
  *reinterpret_cast<bf16x8*>(p) = r;
}
